#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests/test_fused_gpu.py tests/test_parity_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_t36.log 2>&1
rc=$?
tail -n 15 gpurun_out/r3_t36.log
exit $rc
