#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3_t8.log 2>&1 || { tail -40 gpurun_out/r3_t8.log; exit 1; }
tail -2 gpurun_out/r3_t8.log
timeout -k 10 100 python3 $R/tools/curl_probe.py 2>/dev/null | tee gpurun_out/r3_nt_fields2.log
