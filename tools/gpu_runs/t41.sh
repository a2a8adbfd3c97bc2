#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python3 -u -m pytest tests/test_two_stream_gpu.py tests/test_two_ranks_one_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_t41.log 2>&1 || { tail -60 gpurun_out/r3_t41.log; exit 1; }
tail -3 gpurun_out/r3_t41.log
