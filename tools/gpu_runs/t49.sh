#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python3 -u -m pytest tests/test_fused_gpu.py -x -q -k "single_steps or where_the_rim" --timeout 300 --timeout-method thread > gpurun_out/r3_t49.log 2>&1 || { tail -40 gpurun_out/r3_t49.log; exit 1; }
tail -3 gpurun_out/r3_t49.log
