#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_two_ranks_one_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_t1.log 2>&1 || { tail -40 gpurun_out/r3_t1.log; exit 1; }
tail -5 gpurun_out/r3_t1.log
bash tools/gpu_runs/two_ranks.sh
