#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_two_ranks_one_gpu.py -x -q -k peer_copy --timeout 180 --timeout-method thread > gpurun_out/r3_ipc_test.log 2>&1 || { tail -40 gpurun_out/r3_ipc_test.log; exit 1; }
tail -2 gpurun_out/r3_ipc_test.log
bash tools/gpu_runs/t39.sh
