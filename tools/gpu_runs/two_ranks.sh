#!/bin/bash
# The N = 2 code path of bench.py on a ONE-GPU box: both ranks on device 0 over gloo, halo messages through the host.
# Checks the branches no one-rank run reaches (ring neighbours, balanced re-cut of configs[3], per-rank figures).
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BENCH_BACKEND=gloo BENCH_LAUNCH_TIMEOUT=500
timeout -k 10 560 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r3_two_ranks.json 2> gpurun_out/r3_two_ranks.log
tail -c 3000 gpurun_out/r3_two_ranks.json
