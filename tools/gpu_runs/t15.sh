#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_full_gpu.log 2>&1 || { tail -60 gpurun_out/r3_full_gpu.log; exit 1; }
tail -3 gpurun_out/r3_full_gpu.log
