#!/bin/bash
# all GPU tests, smoke, then the round's profile passes
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_full_gpu.log 2>&1 || { tail -60 gpurun_out/r3_full_gpu.log; exit 1; }
tail -3 gpurun_out/r3_full_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
