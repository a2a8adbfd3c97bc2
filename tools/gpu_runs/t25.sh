#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -u -m pytest tests/test_fused_gpu.py -x -q --timeout 300 --timeout-method thread -k "empty_box or obstacles or walls" > gpurun_out/r3_t25.log 2>&1 || { tail -60 gpurun_out/r3_t25.log; exit 1; }
tail -3 gpurun_out/r3_t25.log
L=gpurun_out/r3_wide2.log; : > $L
for w in 0 1; do
  echo "LBM_FUSED_WIDE=$w" >> $L
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 --tunnel >> $L 2>&1
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 512 256 256 --tunnel --car >> $L 2>&1
done
grep -v amdgpu.ids $L
