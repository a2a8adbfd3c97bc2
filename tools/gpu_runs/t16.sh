#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r3_overlap.log; : > $L
for pct in 15 0; do
  echo "LBM_FUSED_OVERLAP_PCT=$pct" >> $L
  LBM_FUSED_OVERLAP_PCT=$pct FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 --tunnel >> $L 2>&1
  LBM_FUSED_OVERLAP_PCT=$pct FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 1024 512 512 --tunnel >> $L 2>&1
  LBM_FUSED_OVERLAP_PCT=$pct FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 512 512 512 >> $L 2>&1
done
grep -v amdgpu.ids $L
