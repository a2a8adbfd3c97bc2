#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r3_edgecost.log; : > $L
for ec in "1,1" "0.91,0.88"; do
  echo "LBM_FUSED_EDGECOST=$ec" >> $L
  LBM_FUSED_EDGECOST=$ec LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 --tunnel >> $L 2>&1
  LBM_FUSED_EDGECOST=$ec LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 >> $L 2>&1
  LBM_FUSED_EDGECOST=$ec LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 512 256 256 --tunnel --car >> $L 2>&1
  LBM_FUSED_EDGECOST=$ec LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 1024 512 512 --tunnel --car >> $L 2>&1
  LBM_FUSED_EDGECOST=$ec LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" FB_STEPS=101 timeout -k 10 200 python3 tools/fused_bench.py 400 100 100 --tunnel >> $L 2>&1
done
grep -v amdgpu.ids $L
