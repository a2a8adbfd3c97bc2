#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r3_forms.log; : > $L
for form in 0 1 2; do
  echo "walls off, kernel form $form" >> $L
  LBM_PROBE=1 LBM_FUSED_FORM=$form LBM_FUSED_WALLS=0 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 --tunnel >> $L 2>&1
done
grep -v amdgpu.ids $L
