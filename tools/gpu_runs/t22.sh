#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r3_nohalo.log; : > $L
for lib in "" cfd_amd/csrc/variants/liblbm_nohalo.so; do
  echo "lib=$lib" >> $L
  LBM_LIB=${lib:+$R/$lib} FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 --tunnel >> $L 2>&1
done
grep -v amdgpu.ids $L
