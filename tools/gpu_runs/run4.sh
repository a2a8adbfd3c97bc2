#!/bin/bash
S="python tools/chain_spread.py 1024 512 64 --slab 0 512"
bash tools/gpu_chain.sh \
 "120 gpurun_out/r3_sp_auto.log env LBM_FUSED_DEBUG=1 $S" \
 "120 gpurun_out/r3_sp_ncu128.log env LBM_FUSED_DEBUG=1 LBM_FUSED_NCU=384 $S" \
 "120 gpurun_out/r3_sp_256.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 256 256 256"
cat gpurun_out/r3_sp_auto.log gpurun_out/r3_sp_ncu128.log gpurun_out/r3_sp_256.log | grep -v "y tile\|amdgpu.ids"
