#!/bin/bash
for z in 64 128; do
timeout -k 10 120 env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 1024 512 64 --slab $z 512 --car > gpurun_out/r3_sp_car_$z.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_sp_car_$z.log | cut -c1-300
done
