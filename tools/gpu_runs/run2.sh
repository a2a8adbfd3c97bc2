#!/bin/bash
S="python tools/chain_spread.py 1024 512 64 --slab 0 512"
bash tools/gpu_chain.sh \
 "600 gpurun_out/r3_fused_tests2.log python -u -m pytest tests/test_fused_gpu.py tests/test_slab_pairs_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "120 gpurun_out/r3_ny_auto.log env LBM_FUSED_DEBUG=1 $S" \
 "120 gpurun_out/r3_ny43.log env LBM_FUSED_DEBUG=1 LBM_FUSED_NY=43 $S" \
 "120 gpurun_out/r3_ny46.log env LBM_FUSED_DEBUG=1 LBM_FUSED_NY=46 $S" \
 "120 gpurun_out/r3_ny48.log env LBM_FUSED_DEBUG=1 LBM_FUSED_NY=48 $S" \
 "120 gpurun_out/r3_ny51.log env LBM_FUSED_DEBUG=1 LBM_FUSED_NY=51 $S" \
 "120 gpurun_out/r3_ny64.log env LBM_FUSED_DEBUG=1 LBM_FUSED_NY=64 $S" \
 "120 gpurun_out/r3_256_auto.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 256 256 256" \
 "300 gpurun_out/r3_rehearse64_ny.log python bench.py --config 3 --size 1024 512 64 --rehearse-slab --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c1_ny.log python bench.py --no-sweep --no-cpu-baseline --no-strong-scaling"
for f in auto 43 46 48 51 64; do echo "== ny $f"; grep -E "^\[fused\] 1024|launch|mean end" gpurun_out/r3_ny$f.log | cut -c1-260; done
grep -E "launch|mean end|^\[fused\] 256" gpurun_out/r3_256_auto.log | cut -c1-260
