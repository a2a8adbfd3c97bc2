#!/bin/bash
# first GPU check of the chain planner
bash tools/gpu_chain.sh \
 "600 gpurun_out/r3_fused_tests.log python -u -m pytest tests/test_fused_gpu.py tests/test_slab_pairs_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "300 gpurun_out/r3_fused_tests_ncu8.log env LBM_FUSED_NCU=8 python -u -m pytest tests/test_fused_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "120 gpurun_out/r3_spread_slab64.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 1024 512 64 --slab 0 512" \
 "120 gpurun_out/r3_spread_slab64_old.log env LBM_FUSED_DEBUG=1 LBM_FUSED_CHAINS=0 python tools/chain_spread.py 1024 512 64 --slab 0 512" \
 "120 gpurun_out/r3_spread_256.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 256 256 256" \
 "300 gpurun_out/r3_rehearse64_chains.log python bench.py --config 3 --size 1024 512 64 --rehearse-slab --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c1_chains.log python bench.py --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c1_nochains.log env LBM_FUSED_CHAINS=0 python bench.py --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c3_chains.log python bench.py --config 3 --no-sweep --no-cpu-baseline --no-strong-scaling"
