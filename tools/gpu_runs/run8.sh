#!/bin/bash
bash tools/gpu_chain.sh \
 "900 gpurun_out/r3_fused_tests_c.log python -u -m pytest tests/test_fused_gpu.py tests/test_slab_pairs_gpu.py tests/test_cpp_mirror_gpu.py tests/test_full_size_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "120 gpurun_out/r3_sp_c_64.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 1024 512 60 --slab 391 512 --car" \
 "120 gpurun_out/r3_sp_c_0.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 1024 512 64 --slab 0 512 --car" \
 "120 gpurun_out/r3_sp_c_256.log env LBM_FUSED_DEBUG=1 python tools/chain_spread.py 256 256 256" \
 "900 gpurun_out/r3_bench_full_c.log env LBM_FUSED_DEBUG=1 python bench.py"
grep "^\[fused\]\|launch " gpurun_out/r3_sp_c_*.log | cut -c1-420
grep "^\[fused\]" gpurun_out/r3_bench_full_c.log | cut -c1-400 | sort | uniq -c | sort -rn | head -40
