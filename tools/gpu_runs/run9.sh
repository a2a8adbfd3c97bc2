#!/bin/bash
bash tools/gpu_chain.sh \
 "900 gpurun_out/r3_fused_tests_d.log python -u -m pytest tests/test_fused_gpu.py tests/test_slab_pairs_gpu.py tests/test_full_size_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "900 gpurun_out/r3_bench_full_d.log env LBM_FUSED_DEBUG=1 python bench.py"
grep "^\[fused\]" gpurun_out/r3_bench_full_d.log | cut -c1-400 | sort | uniq -c | sort -rn | head -40
