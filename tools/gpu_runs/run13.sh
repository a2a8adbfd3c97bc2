#!/bin/bash
bash tools/gpu_chain.sh \
 "900 gpurun_out/r3_parity_g.log python -u -m pytest tests/test_parity_gpu.py tests/test_fused_gpu.py tests/test_cpp_mirror_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "600 gpurun_out/r3_bench_g.log python bench.py --no-strong-scaling --no-cpu-baseline"
