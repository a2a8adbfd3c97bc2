#!/bin/bash
bash tools/gpu_chain.sh \
 "1000 gpurun_out/r3_gpu_tests_b.log python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread" \
 "900 gpurun_out/r3_bench_full_b.log python bench.py"
tail -c 1500 gpurun_out/r3_bench_full_b.log
