#!/bin/bash
bash tools/gpu_chain.sh \
 "1000 gpurun_out/r3_gpu_tests_a.log python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread" \
 "600 gpurun_out/r3_bench_full_a.log python bench.py"
tail -c 6000 gpurun_out/r3_bench_full_a.log
