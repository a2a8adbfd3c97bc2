#!/bin/bash
V=cfd_amd/csrc/variants
timeout -k 10 300 env LBM_LIB=$V/liblbm_f2opt_15.so python -u -m pytest tests/test_fused_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_fused_tests_f.log 2>&1; echo "tests opt15: $?"; tail -3 gpurun_out/r3_fused_tests_f.log
for n in 0 1 2 4 8 12 15 0 15; do
  timeout -k 10 200 env LBM_LIB=$V/liblbm_f2opt_$n.so python bench.py --no-sweep --no-cpu-baseline --no-strong-scaling > gpurun_out/r3_opt_$n.log 2>&1 || exit 1
  echo "opt $n: $(grep -o '"value": [0-9.]*\|"launch_ms": [0-9.]*' gpurun_out/r3_opt_$n.log | tr '\n' ' ')"
done
