#!/bin/bash
V=cfd_amd/csrc/variants
for n in 0 8 1 15 0 8; do
  timeout -k 10 200 env LBM_LIB=$V/liblbm_f2opt_$n.so python bench.py --config 2 --no-sweep --no-cpu-baseline --no-strong-scaling > gpurun_out/r3_opt2_$n.log 2>&1 || exit 1
  echo "opt $n: $(grep -o '"value": [0-9.]*\|"launch_ms": [0-9.]*' gpurun_out/r3_opt2_$n.log | tr '\n' ' ')"
done
