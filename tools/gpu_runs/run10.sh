#!/bin/bash
bash tools/gpu_chain.sh \
 "900 gpurun_out/r3_fused_tests_e.log python -u -m pytest tests/test_fused_gpu.py tests/test_slab_pairs_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "300 gpurun_out/r3_c1_e.log python bench.py --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c1cav_e.log python bench.py --preset cavity --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c2_e.log python bench.py --config 2 --no-sweep --no-cpu-baseline --no-strong-scaling" \
 "300 gpurun_out/r3_c3_e.log python bench.py --config 3 --no-sweep --no-cpu-baseline --no-strong-scaling"
for f in c1 c1cav c2 c3; do grep -o '"value": [0-9.]*\|"launch_ms": [0-9.]*' gpurun_out/r3_${f}_e.log | tr '\n' ' '; echo; done
