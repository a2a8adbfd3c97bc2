#!/bin/bash
B="python bench.py --config 3 --size 1024 512 64 --rehearse-slab --no-sweep --no-cpu-baseline --no-strong-scaling"
bash tools/gpu_chain.sh \
 "600 gpurun_out/r3_tests3.log python -u -m pytest tests/test_slab_pairs_gpu.py tests/test_two_stream_gpu.py tests/test_slab_gpu.py -x -q --timeout 300 --timeout-method thread" \
 "300 gpurun_out/r3_reh_a.log $B" \
 "300 gpurun_out/r3_reh_b.log env LBM_PAIR_RIM_STREAM=0 $B" \
 "300 gpurun_out/r3_reh_c.log env LBM_SIDE_PRIO=0 $B" \
 "300 gpurun_out/r3_reh_d.log env LBM_SIDE_PRIO=0 LBM_PAIR_RIM_STREAM=0 $B"
for f in a b c d; do echo "== $f"; grep -o '"ms_per_step": [0-9.]*\|"launch_ms": [0-9.]*\|"ms_per_step_no_exchange": [0-9.]*' gpurun_out/r3_reh_$f.log | tr '\n' ' '; echo; done
