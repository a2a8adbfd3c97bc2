#!/bin/bash
# the boundary layers' second step of a pair in one-wave workgroups: parity of the slab paths, then its effect
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_slab_pairs_gpu.py tests/test_two_stream_gpu.py tests/test_two_ranks_one_gpu.py tests/test_slab_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_t43.log 2>&1 || { tail -60 gpurun_out/r3_t43.log; exit 1; }
tail -2 gpurun_out/r3_t43.log
for wb in 0 1; do
  export LBM_PAIR_WAVE_BLOCKS=$wb
  timeout -k 10 200 python3 bench.py --rehearse-slab --config 1 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_wb_c1_$wb.json 2>> gpurun_out/r3_wb.log || exit 1
  timeout -k 10 200 python3 bench.py --rehearse-slab --slab-index 0 --config 3 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_wb_c3s0_$wb.json 2>> gpurun_out/r3_wb.log || exit 1
  timeout -k 10 200 python3 bench.py --rehearse-slab --slab-index 3 --config 3 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_wb_c3s3_$wb.json 2>> gpurun_out/r3_wb.log || exit 1
done
python3 - <<'PY'
import json
for n in ("c1", "c3s0", "c3s3"):
    for wb in (0, 1):
        d = json.load(open(f"gpurun_out/r3_wb_{n}_{wb}.json"))
        print(n, "wave blocks", wb, d["value"], d["ms_per_step"], d["slab_step"].get("ms_per_step_no_exchange"), d.get("state_finite"))
PY
