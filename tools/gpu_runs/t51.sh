#!/bin/bash
# z-slabs: the rim's second step in the tail of the fused launch -- parity of the slab paths, then rehearsed slabs of configs[3]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_slab_pairs_gpu.py tests/test_two_stream_gpu.py tests/test_two_ranks_one_gpu.py tests/test_slab_gpu.py tests/test_cpp_mirror_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_t51.log 2>&1 || { tail -60 gpurun_out/r3_t51.log; exit 1; }
tail -2 gpurun_out/r3_t51.log
for tl in 0 1; do
  export LBM_FUSED_TAIL_RIM=$tl
  for sl in 0 1 2 3; do
    timeout -k 10 200 python3 bench.py --rehearse-slab --slab-index $sl --config 3 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_slabtail_s${sl}_$tl.json 2>> gpurun_out/r3_slabtail.log || exit 1
  done
  timeout -k 10 200 python3 bench.py --rehearse-slab --config 1 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_slabtail_c1_$tl.json 2>> gpurun_out/r3_slabtail.log || exit 1
done
python3 - <<'PY'
import json
for n in ("s0", "s1", "s2", "s3", "c1"):
    for tl in (0, 1):
        d = json.load(open(f"gpurun_out/r3_slabtail_{n}_{tl}.json"))
        r = d["roofline"]
        print(n, "tail", tl, d["value"], d["ms_per_step"], "fused launch", r["launch_ms"], r.get("tail_cell_updates_per_launch"), d.get("state_finite"))
PY
