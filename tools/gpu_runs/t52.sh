#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export LBM_FUSED_TAIL_PCT=0
timeout -k 10 600 python3 -u -m pytest tests/test_slab_pairs_gpu.py tests/test_two_ranks_one_gpu.py tests/test_slab_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_t52.log 2>&1 || { tail -60 gpurun_out/r3_t52.log; exit 1; }
tail -2 gpurun_out/r3_t52.log
for sl in 0 1 2 3; do
  timeout -k 10 200 python3 bench.py --rehearse-slab --slab-index $sl --config 3 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_slabtail_s${sl}_f.json 2>> gpurun_out/r3_slabtail.log || exit 1
done
timeout -k 10 200 python3 bench.py --rehearse-slab --config 1 --no-strong-scaling --no-sweep --no-cpu-baseline > gpurun_out/r3_slabtail_c1_f.json 2>> gpurun_out/r3_slabtail.log || exit 1
python3 - <<'PY'
import json
for n in ("s0", "s1", "s2", "s3", "c1"):
    d = json.load(open(f"gpurun_out/r3_slabtail_{n}_f.json"))
    r = d["roofline"]
    print(n, "tail forced", d["value"], d["ms_per_step"], "fused launch", r["launch_ms"], r.get("tail_cell_updates_per_launch"), d.get("state_finite"))
PY
