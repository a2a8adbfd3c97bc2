#!/bin/bash
# the slab path on one rank (ring to itself), RCCL against peer copies: 256^3 and one 64-layer slab of configs[3]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r3_ipc_bench.log
for tr in rccl ipc; do
  timeout -k 10 200 python3 bench.py --rehearse-slab --config 1 --no-strong-scaling --no-sweep --no-cpu-baseline --transport $tr > gpurun_out/r3_ipc_c1_$tr.json 2>> gpurun_out/r3_ipc_bench.log || exit 1
  timeout -k 10 200 python3 bench.py --rehearse-slab --slab-index 0 --config 3 --no-strong-scaling --no-sweep --no-cpu-baseline --transport $tr > gpurun_out/r3_ipc_c3s0_$tr.json 2>> gpurun_out/r3_ipc_bench.log || exit 1
done
python3 - <<'PY'
import json
for n in ("c1", "c3s0"):
    for tr in ("rccl", "ipc"):
        d = json.load(open(f"gpurun_out/r3_ipc_{n}_{tr}.json"))
        s = d["slab_step"]
        print(n, tr, d["value"], d["ms_per_step"], s.get("ms_per_step_no_exchange"), s.get("exposed_comm_ms_per_step"), d.get("state_finite"))
PY

