#!/bin/bash
# the rim's first step in the tail of the fused launch: parity, then its effect per domain
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests/test_fused_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_t46.log 2>&1 || { tail -60 gpurun_out/r3_t46.log; exit 1; }
tail -2 gpurun_out/r3_t46.log
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-strong-scaling --no-sweep --no-cpu-baseline "$@" > gpurun_out/r3_tail_$name.json 2>> gpurun_out/r3_tail.log || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/r3_tail_$name.json')); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], 'fused launch', r['launch_ms'], 'share of time', r.get('fused_share_of_time'), r.get('first_step_loads'))"
}
run c1_off LBM_FUSED_TAIL_RIM=0 --
run c1_on LBM_FUSED_TAIL_RIM=1 --
run c1cav_off LBM_FUSED_TAIL_RIM=0 -- --preset cavity
run c1cav_on LBM_FUSED_TAIL_RIM=1 -- --preset cavity
run c2_off LBM_FUSED_TAIL_RIM=0 -- --config 2
run c2_on_noverlap LBM_FUSED_TAIL_RIM=1 LBM_FUSED_OVERLAP=0 -- --config 2
run c2_off_noverlap LBM_FUSED_TAIL_RIM=0 LBM_FUSED_OVERLAP=0 -- --config 2
run c3_off LBM_FUSED_TAIL_RIM=0 -- --config 3 --no-rehearsal-8
run c3_on_noverlap LBM_FUSED_TAIL_RIM=1 LBM_FUSED_OVERLAP=0 -- --config 3 --no-rehearsal-8
