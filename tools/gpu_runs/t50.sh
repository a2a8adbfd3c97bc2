#!/bin/bash
# narrow first-step loads + the rim's first step in the tail (the narrow form's edge columns end early) against the default
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-strong-scaling --no-sweep --no-cpu-baseline "$@" > gpurun_out/r3_nt_$name.json 2>> gpurun_out/r3_nt.log || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/r3_nt_$name.json')); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], 'fused launch', r['launch_ms'], 'share of time', r.get('fused_share_of_time'), r.get('first_step_loads'))"
}
run c1_default X=1 --
run c1_narrow_tail LBM_FUSED_WIDE=0 LBM_FUSED_TAIL_PCT=0 --
run c1_wide_tail LBM_FUSED_WIDE=1 LBM_FUSED_TAIL_PCT=0 --
run c1_narrow LBM_FUSED_WIDE=0 --
run c1cav_default X=1 -- --preset cavity
run c1cav_narrow_tail LBM_FUSED_WIDE=0 LBM_FUSED_TAIL_PCT=0 -- --preset cavity
run c3_narrow_tail LBM_FUSED_WIDE=0 LBM_FUSED_TAIL_PCT=0 -- --config 3 --no-rehearsal-8
run c512_default X=1 -- --size 512 512 512
run c512_narrow_tail LBM_FUSED_WIDE=0 LBM_FUSED_TAIL_PCT=0 -- --size 512 512 512
