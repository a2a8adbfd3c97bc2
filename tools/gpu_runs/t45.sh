#!/bin/bash
# the rim of the undecomposed path beside the fused kernel, in workgroups of four waves and of one
# (LBM_RIM_WAVE_BLOCKS existed in a probe build only, profiles/r03_tuning.md section 12: without it the "wave" lines repeat the others)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-strong-scaling --no-sweep --no-cpu-baseline "$@" > gpurun_out/r3_rimwb_$name.json 2>> gpurun_out/r3_rimwb.log || exit 1
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3_rimwb_$name.json')); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], 'fused launch', r['launch_ms'], 'share of time', r.get('fused_share_of_time'), r.get('first_step_loads'))"
}
run c1_default X=1 --
run c1_overlap LBM_FUSED_OVERLAP_PCT=0 --
run c1_overlap_wave LBM_FUSED_OVERLAP_PCT=0 LBM_RIM_WAVE_BLOCKS=1 --
run c1cav_default X=1 -- --preset cavity
run c1cav_overlap_wave LBM_FUSED_OVERLAP_PCT=0 LBM_RIM_WAVE_BLOCKS=1 -- --preset cavity
run c2_default X=1 -- --config 2
run c2_wave LBM_RIM_WAVE_BLOCKS=1 -- --config 2
run c3_default X=1 -- --config 3 --no-rehearsal-8
run c3_wave LBM_RIM_WAVE_BLOCKS=1 -- --config 3 --no-rehearsal-8
