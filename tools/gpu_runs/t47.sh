#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-strong-scaling --no-sweep --no-cpu-baseline "$@" > gpurun_out/r3_tail2_$name.json 2>> gpurun_out/r3_tail2.log || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/r3_tail2_$name.json')); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], 'fused launch', r['launch_ms'], 'share of time', r.get('fused_share_of_time'), r.get('first_step_loads'))"
}
run c1 X=1 --
run c2 X=1 -- --config 2
run c2_side LBM_FUSED_TAIL_RIM=0 -- --config 2
run c3 X=1 -- --config 3 --no-rehearsal-8
run c3_side LBM_FUSED_TAIL_RIM=0 -- --config 3 --no-rehearsal-8
run def400 X=1 -- --size 400 100 100
run def400_side LBM_FUSED_TAIL_RIM=0 -- --size 400 100 100
run def400_seq LBM_FUSED_TAIL_RIM=0 LBM_FUSED_OVERLAP=0 -- --size 400 100 100
run c512 X=1 -- --size 512 512 512
run c512_seq LBM_FUSED_TAIL_RIM=0 -- --size 512 512 512
