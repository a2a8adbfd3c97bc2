#!/bin/bash
# the rule "several rounds of blocks, or a rim above 4 %": every domain of the bench, tail rule on (default) and off
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python3 bench.py --no-strong-scaling --no-sweep --no-cpu-baseline "$@" > gpurun_out/r3_rule_$name.json 2>> gpurun_out/r3_rule.log || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/r3_rule_$name.json')); r=d['roofline']
print('$name', d['value'], d['ms_per_step'], 'fused launch', r['launch_ms'], 'tail', r.get('tail_cell_updates_per_launch'), r.get('first_step_loads'))"
}
run c1 X=1 --
run c1cav X=1 -- --preset cavity
run c2 X=1 -- --config 2
run c3 X=1 -- --config 3 --no-rehearsal-8
run c3_off LBM_FUSED_TAIL_RIM=0 -- --config 3 --no-rehearsal-8
run c512 X=1 -- --size 512 512 512
run c512_off LBM_FUSED_TAIL_RIM=0 -- --size 512 512 512
run c512cav X=1 -- --size 512 512 512 --preset cavity
run c512cav_off LBM_FUSED_TAIL_RIM=0 -- --size 512 512 512 --preset cavity
run s1 X=1 -- --rehearse-slab --slab-index 1 --config 3
run c1slab X=1 -- --rehearse-slab
