#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 bench.py --rehearse-slab --no-sweep --no-cpu-baseline --no-strong-scaling > gpurun_out/r3_slab_c1.json 2> gpurun_out/r3_slab_c1.log
python3 -c "
import json; d=json.load(open('gpurun_out/r3_slab_c1.json')); print(d['value'], d['ms_per_step'], d['config']['workload']); print(d['slab_step']); print({k:d['roofline'][k] for k in ('launch_ms','launches_per_step','fused_share_of_cells','fused_share_of_time')})"
