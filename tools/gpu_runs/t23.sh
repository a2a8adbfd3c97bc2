#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --no-sweep --no-cpu-baseline > gpurun_out/r3_bench_ss.json 2> gpurun_out/r3_bench_ss.log
python3 -c "
import json; d=json.load(open('gpurun_out/r3_bench_ss.json')); ss=d['strong_scaling']
print(d['value'], ss['ms_per_step'], ss.get('projected_efficiency_8_even_split'), ss.get('projected_efficiency_8'), ss.get('projected_ms_per_step_8'))
print(ss.get('rehearsal_8_partitions_tried'))
print([(x['layers'], x['ms_per_step']) for x in ss['rehearsal_8']])"
