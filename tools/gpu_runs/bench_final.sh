#!/bin/bash
# the default bench line, as the driver runs it, into gpurun_out/ (copied to profiles/r03_bench.json afterwards)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.log
rc=$?
tail -n 5 gpurun_out/r3_bench_final.log
python3 -c "
import json
d=json.load(open('gpurun_out/r3_bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['strong_scaling'].get('projected_efficiency_8'))
for c in d['configs']: print(c.get('mlups'), c['roofline']['frac'] if 'roofline' in c else c.get('error'), c['workload'][:60])
"
exit $rc
