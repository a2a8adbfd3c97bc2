#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
o=$R/gpurun_out/prof_slab_ipc_wb
mkdir -p $o
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $o/stats -o stats --output-format csv -- python3 $R/bench.py --rehearse-slab --config 1 --steps 201 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling --transport ipc > $o/bench.json 2> $o/bench.log || exit 1
f=$(find $o/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if float(r["Percentage"]) < 0.2: break
    print(f'{n[:75]:75s} {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.1f} min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
PY
