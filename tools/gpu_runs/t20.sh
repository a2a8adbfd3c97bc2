#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_slabc1 -o t --output-format csv -- python3 $R/bench.py --rehearse-slab --no-sweep --no-cpu-baseline --no-strong-scaling --steps 21 --warmup 5 --min-seconds 0 > $R/gpurun_out/r3_slab_c1_trace.log 2>&1
ls $R/gpurun_out/prof_slabc1
