#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
PROBE_RCCL=1 timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_slabrccl -o t --output-format csv -- python3 $R/tools/slab_host_probe.py 256 256 256 41 > $R/gpurun_out/r3_slabrccl.log 2>&1
ls $R/gpurun_out/prof_slabrccl
