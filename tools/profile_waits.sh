#!/bin/bash
# where the waves of the dominant kernel wait: SQ wait / LDS counters of one bench.py configuration (one pass of 8 SQ counters)
# usage: tools/profile_waits.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/wait_$tag
mkdir -p $o
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY -d $o/sq -o sq --output-format csv -- python3 $R/bench.py --steps 11 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_sq.log 2>&1 || exit 1
ls $o/*/*.csv
