#!/bin/bash
# rocprofv3 passes for one bench.py configuration; results under gpurun_out/prof_<tag>/
# usage: tools/profile_round.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/prof_$tag
mkdir -p $o
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $o/stats -o stats --output-format csv -- python3 $R/bench.py --steps ${PR_STEPS:-51} --warmup ${PR_WARMUP:-5} --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $o/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps ${PR_PMC_STEPS:-21} --warmup ${PR_WARMUP:-5} --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $o/write -o write --output-format csv -- python3 $R/bench.py --steps ${PR_PMC_STEPS:-21} --warmup ${PR_WARMUP:-5} --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_write.log 2>&1 || exit 1
find $o -name "*.csv" | head -20
