#!/bin/bash
# rocprofv3 passes for one bench.py configuration; results under gpurun_out/prof_<tag>/
# usage: tools/profile_round.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/prof_$tag
mkdir -p $o
# k_fused2 times the two forms of its first-step loads on the first launch after a plan; the passes below must all run
# the form an unprofiled run settles on (and no trial launches among the dispatches they average over)
form=$(python3 $R/bench.py --steps 21 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" 2>/dev/null | python3 -c "import sys, json; print({'wide': 1, 'narrow': 0}.get(json.loads(sys.stdin.read())['roofline'].get('first_step_loads'), -1))")
[ "$form" != "-1" ] && export LBM_FUSED_WIDE=$form
echo "first-step loads of k_fused2 pinned to LBM_FUSED_WIDE=${LBM_FUSED_WIDE:-unset}" > $o/form.log
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $o/stats -o stats --output-format csv -- python3 $R/bench.py --steps ${PR_STEPS:-51} --warmup ${PR_WARMUP:-5} --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $o/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps ${PR_PMC_STEPS:-21} --warmup ${PR_WARMUP:-5} --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $o/write -o write --output-format csv -- python3 $R/bench.py --steps ${PR_PMC_STEPS:-21} --warmup ${PR_WARMUP:-5} --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_write.log 2>&1 || exit 1
find $o -name "*.csv" | head -20
