#!/bin/bash
# kernel trace of the slab path at 256^3 (one rank, ring to itself) on both transports
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tr in rccl ipc; do
  o=$R/gpurun_out/prof_slab_$tr
  mkdir -p $o
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $o/stats -o stats --output-format csv -- python3 $R/bench.py --rehearse-slab --config 1 --steps 201 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling --transport $tr > $o/bench.json 2> $o/bench.log || exit 1
  f=$(find $o/stats -name "*kernel_stats.csv" | head -1)
  echo "== $tr: $f"
  head -n 14 "$f" | cut -c1-200
done
