#!/bin/bash
# TA / TCP (vector L1) counter passes for one bench.py configuration (2 TA, 4 TCP counters per pass)
# usage: tools/profile_l1.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/l1_$tag
mkdir -p $o
B="python3 $R/bench.py --steps 11 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -d $o/ta -o ta --output-format csv -- $B "$@" > $o/ta.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -d $o/tcp1 -o tcp1 --output-format csv -- $B "$@" > $o/tcp1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum -d $o/tcp2 -o tcp2 --output-format csv -- $B "$@" > $o/tcp2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum -d $o/td -o td --output-format csv -- $B "$@" > $o/td.log 2>&1 || exit 1
python3 - <<P
import csv, glob, collections
for f in sorted(glob.glob("$o/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fused2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:40s} {sum(v)/len(v):16.0f}  ({len(v)} dispatches)")
P
