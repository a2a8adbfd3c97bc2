#!/bin/bash
# SQ and TCC counter passes for one bench.py configuration (each block within its hardware limit: 8 SQ, 4 TCC)
# usage: tools/profile_counters.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/ctr_$tag
mkdir -p $o
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $o/sq -o sq --output-format csv -- python3 $R/bench.py --steps 11 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_sq.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $o/tcc -o tcc --output-format csv -- python3 $R/bench.py --steps 11 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_tcc.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $o/grbm -o grbm --output-format csv -- python3 $R/bench.py --steps 11 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-sweep --no-strong-scaling "$@" > $o/bench_grbm.log 2>&1 || exit 1
ls $o/*/*.csv
