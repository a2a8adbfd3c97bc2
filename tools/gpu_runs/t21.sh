#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 60 rocprofv3 --list-avail > $R/gpurun_out/r3_counters_avail.txt 2>&1
grep -o "TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*\|TD_[A-Z_a-z0-9]*" $R/gpurun_out/r3_counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000
