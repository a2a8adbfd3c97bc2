#!/bin/bash
R=$GRAFT_REPO_ROOT
f=$(grep -o "LBM_FUSED_WIDE=[01]" $R/gpurun_out/prof_c1/form.log) && export $f
echo "pinned: $f"
bash $R/tools/profile_counters.sh c1 > /dev/null || exit 1
bash $R/tools/profile_waits.sh c1 > /dev/null || exit 1
bash $R/tools/profile_l1.sh c1 > $R/gpurun_out/r3_l1_c1.txt || exit 1
cat $R/gpurun_out/r3_l1_c1.txt
