#!/bin/bash
# round-3 profiles: kernel-trace stats + FETCH/WRITE passes per configuration (tools/profile_round.sh), then counters of the headline
set -o pipefail
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh c1 || exit 1
bash $R/tools/profile_round.sh c1cav --preset cavity || exit 1
bash $R/tools/profile_round.sh c2 --config 2 || exit 1
PR_STEPS=21 PR_PMC_STEPS=11 bash $R/tools/profile_round.sh c3 --config 3 || exit 1
PR_STEPS=9 PR_PMC_STEPS=5 PR_WARMUP=2 bash $R/tools/profile_round.sh c4 --config 4 || exit 1
PR_STEPS=9 PR_PMC_STEPS=5 PR_WARMUP=2 bash $R/tools/profile_round.sh c4aa --config 4 --scheme aa || exit 1
# the counter passes of the headline run the form of the first-step loads its profile passes were pinned to
f=$(grep -o "LBM_FUSED_WIDE=[01]" $R/gpurun_out/prof_c1/form.log) && export $f
bash $R/tools/profile_counters.sh c1 || exit 1
bash $R/tools/profile_waits.sh c1 || exit 1
bash $R/tools/profile_l1.sh c1 > $R/gpurun_out/r3_l1_c1.txt || exit 1
echo done
