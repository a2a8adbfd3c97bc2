#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r3_wide3.log; : > $L
for w in 0 1; do
  echo "LBM_FUSED_WIDE=$w" >> $L
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 >> $L 2>&1
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 1024 512 512 --tunnel >> $L 2>&1
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 1024 512 512 --tunnel --car >> $L 2>&1
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" FB_STEPS=101 timeout -k 10 200 python3 tools/fused_bench.py 400 100 100 --tunnel >> $L 2>&1
  LBM_FUSED_WIDE=$w FB_VARIANTS="12,1,1" FB_STEPS=101 timeout -k 10 200 python3 tools/fused_bench.py 512 512 512 --tunnel >> $L 2>&1
  for idx in 0 4; do
    LBM_FUSED_WIDE=$w timeout -k 10 200 python3 bench.py --config 3 --rehearse-slab --slab-index $idx --slabs 8 --no-sweep --no-cpu-baseline --no-strong-scaling --steps 40 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab', $idx, d['ms_per_step'], d['roofline'].get('launch_ms'), d['roofline'].get('fused_share_of_cells'))" >> $L
  done
done
grep -v amdgpu.ids $L
