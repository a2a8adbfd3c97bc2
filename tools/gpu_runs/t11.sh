#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
for w in 1 0; do
LBM_FUSED_WALLS=$w LBM_FUSED_DEBUG=4 timeout -k 10 100 python3 tools/chain_spread.py 256 256 256 2> gpurun_out/r3_blocks_w$w.txt > /dev/null
LBM_FUSED_WALLS=$w timeout -k 10 100 python3 tools/chain_spread.py 256 256 256 --blocks gpurun_out/r3_blocks_w$w.txt 2>/dev/null | tee gpurun_out/r3_spread_walls$w.log
done
