#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
LBM_FUSED_DEBUG=4 timeout -k 10 100 python3 tools/chain_spread.py 256 256 256 --steps 41 2> gpurun_out/r3_blocks.txt > /dev/null
timeout -k 10 100 python3 tools/chain_spread.py 256 256 256 --steps 41 --blocks gpurun_out/r3_blocks.txt 2>/dev/null | tee gpurun_out/r3_spread_bx.log
