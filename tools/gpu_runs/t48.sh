#!/bin/bash
# where the blocks of the 256^3 launch end, narrow and wide first-step loads
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in 0 1; do
  echo "== LBM_FUSED_WIDE=$w" >> gpurun_out/r3_spread_wide.log
  LBM_FUSED_WIDE=$w LBM_FUSED_DEBUG=4 timeout -k 10 120 python3 tools/chain_spread.py 256 256 256 --steps 7 2> gpurun_out/r3_spread_blocks_$w.err >> gpurun_out/r3_spread_wide.log || exit 1
  LBM_FUSED_WIDE=$w timeout -k 10 120 python3 tools/chain_spread.py 256 256 256 --steps 7 --blocks gpurun_out/r3_spread_blocks_$w.err >> gpurun_out/r3_spread_wide.log 2>&1 || exit 1
done
tail -n 40 gpurun_out/r3_spread_wide.log
