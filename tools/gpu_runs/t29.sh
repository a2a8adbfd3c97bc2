#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
LBM_FUSED_WIDE=1 timeout -k 10 900 python -u -m pytest tests/test_fused_gpu.py -x -q --timeout 300 --timeout-method thread -k "load_forms or empty_box or obstacles or 256" > gpurun_out/r3_t29.log 2>&1 || { tail -60 gpurun_out/r3_t29.log; exit 1; }
tail -3 gpurun_out/r3_t29.log
L=gpurun_out/r3_wide6.log; : > $L
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 --tunnel >> $L 2>&1
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 256 256 256 >> $L 2>&1
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" timeout -k 10 200 python3 tools/fused_bench.py 512 256 256 --tunnel --car >> $L 2>&1
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 1024 512 512 --tunnel --car >> $L 2>&1
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 1024 512 512 --tunnel >> $L 2>&1
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" FB_STEPS=41 timeout -k 10 200 python3 tools/fused_bench.py 512 512 512 --tunnel >> $L 2>&1
LBM_FUSED_DEBUG=1 FB_VARIANTS="12,1,1" FB_STEPS=101 timeout -k 10 200 python3 tools/fused_bench.py 400 100 100 --tunnel >> $L 2>&1
grep -v "amdgpu.ids\|fused\] [0-9]" $L
