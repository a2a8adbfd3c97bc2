#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py tests/test_cpp_mirror_gpu.py -x -q -k "curl or frame or mirror or field" --timeout 200 --timeout-method thread > gpurun_out/r3_t6.log 2>&1 || { tail -40 gpurun_out/r3_t6.log; exit 1; }
tail -2 gpurun_out/r3_t6.log
L=$R/gpurun_out/r3_curl_final.log; : > $L
for sz in "256 256 256" "512 256 256" "400 100 100" "1024 512 512" "200 130 77"; do
  timeout -k 10 100 python3 $R/tools/curl_probe.py $sz 2>/dev/null >> $L
done
LBM_PROBE=1 LBM_CURL_ZCHUNK=16 timeout -k 10 100 python3 $R/tools/curl_probe.py 400 100 100 2>/dev/null >> $L
cat $L
