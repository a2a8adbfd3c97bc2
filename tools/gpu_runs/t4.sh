#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
LBM_CURL_TILE=822 timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -q -k "curl or frame" --timeout 200 --timeout-method thread > gpurun_out/r3_t4.log 2>&1 || { tail -40 gpurun_out/r3_t4.log; exit 1; }
tail -2 gpurun_out/r3_t4.log
LBM_CURL_TILE=442 LBM_CURL_ZCHUNK=16 timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -q -k "curl or frame" --timeout 200 --timeout-method thread > gpurun_out/r3_t4b.log 2>&1 || { tail -40 gpurun_out/r3_t4b.log; exit 1; }
tail -2 gpurun_out/r3_t4b.log
L=$R/gpurun_out/r3_curl_tiles.log; : > $L
for v in "442 32" "442 64" "822 32" "822 64" "842 16" "842 32" "842 64" "824 32" "824 64" "1622 16" "1622 32" "1622 64" "1614 32" "1614 64" "444 32"; do
  set -- $v
  echo "tile $1 zchunk $2" >> $L
  LBM_CURL_TILE=$1 LBM_CURL_ZCHUNK=$2 timeout -k 10 100 python3 $R/tools/curl_probe.py 2>/dev/null >> $L
done
cat $L
