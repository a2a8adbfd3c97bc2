#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=$R/gpurun_out/r3_curl_tiles2.log; : > $L
for v in "824 32" "824 64" "824 128" "442 32" "0 32"; do
  set -- $v
  echo "tile $1 zchunk $2" >> $L
  LBM_CURL_TILE=$1 LBM_CURL_ZCHUNK=$2 timeout -k 10 100 python3 $R/tools/curl_probe.py 512 256 256 2>/dev/null >> $L
  LBM_CURL_TILE=$1 LBM_CURL_ZCHUNK=$2 timeout -k 10 100 python3 $R/tools/curl_probe.py 400 100 100 2>/dev/null >> $L
  LBM_CURL_TILE=$1 LBM_CURL_ZCHUNK=$2 timeout -k 10 100 python3 $R/tools/curl_probe.py 768 384 384 2>/dev/null >> $L
done
cat $L
