#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r3_curl_variants.log; : > $L
for v in "32 4" "64 4" "128 4" "256 4" "32 2" "64 2" "32 8" "64 8" "16 4" "64 1"; do
  set -- $v
  echo "zchunk $1 U $2" >> $L
  LBM_CURL_ZCHUNK=$1 LBM_CURL_U=$2 timeout -k 10 100 python3 $R/tools/curl_probe.py >> $L 2>&1
done
cat $L
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_curl/fetch -o fetch --output-format csv -- python3 $R/tools/curl_probe.py > $R/gpurun_out/r3_curl_fetch.log 2>&1
python3 - <<'P'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_curl/fetch/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, len(v), "FETCH_SIZE avg", sum(v) / len(v), "-> MB (x1024 x2)", sum(v) / len(v) * 2048 / 1e6)
P
