#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
L=$R/gpurun_out/r3_nt_fields.log; : > $L
for lib in "" cfd_amd/csrc/variants/liblbm_nt.so "" cfd_amd/csrc/variants/liblbm_nt.so; do
  echo "lib=$lib" >> $L
  LBM_LIB=${lib:+$R/$lib} timeout -k 10 100 python3 $R/tools/curl_probe.py 2>/dev/null >> $L
done
cat $L
