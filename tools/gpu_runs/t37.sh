#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -u -m pytest tests/test_cpp_mirror_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_t37.log 2>&1 || { tail -40 gpurun_out/r3_t37.log; exit 1; }
tail -2 gpurun_out/r3_t37.log
L=gpurun_out/r3_copy_engine.log; : > $L
for x in "" "--copy-engine"; do
  echo "lbm_slab_rccl $x" >> $L
  timeout -k 10 120 cfd_amd/csrc/lbm_slab_rccl --dims 256 256 256 --gpus 1 --steps 401 --autotune $x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 >> $L
  timeout -k 10 120 cfd_amd/csrc/lbm_slab_rccl --dims 1024 512 64 --gpus 1 --steps 201 --autotune $x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2 >> $L
done
cat $L
