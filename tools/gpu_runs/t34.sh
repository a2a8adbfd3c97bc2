#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
L=gpurun_out/r3_p2p_channels.log; : > $L
for ch in 16 32 64; do
  echo "NCCL_MAX_P2P_NCHANNELS=$ch" >> $L
  if [ $ch != default ]; then export NCCL_MAX_P2P_NCHANNELS=$ch NCCL_MIN_P2P_NCHANNELS=$ch; fi
  PROBE_RCCL=1 timeout -k 5 120 python tools/slab_host_probe.py 2>&1 | grep "finished" | tail -1 >> $L
  PROBE_RCCL=1 timeout -k 5 120 python tools/slab_host_probe.py 1024 512 64 2>&1 | grep "finished" | tail -1 >> $L
done
cat $L
