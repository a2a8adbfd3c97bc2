#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -u tools/small_batch_probe.py > gpurun_out/r3_small_batches.log 2>&1
rc=$?
tail -n 12 gpurun_out/r3_small_batches.log
exit $rc
