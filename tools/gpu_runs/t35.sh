#!/bin/bash
# rehearsal slabs that report a non-finite state: the reference's parameters against tame ones
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python3 -u tools/rehearsal_state_probe.py --slab 2 3 > gpurun_out/r3_rehearsal_state.log 2>&1
rc=$?
tail -n 12 gpurun_out/r3_rehearsal_state.log
exit $rc
