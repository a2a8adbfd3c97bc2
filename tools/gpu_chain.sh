#!/bin/bash
# gpu_chain.sh -- run GPU steps one after the other, each under its own time limit; a step that ends in
# a time-out, a signal, an abort or a segmentation fault (exit status >= 124) ends the chain: nothing more is
# started on the GPU after it.  Ordinary failures (a test that fails: status 1) do not stop the chain.
# usage: gpu_chain.sh "<seconds> <log> <command...>" ...
status=0
for step in "$@"; do
    set -- $step
    limit=$1; log=$2; shift 2
    echo "[chain] $* (limit ${limit}s) > $log"
    timeout -k 10 "$limit" "$@" > "$log" 2>&1
    rc=$?
    echo "[chain] exit $rc"
    tail -n 4 "$log"
    if [ $rc -ge 124 ]; then echo "[chain] stopping: step ended with status $rc"; exit $rc; fi
    [ $rc -ne 0 ] && status=$rc
done
exit $status
