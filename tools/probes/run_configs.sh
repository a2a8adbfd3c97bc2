#!/bin/bash
# Runs bench.py on every BASELINE.json config that fits one GPU; one JSON line each in gpurun_out/configs.jsonl
set -o pipefail
out=gpurun_out/configs.jsonl
: > $out
run() { t=$1; shift; timeout -k 10 $t python -u bench.py "$@" >> $out 2>> gpurun_out/configs.err || exit 1; }
run 300 --config 1
run 300 --config 1 --scheme aa --no-cpu-baseline
run 300 --config 1 --scheme slide --no-cpu-baseline
run 300 --config 1 --preset wind_tunnel --no-cpu-baseline
run 300 --config 2 --steps 100
run 300 --config 2 --steps 100 --scheme aa --no-cpu-baseline
run 300 --config 2 --steps 100 --scheme slide --no-cpu-baseline
run 400 --config 3 --steps 30 --warmup 5
run 400 --config 3 --steps 30 --warmup 5 --scheme aa --no-cpu-baseline
run 400 --config 3 --steps 30 --warmup 5 --scheme slide --no-cpu-baseline
run 600 --config 4 --steps 20 --warmup 3
run 600 --config 4 --steps 20 --warmup 3 --scheme slide --no-cpu-baseline
