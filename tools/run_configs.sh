#!/bin/bash
# Runs bench.py on every BASELINE.json config that fits one GPU; one JSON line each in gpurun_out/configs.jsonl
set -o pipefail
out=gpurun_out/configs.jsonl
: > $out
timeout -k 10 300 python -u bench.py --config 1 >> $out 2> gpurun_out/configs.err || exit 1
timeout -k 10 300 python -u bench.py --config 1 --scheme aa --no-cpu-baseline >> $out 2>> gpurun_out/configs.err || exit 1
timeout -k 10 300 python -u bench.py --config 1 --preset wind_tunnel --no-cpu-baseline >> $out 2>> gpurun_out/configs.err || exit 1
timeout -k 10 300 python -u bench.py --config 2 --steps 100 >> $out 2>> gpurun_out/configs.err || exit 1
timeout -k 10 300 python -u bench.py --config 2 --steps 100 --scheme aa --no-cpu-baseline >> $out 2>> gpurun_out/configs.err || exit 1
timeout -k 10 400 python -u bench.py --config 3 --steps 30 --warmup 5 >> $out 2>> gpurun_out/configs.err || exit 1
timeout -k 10 400 python -u bench.py --config 3 --steps 30 --warmup 5 --scheme aa --no-cpu-baseline >> $out 2>> gpurun_out/configs.err || exit 1
timeout -k 10 600 python -u bench.py --config 4 --steps 20 --warmup 3 >> $out 2>> gpurun_out/configs.err || exit 1
