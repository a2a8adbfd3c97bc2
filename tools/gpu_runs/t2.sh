#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -q -k "curl or frame" --timeout 200 --timeout-method thread > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
timeout -k 10 120 python tools/curl_probe.py | tee gpurun_out/r3_curl_probe.log
timeout -k 10 120 python tools/curl_probe.py 512 256 256 | tee -a gpurun_out/r3_curl_probe.log
