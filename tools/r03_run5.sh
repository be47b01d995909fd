#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/bench_buckets.py cfg2 cfg4 cfg5 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/buckets.txt
