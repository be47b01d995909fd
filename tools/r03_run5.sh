#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do timeout -k 10 900 python -m pytest tests/test_gpu_ddp.py -q -s -k "dptnet" 2>&1 | grep -E "^E  |passed|failed|step_graph" | head -12; done
