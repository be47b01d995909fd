#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ddp.py -q -s  > gpurun_out/ddp.log 2>&1
grep -n "AssertionError\|Error\|assert \|step_graph:\|^E   " gpurun_out/ddp.log | cut -c1-300 | head -60
