#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_export.py tests/test_gpu_infer.py -q -x 2>&1 | tail -15
