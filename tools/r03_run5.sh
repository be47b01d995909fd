#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_htdemucs.py -q -s -k "benchmarked_path" 2>&1 | tail -30 | cut -c1-300
