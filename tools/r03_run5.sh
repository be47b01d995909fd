#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_converge.py -x -q 2>&1 | tail -15
timeout -k 10 600 python -m pytest tests/test_gpu_kdstep_path.py -x -q -k "next_to_the_teacher" 2>&1 | tail -5
