#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/r03_trace_calls.py 2>&1 | tail -45
timeout -k 10 900 python -m pytest tests/test_gpu_kdstep_path.py tests/test_gpu_model.py -q -x 2>&1 | tail -4
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/b1.json 2>gpurun_out/b1.err; python -c "import json;d=json.load(open('gpurun_out/b1.json'));print('ms', d['ms_per_step'])"
