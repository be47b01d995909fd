#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sepformer.py tests/test_gpu_dptnet.py tests/test_gpu_htdemucs.py tests/test_gpu_kdstep_path.py -q -x 2>&1 | tail -8
ms() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], d['ms_per_step'])" $1 "$2"; }
FQSS_FUSE_ADDLN=0 timeout -k 10 300 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/cfg4_base.json 2>gpurun_out/cfg4_base.err; ms gpurun_out/cfg4_base.json "cfg4 unfused add+LN, float4 LN"
timeout -k 10 300 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/cfg4_ahead.json 2>gpurun_out/cfg4_ahead.err; ms gpurun_out/cfg4_ahead.json "cfg4 fused add+LN, float4 LN"
timeout -k 10 300 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/cfg3_ahead.json 2>gpurun_out/cfg3_ahead.err; ms gpurun_out/cfg3_ahead.json "cfg3"
