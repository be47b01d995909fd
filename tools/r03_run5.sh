#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "mha_prep" 2>&1 | tail -8
timeout -k 10 900 python -m pytest tests/test_gpu_dptnet.py tests/test_gpu_sepformer.py tests/test_gpu_kdstep_path.py -q -x 2>&1 | tail -5
ms() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], d['ms_per_step'])" $1 "$2"; }
for w in cfg3 cfg4; do
  FQSS_FUSE_MHA_PREP=0 timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${w}_base.json 2>gpurun_out/${w}_base.err; ms gpurun_out/${w}_base.json "$w unfused prep"
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${w}_ahead.json 2>gpurun_out/${w}_ahead.err; ms gpurun_out/${w}_ahead.json "$w fused prep"
done
