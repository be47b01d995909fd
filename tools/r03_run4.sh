#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline"
ms() { python -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2], d['ms_per_step'])" $1 "$2"; }
$B --no-teacher-ahead > gpurun_out/b0.json 2>gpurun_out/b0.err; ms gpurun_out/b0.json "base (teacher beside own forward)"
$B > gpurun_out/b1.json 2>gpurun_out/b1.err; ms gpurun_out/b1.json "ahead"
FQSS_TGEMM_PAD_LDS=24576 $B > gpurun_out/b2.json 2>gpurun_out/b2.err; ms gpurun_out/b2.json "ahead + 1 teacher WG/CU"
FQSS_MAIN_PRIO=-1 $B > gpurun_out/b3.json 2>gpurun_out/b3.err; ms gpurun_out/b3.json "ahead + main stream high priority"
FQSS_MAIN_PRIO=-1 FQSS_TGEMM_PAD_LDS=24576 $B > gpurun_out/b4.json 2>gpurun_out/b4.err; ms gpurun_out/b4.json "ahead + prio + 1 WG/CU"
FQSS_TGEMM_PAD_LDS=24576 $B --no-teacher-ahead > gpurun_out/b5.json 2>gpurun_out/b5.err; ms gpurun_out/b5.json "base + 1 WG/CU"
python -c "import torch;print('prio range', torch.cuda.Stream.priority_range())"
