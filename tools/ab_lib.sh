#!/bin/bash
# A/B of library variants on one box (GPU box, repo root): bash tools/ab_lib.sh <lib.so> [<lib.so> ...]
# each: the row-GEMM probe timings (tools/kprobe.py) and the cfg 3 / 4 / 5 bench lines, two rounds
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  FQSS_LIB=$PWD/$lib python3 tools/kprobe.py time 2>&1 | grep -v "attn\|amdgpu.ids"
done > gpurun_out/ab_lib_kprobe.txt
for i in 1 2; do for lib in "$@"; do for w in cfg3 cfg4 cfg5; do
  FQSS_LIB=$PWD/$lib python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json;print('$lib $w',json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done; done; done
