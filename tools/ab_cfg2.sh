#!/bin/bash
# A/B of library variants on the headline workload (GPU box, repo root): bash tools/ab_cfg2.sh <lib.so> [<lib.so> ...]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do for lib in "$@"; do
  FQSS_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib',d['ms_per_step'],[ (k['kernel'],k['launch_us']) for k in d['roofline_other_kernels'] if 'ewq_bwd' in k['kernel']])"
done; done
