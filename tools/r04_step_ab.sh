#!/bin/bash
# cfg-2 step, product library vs variants / env settings, interleaved on one box: bash tools/r04_step_ab.sh "<env assignments>" ...
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-workloads --steps 40 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('[$cfg] ms_per_step',d['ms_per_step'],[ (k['kernel'],k['launch_us']) for k in [d['roofline']]+d['roofline_other_kernels'] if 'tgemm' in k['kernel']])" | tee -a gpurun_out/r04_step_ab.txt || exit 1
done
done
