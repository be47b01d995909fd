#!/bin/bash
# The other configurations' evidence (GPU box, repo root): bench line WITH the cpu_baseline leg, rocprofv3 kernel stats of the same
# command, the eager per-step kernel table.  bash tools/bench_cfg345.sh [r03]
set -o pipefail
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
mkdir -p $O
for w in cfg3 cfg4 cfg5; do
  python3 bench.py --workload $w > $O/${R}_${w}_bench.json 2> $O/${w}_bench.err || { tail -5 $O/${w}_bench.err; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${w}_prof -- python3 bench.py --workload $w --no-cpu-baseline > $O/${w}_prof.log 2>&1
  cp "$(find $O/${w}_prof -name '*kernel_stats.csv' | head -1)" $O/${R}_${w}_kernel_stats.csv
  rm -rf $O/${w}_prof
  rocprofv3 --kernel-trace --output-format csv -d $O/${w}_trace -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/${w}_trace.log 2>&1
  csv="$(find $O/${w}_trace -name '*kernel_trace.csv' | head -1)"
  python3 tools/trace_summary.py "$csv" 2 > $O/${R}_${w}_step_table.txt
  python3 tools/trace_by_grid.py "$csv" 2 k_frames k_qgemm k_gemm_x3 k_gemm_f32 k_attn > $O/${R}_${w}_by_grid.txt
  rm -rf $O/${w}_trace
  python3 -c "import json;d=json.loads(open('$O/${R}_${w}_bench.json').read().strip().splitlines()[-1]);print('$w',d['ms_per_step'],d['value'],d.get('cpu_baseline'))"
done
