#!/bin/bash
# launch sequence of one eager single-stream step of a workload (GPU box, repo root): bash tools/seq_cfg.sh cfg4
set -o pipefail
w=${1:-cfg4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/seq_$w
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-teacher-ahead > $O/trace.log 2>&1
csv="$(find $O/trace -name '*kernel_trace.csv' | head -1)"
python3 tools/trace_seq.py "$csv" > $O/seq.txt
python3 tools/trace_summary.py "$csv" 2 > $O/table.txt
rm -rf $O/trace
wc -l $O/seq.txt
