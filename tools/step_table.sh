# per-step kernel table of a workload run eagerly: bash tools/step_table.sh cfg3 [name substrings for a per-grid breakdown ...]
# (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1; shift
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${w}_trace -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/${w}_trace.log 2>&1
csv="$(find gpurun_out/${w}_trace -name '*kernel_trace.csv' | head -1)"
python3 tools/trace_summary.py "$csv" 2 > gpurun_out/${R:-r03}_${w}_step_table.txt
if [ $# -gt 0 ]; then python3 tools/trace_by_grid.py "$csv" 2 "$@" > gpurun_out/${R:-r03}_${w}_by_grid.txt; fi
rm -rf gpurun_out/${w}_trace
cat gpurun_out/${R:-r03}_${w}_step_table.txt
