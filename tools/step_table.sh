# per-step kernel table of a workload run eagerly: bash tools/step_table.sh cfg3 (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${w}_trace -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/${w}_trace.log 2>&1
python3 tools/trace_summary.py "$(find gpurun_out/${w}_trace -name '*kernel_trace.csv' | head -1)" 2 > gpurun_out/${R:-r03}_${w}_step_table.txt
rm -rf gpurun_out/${w}_trace
cat gpurun_out/${R:-r03}_${w}_step_table.txt
