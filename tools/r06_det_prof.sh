cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
export FQSS_DETERMINISTIC=1
rocprofv3 --kernel-trace --output-format csv -d $O/det_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-det-leg --no-graph > $O/det_trace.log 2>&1
python3 tools/trace_summary.py "$(find $O/det_trace -name '*kernel_trace.csv' | head -1)" 2 > $O/det_step_table.txt
rm -rf $O/det_trace
head -50 $O/det_step_table.txt
