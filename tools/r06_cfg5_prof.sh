cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
python3 tools/aten_probe.py > $O/cfg5_aten.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/cfg5_trace -- python3 bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/cfg5_trace.log 2>&1
csv="$(find $O/cfg5_trace -name '*kernel_trace.csv' | head -1)"
python3 tools/trace_summary.py "$csv" 2 > $O/cfg5_step_table.txt
rm -rf $O/cfg5_trace
head -60 $O/cfg5_step_table.txt
