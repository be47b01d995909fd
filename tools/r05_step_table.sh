# eager steady-state kernel table of the cfg-2 step (GPU box, repo root): bash tools/r05_step_table.sh [extra env assignments are inherited]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cfg2_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-other-workloads > gpurun_out/cfg2_trace.log 2>&1
csv="$(find gpurun_out/cfg2_trace -name '*kernel_trace.csv' | head -1)"
python3 tools/trace_summary.py "$csv" 2 > gpurun_out/${R:-r05}_step_eager_steady_state.txt
rm -rf gpurun_out/cfg2_trace
head -${N:-40} gpurun_out/${R:-r05}_step_eager_steady_state.txt
