cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gtrace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/gtrace.log 2>&1
python3 tools/graph_gaps.py "$(find gpurun_out/gtrace -name '*kernel_trace.csv' | head -1)" 5
rm -rf gpurun_out/gtrace
