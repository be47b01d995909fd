#!/bin/bash
# Regenerates the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root); summaries land in gpurun_out/.
# Counters are collected in their own passes (never combined with trace domains).
set -o pipefail
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --no-cpu-baseline --no-other-workloads > $O/bench_line.json 2> $O/bench.err
cp "$(find $O/bench -name '*kernel_stats.csv' | head -1)" $O/${R}_bench_kernel_stats.csv
rm -rf $O/bench
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-graph > $O/trace.log 2>&1
python3 tools/trace_summary.py "$(find $O/trace -name '*kernel_trace.csv' | head -1)" 2 > $O/${R}_step_eager_steady_state.txt
rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rl_stats -- python3 tools/roofline_probe.py > $O/rl_stats.log 2>&1
cp "$(find $O/rl_stats -name '*kernel_stats.csv' | head -1)" $O/${R}_roofline_probe_kernel_stats.csv
rm -rf $O/rl_stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rl_fetch -- python3 tools/roofline_probe.py > $O/rl_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rl_write -- python3 tools/roofline_probe.py > $O/rl_write.log 2>&1
python3 tools/roofline_probe.py --reduce "$(find $O/rl_fetch -name '*counter_collection.csv' | head -1)" "$(find $O/rl_write -name '*counter_collection.csv' | head -1)" \
    gpurun_out/rl_manifest.json $O/${R}_pmc_traffic.json > $O/${R}_pmc_traffic.txt
rm -rf $O/rl_fetch $O/rl_write
python3 bench.py > $O/${R}_bench_line.json 2> $O/bench2.err
ls -la $O
# the roofline kernels of cfg 3 / 4 / 5 under the same two counter passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rl2_fetch -- python3 tools/roofline_probe.py --set other > $O/rl2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rl2_write -- python3 tools/roofline_probe.py --set other > $O/rl2_write.log 2>&1
python3 tools/roofline_probe.py --reduce "$(find $O/rl2_fetch -name '*counter_collection.csv' | head -1)" "$(find $O/rl2_write -name '*counter_collection.csv' | head -1)" \
    gpurun_out/rl_manifest.json $O/${R}_pmc_traffic_cfg345.json > $O/${R}_pmc_traffic_cfg345.txt
rm -rf $O/rl2_fetch $O/rl2_write
