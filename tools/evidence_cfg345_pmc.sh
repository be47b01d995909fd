#!/bin/bash
# the cfg 3 / 4 / 5 roofline kernels under the two counter passes (the second half of tools/make_profiles.sh on its own) + the default bench line
set -o pipefail
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rl2_fetch -- python3 tools/roofline_probe.py --set other > $O/rl2_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rl2_write -- python3 tools/roofline_probe.py --set other > $O/rl2_write.log 2>&1 || exit 1
python3 tools/roofline_probe.py --reduce "$(find $O/rl2_fetch -name '*counter_collection.csv' | head -1)" "$(find $O/rl2_write -name '*counter_collection.csv' | head -1)" \
    gpurun_out/rl_manifest.json $O/${R}_pmc_traffic_cfg345.json > $O/${R}_pmc_traffic_cfg345.txt || exit 1
rm -rf $O/rl2_fetch $O/rl2_write
cp $O/${R}_pmc_traffic_cfg345.json profiles/
python3 bench.py > $O/${R}_bench_line.json 2> $O/bench2.err
