cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03; R=r03; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rl2_fetch -- python3 tools/roofline_probe.py --set other > $O/rl2_fetch.log 2>&1 || { tail -5 $O/rl2_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rl2_write -- python3 tools/roofline_probe.py --set other > $O/rl2_write.log 2>&1
python3 tools/roofline_probe.py --reduce "$(find $O/rl2_fetch -name '*counter_collection.csv' | head -1)" "$(find $O/rl2_write -name '*counter_collection.csv' | head -1)" \
    gpurun_out/rl_manifest.json $O/${R}_pmc_traffic_cfg345.json > $O/${R}_pmc_traffic_cfg345.txt
cat $O/${R}_pmc_traffic_cfg345.txt
