set -e
cd /root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_gpu_tests.log 2>&1 || { tail -40 gpurun_out/full_gpu_tests.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
python bench.py --workload cfg5 > gpurun_out/prof_r03_cfg5_bench.json 2> gpurun_out/cfg5_bench.err
tail -c 900 gpurun_out/prof_r03_cfg5_bench.json
