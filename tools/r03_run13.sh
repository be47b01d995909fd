set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fq.py -x -q -m gpu > gpurun_out/r13_tests.log 2>&1 || { tail -40 gpurun_out/r13_tests.log; exit 1; }
tail -2 gpurun_out/r13_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_htdemucs.py tests/test_gpu_dptnet.py -x -q -m gpu > gpurun_out/r13_hd.log 2>&1 || { tail -40 gpurun_out/r13_hd.log; exit 1; }
tail -2 gpurun_out/r13_hd.log
python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r13_cfg5.json 2> gpurun_out/r13_cfg5.err || { tail -20 gpurun_out/r13_cfg5.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r13_cfg5.json').read().strip().splitlines()[-1]);print('cfg5',d['ms_per_step'])"
true
true
