set -e
cd /root/repo
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_htdemucs.py tests/test_gpu_fq.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r6_tests.log 2>&1 || { tail -30 gpurun_out/r6_tests.log; exit 1; }
tail -3 gpurun_out/r6_tests.log
python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r6_cfg5.json 2> gpurun_out/r6_cfg5.err || { tail -20 gpurun_out/r6_cfg5.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r6_cfg5.json').read().strip().splitlines()[-1]);print('cfg5',d['ms_per_step'])"
