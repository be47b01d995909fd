set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dptnet.py tests/test_gpu_sepformer.py -x -q -m gpu > gpurun_out/r8_tests.log 2>&1 || { tail -40 gpurun_out/r8_tests.log; exit 1; }
tail -2 gpurun_out/r8_tests.log
for w in cfg3 cfg4 cfg5; do
python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r8_$w.json 2> gpurun_out/r8_$w.err || { tail -20 gpurun_out/r8_$w.err; exit 1; }
python -c "import json,sys;d=json.loads(open('gpurun_out/r8_$w.json').read().strip().splitlines()[-1]);print('$w',d['ms_per_step'])"
done
