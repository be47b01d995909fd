set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_sepformer.py tests/test_gpu_dptnet.py -x -q -m gpu > gpurun_out/r16_tests.log 2>&1 || { tail -40 gpurun_out/r16_tests.log; exit 1; }
tail -2 gpurun_out/r16_tests.log
for st in 1 0 1 0; do
FQSS_FUSE_POSTRELU=$st python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r16.json 2> gpurun_out/r16.err || { tail -20 gpurun_out/r16.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r16.json').read().strip().splitlines()[-1]);print('cfg4 FUSE_POSTRELU=$st',d['ms_per_step'])"
done
