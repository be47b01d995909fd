set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_long" > gpurun_out/r7_tests.log 2>&1 || { tail -40 gpurun_out/r7_tests.log; exit 1; }
tail -2 gpurun_out/r7_tests.log
echo "--- split-bf16" > gpurun_out/r7_probe.txt
timeout -k 10 300 python tools/attn_long_probe.py >> gpurun_out/r7_probe.txt 2>&1


cat gpurun_out/r7_probe.txt
true
true
python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r7_cfg5.json 2> gpurun_out/r7_cfg5.err || { tail -20 gpurun_out/r7_cfg5.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r7_cfg5.json').read().strip().splitlines()[-1]);print('cfg5',d['ms_per_step'])"
