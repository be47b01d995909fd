set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_htdemucs.py tests/test_gpu_kdstep_path.py -x -q -m gpu > gpurun_out/r11_tests.log 2>&1 || { tail -40 gpurun_out/r11_tests.log; exit 1; }
tail -2 gpurun_out/r11_tests.log
python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r11_cfg5.json 2> gpurun_out/r11_cfg5.err || { tail -20 gpurun_out/r11_cfg5.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r11_cfg5.json').read().strip().splitlines()[-1]);print('cfg5',d['ms_per_step'])"
FQSS_ATTN_CODED=0 python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r11_cfg5b.json 2> gpurun_out/r11_cfg5b.err
python -c "import json;d=json.loads(open('gpurun_out/r11_cfg5b.json').read().strip().splitlines()[-1]);print('cfg5 float-operand attention',d['ms_per_step'])"
