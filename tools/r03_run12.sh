set -e
cd /root/repo
for mc in 64 0 1024; do
FQSS_CONV_IMPLICIT_MAX_CO=$mc python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r12_cfg5.json 2> gpurun_out/r12_cfg5.err || { tail -20 gpurun_out/r12_cfg5.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r12_cfg5.json').read().strip().splitlines()[-1]);print('cfg5 implicit for Co <= $mc:',d['ms_per_step'])"
done
