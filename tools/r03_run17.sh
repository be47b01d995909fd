set -e
cd /root/repo
for st in 1 0 1 0; do
FQSS_UNALIGNED_VEC=$st python bench.py --workload cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r17.json 2> gpurun_out/r17.err || { tail -20 gpurun_out/r17.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r17.json').read().strip().splitlines()[-1]);print('cfg5 UNALIGNED_VEC=$st',d['ms_per_step'])"
done
