set -e
cd /root/repo
for rep in 1 2; do for st in 1 0; do for w in cfg4 cfg5; do
FQSS_X3_STAGED=$st python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r14.json 2> gpurun_out/r14.err || { tail -20 gpurun_out/r14.err; exit 1; }
python -c "import json;d=json.loads(open('gpurun_out/r14.json').read().strip().splitlines()[-1]);print('$w staged=$st',d['ms_per_step'])"
done; done; done
