for st in 0 2000 4000 8000 12000 16000; do FQSS_T2_STAGGER=$st python bench.py --no-other-workloads --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.readlines()[-1]);print('stagger $st', o['ms_per_step'], [(k['kernel'],k['launch_us']) for k in o['roofline_other_kernels'] if 'tgemm' in k['kernel']])"; done
