for h in after before both; do FQSS_GN_HAND=$h python bench.py --no-other-workloads --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.readlines()[-1]);print('$h', o['ms_per_step'])"; done
FQSS_FUSE_GN_BWD_DW=0 python bench.py --no-other-workloads --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.readlines()[-1]);print('none', o['ms_per_step'])"
