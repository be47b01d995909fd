# A/B of the skip sum's AddQ chain backward as one launch (GPU box, repo root): interleaved, two rounds
for r in 1 2; do for f in 1 0; do FQSS_FUSE_ADD_CHAIN=$f python bench.py --no-other-workloads --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.readlines()[-1]);print('FQSS_FUSE_ADD_CHAIN=$f', o['ms_per_step'])"; done; done
