# A/B of the AddQ forward in the pair GEMM's epilogue (GPU box, repo root): interleaved, two rounds
for r in 1 2; do for f in 1 0; do FQSS_FUSE_ADD_FWD=$f python bench.py --no-other-workloads --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys;o=json.loads(sys.stdin.readlines()[-1]);print('FQSS_FUSE_ADD_FWD=$f', o['ms_per_step'])"; done; done
