#!/bin/bash
# same-box A/B of whole workloads between library / source variants: bash tools/ab_workload.sh "<env assignments A>" "<env assignments B>" cfg4 [cfg3 ...]
# each variant's bench line twice, interleaved (the pool's boxes differ by ~4 %: only same-box pairs resolve 1 % changes)
A="$1"; B="$2"; shift 2
for w in "$@"; do for rep in 1 2; do for v in "$A" "$B"; do
  ms=$(env $v python3 bench.py --workload $w --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import sys,json;print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$w [$v] $ms ms"
done; done; done
