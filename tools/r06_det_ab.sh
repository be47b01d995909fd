for rep in 1 2; do for v in "FQSS_DETERMINISTIC=0" "FQSS_DETERMINISTIC=1"; do
  ms=$(env $v python3 bench.py --no-cpu-baseline --no-other-workloads --no-det-leg --steps 60 2>/dev/null | python3 -c "import sys,json;print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "cfg2 [$v] $ms ms"
done; done
bash tools/ab_workload.sh "FQSS_DETERMINISTIC=0" "FQSS_DETERMINISTIC=1" cfg3 cfg4 cfg5
