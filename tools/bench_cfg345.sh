cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in cfg3 cfg4 cfg5; do
  python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${w}_line.json 2> gpurun_out/${w}_line.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${w}_prof -- python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/${w}_prof.log 2>&1
  cp "$(find gpurun_out/${w}_prof -name '*kernel_stats.csv' | head -1)" gpurun_out/r02_${w}_kernel_stats.csv
  rm -rf gpurun_out/${w}_prof
done
python3 - <<'PY'
import json,csv
for w in ("cfg3","cfg4","cfg5"):
    d=json.load(open(f"gpurun_out/{w}_line.json")); print(w, d["ms_per_step"], d["value"])
    rows=list(csv.DictReader(open(f"gpurun_out/r02_{w}_kernel_stats.csv")))
    for r in rows[:8]: print("   ", r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
