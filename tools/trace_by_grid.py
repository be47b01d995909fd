#!/usr/bin/env python3
"""Per-(kernel, grid) breakdown of the steady-state steps of a rocprofv3 --kernel-trace CSV of bench.py: which SHAPES of a kernel
the time goes to.  Usage: trace_by_grid.py <kernel_trace.csv> <nsteps> <name substring> [...]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2])
pats = sys.argv[3:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_step_end" in r["Kernel_Name"]]
seg = rows[ends[-1 - nst] + 1:ends[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("fqss::", "")
    if not any(p in n for p in pats):
        continue
    grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[(n, grid)][0] += 1
    agg[(n, grid)][1] += d
print(f"{'kernel':48s} {'workgroups':>16s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s}")
for (n, g), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{n[:48]:48s} {g:>16s} {c / nst:10.1f} {d / 1e6 / nst:9.3f} {d / c / 1e3:9.1f}")
