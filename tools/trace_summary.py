#!/usr/bin/env python3
"""Steady-state per-step kernel summary from a rocprofv3 --kernel-trace CSV of bench.py
(steps are delimited by the k_step_end marker kernel).  Usage: trace_summary.py <kernel_trace.csv> [nsteps]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_step_end" in r["Kernel_Name"]]
a, b = ends[-1 - nst], ends[-1]
seg = rows[a + 1:b + 1]
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6 / nst
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("fqss::", "")
    agg[n][0] += 1
    agg[n][1] += d
    busy += d
print(f"steady state over {nst} steps: wall {wall:.2f} ms/step, GPU busy {busy / 1e6 / nst:.2f} ms/step, {len(seg) / nst:.0f} launches/step")
print(f"{'kernel':72s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'share':>6s}")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n[:72]:72s} {c / nst:10.1f} {d / 1e6 / nst:9.3f} {d / c / 1e3:9.1f} {100 * d / busy:5.1f}%")
