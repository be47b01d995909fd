#!/usr/bin/env python3
"""The launches of the LAST steady-state step of a rocprofv3 --kernel-trace CSV of bench.py, in start order: name, workgroups, duration,
gap to the previous kernel's end.  Run the bench with --no-graph --no-teacher-ahead for a single-stream order.
Usage: trace_seq.py <kernel_trace.csv> [first] [count]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_step_end" in r["Kernel_Name"]]
seg = rows[ends[-2] + 1:ends[-1] + 1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(seg)
prev = None
t0 = int(seg[0]["Start_Timestamp"])
for i, r in enumerate(seg):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("fqss::", "")
    grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    prev = max(prev or 0, e)
    if first <= i < first + count:
        print(f"{i:5d} {(s - t0) / 1e6:8.3f} ms  {n[:70]:70s} {grid:>14s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}")
