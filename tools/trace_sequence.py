"""Kernel sequence of the LAST training step in a rocprofv3 kernel trace (eager run): one line per launch outside the repeated TCN-block
kernels (those are summarised as a count), with durations.  python tools/trace_sequence.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_adam_clip" in r["Kernel_Name"]]      # a step ends with its clip + Adam launch
last = rows[ends[-2] + 1:ends[-1] + 1]
print(f"{len(last)} launches between the last two clip+Adam launches")
BLOCK = ("k_qgemm", "k_qwgrad", "k_gnq_", "k_dwq_", "k_ewq_", "k_tgemm<1>", "k_tgemm<0>", "k_tdw")
run = 0
t_run = 0.0
for r in last:
    name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("fqss::", "")
    short = name.split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if short.startswith(BLOCK):
        run += 1
        t_run += d
        continue
    if run:
        print(f"    ... {run} TCN-block launches, {t_run:.0f} us")
        run, t_run = 0, 0.0
    print(f"{d:8.1f} us  {short}  grid {r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')} wg {r.get('Workgroup_Size_X','')}")
if run:
    print(f"    ... {run} TCN-block launches, {t_run:.0f} us")
