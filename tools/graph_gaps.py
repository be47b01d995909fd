#!/usr/bin/env python3
"""Idle time inside the hipGraph replay of the cfg-2 step from a rocprofv3 --kernel-trace CSV of `bench.py` (graph mode): the last
replays are located by the k_step_end marker; per step: wall, union of kernel intervals (GPU busy, overlap counted once), the idle
remainder, and the distribution of the gaps between consecutive kernels.  Usage: graph_gaps.py <kernel_trace.csv> [nsteps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_step_end" in r["Kernel_Name"]]
a, b = ends[-1 - nst], ends[-1]
seg = rows[a + 1:b + 1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
wall = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ssum = sum(e - s for s, e in iv)
gaps.sort()
n = len(gaps)
print(f"{nst} replayed steps: wall {wall / 1e6 / nst:.3f} ms/step, kernel time summed {ssum / 1e6 / nst:.3f}, union (busy) {busy / 1e6 / nst:.3f}, "
      f"idle {(wall - busy) / 1e6 / nst:.3f} ms/step in {n / nst:.0f} gaps/step ({len(seg) / nst:.0f} launches/step)")
if n:
    print(f"gap us: median {gaps[n // 2] / 1e3:.2f}, mean {sum(gaps) / n / 1e3:.2f}, p90 {gaps[int(0.9 * n)] / 1e3:.2f}, max {gaps[-1] / 1e3:.1f}")
