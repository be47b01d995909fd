#!/usr/bin/env python3
"""Launches of one step (tools/seq_cfg.sh -> seq.txt) that leave most of the 256 CUs without a workgroup and still run long:
python tools/small_grids.py gpurun_out/seq_cfg5/seq.txt [max workgroups = 256] [min us = 30]"""
import re, sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+([\d.]+) ms\s+(.*?)\s+(\d+)x(\d+)x(\d+)\s+([\d.]+) us", l)
    if m:
        rows.append((float(m.group(7)), int(m.group(4)) * int(m.group(5)) * int(m.group(6)), m.group(3).strip(), f"{m.group(4)}x{m.group(5)}x{m.group(6)}"))
wmax = int(sys.argv[2]) if len(sys.argv) > 2 else 256
umin = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
small = sorted((r for r in rows if r[1] < wmax and r[0] > umin), reverse=True)
print(f"launches with < {wmax} workgroups and > {umin} us: {len(small)}, {sum(r[0] for r in small) / 1e3:.2f} ms of {sum(r[0] for r in rows) / 1e3:.2f} ms")
for r in small[:40]:
    print(f"{r[0]:8.1f} us  {r[1]:5d} wgs {r[3]:>12s}  {r[2][:90]}")
