#!/usr/bin/env python3
"""Rewrites the measured columns of docs/history/DESIGN_rounds_1-5.md §4's kernel table from profiles/r05_bench_line.json (every figure from ONE bench line).
    python tools/r05_design_table.py            (idempotent; rows are found by their kernel name)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o = json.loads(open(os.path.join(ROOT, "profiles/r05_bench_line.json")).read().strip().splitlines()[-1])
K = {k["kernel"]: k for k in o["roofline_other_kernels"]}
K[o["roofline"]["kernel"]] = o["roofline"]
mb = lambda k: f'{K[k]["algorithmic_bytes_per_launch"] / 1e6:.0f}'
tr = lambda k: f'{K[k]["traffic"] / K[k]["algorithmic_bytes_per_launch"]:.2f}' if K[k].get("traffic") else "n/a"
tm = lambda k: f'{K[k]["launch_us"]:.1f} ({K[k]["ms_per_step"]:.2f})'
lines = open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")


def setrow(prefix, cols):
    for i, l in enumerate(lines):
        if l.startswith(prefix):
            parts = l.split(" | ")
            for ci, v in cols.items():
                parts[ci] = v
            lines[i] = " | ".join(parts)
            return
    sys.exit("row not found: " + prefix)


r = K["k_dwq_bwd<3, GA, GB>"]
setrow("| `k_dwq_bwd<3, GA, GB>` (24)", {0: f"| `k_dwq_bwd<3, GA, GB>` (24) **← `roofline` (largest time per step: {r['ms_per_step']:.2f} ms)**", 3: tm("k_dwq_bwd<3, GA, GB>"),
                                        4: f"{r['frac']:.2f} of HBM (VALU-issue bound: 118 vector instructions per element, `profiles/r05_sq_counters.txt`); traffic {tr('k_dwq_bwd<3, GA, GB>')} |"})
setrow("| `k_qgemm<1>` (48)", {2: mb("k_qgemm<1>") + " avg (82 / 98)", 3: tm("k_qgemm<1>").replace(" (", " avg ("), 4: f"{K['k_qgemm<1>']['frac']:.2f} of HBM; traffic {tr('k_qgemm<1>')} |"})
setrow("| `k_tgemm2<1>` (24; round 4)", {3: tm("k_tgemm2<1>"), 4: f"floor max(50.3 GFLOP issued / 2.5 PF, 131 MB / 8 TB/s) = 20.1 us: {K['k_tgemm2<1>']['frac']:.2f} (bound: MFMA); traffic {tr('k_tgemm2<1>')} |"})
x1 = K["k_qgemm<0>"]["traffic_x1"] / K["k_qgemm<0>"]["algorithmic_bytes_per_launch"]
setrow("| `k_qgemm<0>` (48)", {2: mb("k_qgemm<0>") + " avg (86 / 74)", 3: tm("k_qgemm<0>").replace(" (", " avg ("),
                               4: f"{K['k_qgemm<0>']['frac']:.2f} of HBM; traffic {tr('k_qgemm<0>')} (u8 code rows are read in 64-B requests: {x1:.2f} under the x1 rule) |"})
setrow("| `k_qwgrad_group` (2; round 5)", {3: tm("k_qwgrad_group")})
setrow("| `k_gnq_bwd_apply<true>` (24)", {3: tm("k_gnq_bwd_apply<true>"), 4: f"{K['k_gnq_bwd_apply<true>']['frac']:.2f} of HBM: at the practical HBM rate; traffic {tr('k_gnq_bwd_apply<true>')} |"})
setrow("| `k_ewq_bwd<true>` (23 + 2 generic)", {2: mb("k_ewq_bwd"), 3: tm("k_ewq_bwd"), 4: f"{K['k_ewq_bwd']['frac']:.2f} of HBM; traffic {tr('k_ewq_bwd')} |"})
setrow("| `k_ewq_chain_bwd` (1; round 5)", {2: mb("k_ewq_chain_bwd"), 3: tm("k_ewq_chain_bwd"), 4: f"{K['k_ewq_chain_bwd']['frac']:.2f} of HBM; traffic {tr('k_ewq_chain_bwd')} |"})
t1 = "k_tgemm_k128" if "k_tgemm_k128" in K else "k_tgemm2<0>"
setrow("| `k_tgemm", {}) if False else None
for i, l in enumerate(lines):
    if l.startswith("| `k_tgemm2<0>` (24; round 4)") or l.startswith("| `k_tgemm_k128` (24; round 5)"):
        lines[i] = (f"| `k_tgemm_k128` (24; round 5) | teacher T1: 1×1 conv 128→512 + PReLU + GN statistics with the WEIGHTS IN REGISTERS (K = 128: 96 VGPRs per lane in A-fragment order), "
                    f"64-column activation tiles split into 55 KB of LDS, 96 MFMAs per wave and tile, 74 KB of LDS = two workgroups per CU (`k_tgemm2<0>`, 41.5 us, behind `FQSS_T1_K128=0`) "
                    f"| {mb(t1)} | {tm(t1)} | floor 10.2 us (bound: HBM; 10.1 us of MFMA issue): {K[t1]['frac']:.2f}; traffic {tr(t1)} |")
        break
else:
    sys.exit("T1 row not found")
setrow("| `k_tdw` (24)", {3: tm("k_tdw"), 4: f"{K['k_tdw']['frac']:.2f} of HBM; traffic {tr('k_tdw')} |"})
setrow("| `k_dwq_fwd<3>` (24)", {3: tm("k_dwq_fwd<3>"), 4: f"{K['k_dwq_fwd<3>']['frac']:.2f} of HBM (VALU-issue bound); traffic {tr('k_dwq_fwd<3>')} |"})
setrow("| `k_gnq_apply_t` (49; round 5)", {3: tm("k_gnq_apply_t"), 4: f"{K['k_gnq_apply_t']['frac']:.2f} of HBM (0.29 for the per-element form it replaces); traffic {tr('k_gnq_apply_t')} |"})
setrow("| `k_gnq_bwd_rows` (25)", {3: tm("k_gnq_bwd_rows"), 4: f"{K['k_gnq_bwd_rows']['frac']:.2f} of HBM; traffic {tr('k_gnq_bwd_rows')} |"})
a, b, c, d = K["k_ewq_fwd"], K["k_axpby"], K["k_actq_bwd"], K["k_gnq_bwd_rows+apply<false>"]
setrow("| `k_ewq_fwd` (2: the adds without a pair GEMM in front)", {3: f"{a['launch_us']:.1f} / {b['launch_us']:.1f} / {c['launch_us']:.1f} / {d['launch_us']:.1f}",
                                                                    4: f"{a['frac']:.2f} / {b['frac']:.2f} / {c['frac']:.2f} / {d['frac']:.2f} |"})
open(os.path.join(ROOT, "DESIGN.md"), "w").write("\n".join(lines))
print("bench line:", o["ms_per_step"], "ms/step;", o["value"], o["unit"], "; step traffic", o["step_traffic_GB"], "GB =", o["step_traffic_frac_of_hbm_peak"], "of HBM; survey", o["step_algorithmic_frac_of_hbm_peak"],
      "; isolated kernel sum", o["isolated_kernel_ms_sum"], "ms; missing PMC:", o["pmc_missing_kernels"])
