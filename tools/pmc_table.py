#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one or more passes) into one row per kernel: average counter values per dispatch and
the derived figures used in DESIGN.md (VALU instructions per SIMD, share of wave time active / issue-stalled / waiting)."""
import collections
import csv
import glob
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("void ", "").replace("fqss::", "").split("(")[0]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        if not k.startswith("k_"):
            continue
        m = {c: sum(v) / len(v) for c, v in d.items()}
        n = len(next(iter(d.values())))
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        out = [f"{k:28s} n={n:4d} waves {m.get('SQ_WAVES', 0):7.0f}"]
        if "SQ_INSTS_VALU" in m:
            out.append(f"VALU/SIMD {m['SQ_INSTS_VALU'] / 1024:8.0f}  active {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} issue-stall {m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} "
                       f"wait {m.get('SQ_WAIT_ANY', 0) / wc:.2f}  gui-cycles {m.get('GRBM_GUI_ACTIVE', 0) / 8:8.0f}")
        if "SQ_INSTS_MFMA" in m:
            out.append(f"MFMA/SIMD {m['SQ_INSTS_MFMA'] / 1024:6.0f} mfma-busy-cyc/SIMD {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024:8.0f} SALU/SIMD {m.get('SQ_INSTS_SALU', 0) / 1024:6.0f} "
                       f"VMEM rd/wr per SIMD {m.get('SQ_INSTS_VMEM_RD', 0) / 1024:5.0f}/{m.get('SQ_INSTS_VMEM_WR', 0) / 1024:5.0f} LDS/SIMD {m.get('SQ_INSTS_LDS', 0) / 1024:6.0f} "
                       f"lds-conflict {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_ACTIVE_INST_LDS', 1), 1):.2f}")
        print("  ".join(out))


if __name__ == "__main__":
    main()
