# counter passes over tools/kprobe.py (GPU box, repo root): bash tools/kprobe_pmc.sh <out-file>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/kprobe
rm -rf $O && mkdir -p $O
python3 tools/kprobe.py time > $O/time.txt 2>&1 || { cat $O/time.txt; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 tools/kprobe.py > $O/p1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $O/p2 -- python3 tools/kprobe.py > $O/p2.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p3 -- python3 tools/kprobe.py > $O/p3.log 2>&1
python3 tools/pmc_table.py $O/p1 $O/p2 $O/p3 > ${1:-gpurun_out/kprobe_pmc.txt} 2>&1
python3 - <<'PY' >> ${1:-gpurun_out/kprobe_pmc.txt}
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/kprobe/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("fqss::", "").split("(")[0]
        if k.startswith("k_"):
            agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\nper (kernel, grid): averages per dispatch")
for (k, g), d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k[:44]:44s} grid {g:>9s} gui-us {m.get('GRBM_GUI_ACTIVE',0)/8/2400:8.1f} waves {m.get('SQ_WAVES',0):6.0f} wavecyc/wave {wc/max(m.get('SQ_WAVES',1),1):8.0f} "
          f"wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:.2f} "
          f"valu/wave {m.get('SQ_INSTS_VALU',0)/max(m.get('SQ_WAVES',1),1):6.0f} mfma/wave {m.get('SQ_INSTS_MFMA',0)/max(m.get('SQ_WAVES',1),1):6.0f} mfma_busy/wavecyc {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/wc:.3f} "
          f"lds/wave {m.get('SQ_INSTS_LDS',0)/max(m.get('SQ_WAVES',1),1):6.0f} conflict/lds_active {m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):.2f} vmem_rd/wave {m.get('SQ_INSTS_VMEM_RD',0)/max(m.get('SQ_WAVES',1),1):5.0f}")
PY
cat $O/time.txt
