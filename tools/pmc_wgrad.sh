cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcwg; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 tools/kbench.py qpw_bwd_w > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -- python3 tools/kbench.py qpw_bwd_w > $O/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/p3 -- python3 tools/kbench.py qpw_bwd_w > $O/p3.log 2>&1
python3 tools/pmc_table.py $O/p1 $O/p2
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcwg/p3/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qwgrad" in r["Kernel_Name"]:
            agg["wg"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print({c: round(sum(v)/len(v)) for c,v in d.items()}, len(next(iter(d.values()))))
PY
tail -3 $O/p3.log
