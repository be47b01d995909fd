#!/bin/bash
# SQ counters of the two H = 128 forward kernels on the probe (separate passes; rocprofv3 takes a few counters per pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/lstm_pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $R/gpurun_out/lstm_pmc/p$i -o p$i --output-format csv -- python3 $R/tools/lstm_probe.py 3 > $R/gpurun_out/lstm_pmc/p$i.log 2>&1 || echo "pass $i failed" >> $R/gpurun_out/lstm_pmc/fail.txt
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
out=open(R+"/gpurun_out/lstm_pmc/summary.txt","w")
for f in sorted(glob.glob(R+"/gpurun_out/lstm_pmc/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0.0,0])
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "lstm_fwd" not in k: continue
        if row.get("Grid_Size") not in ("99328","49664"): pass
        key=(k.split("(")[0][-40:], row.get("Grid_Size"), row["Counter_Name"])
        acc[key][0]+=float(row["Counter_Value"]); acc[key][1]+=1
    for key,(v,n) in sorted(acc.items()):
        out.write("%-42s grid %-8s %-26s %14.0f per launch (%d launches)\n" % (key[0], key[1], key[2], v/n, n))
out.close()
PY
