# SQ counters of every kernel of the cfg-2 step at its launch shape (tools/roofline_probe.py: the cases of fqss_amd/roofline_cases.py), two
# counter-only passes (never combined with trace domains).  GPU box, repo root:   bash tools/r06_pmc.sh > gpurun_out/r06_sq_counters.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_r06; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 tools/roofline_probe.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -- python3 tools/roofline_probe.py > $O/p2.log 2>&1
python3 tools/pmc_table.py $O/p1 $O/p2
rm -rf $O/p1 $O/p2
