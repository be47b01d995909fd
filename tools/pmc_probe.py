"""Launch a subset of the roofline cases a few times (for rocprofv3 --pmc SQ_* passes): python3 tools/pmc_probe.py k_tgemm,k_qwgrad"""
import sys, torch
sys.path.insert(0, ".")
from fqss_amd import roofline_cases as RC
dev = torch.device("cuda", 0)
cases = [c for c in RC.build(dev) if c["kernel"] in sys.argv[1].split(",")]
torch.cuda.synchronize()
for c in cases:
    for _ in range(5):
        c["fn"]()
    torch.cuda.synchronize()
