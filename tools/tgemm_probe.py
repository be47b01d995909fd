"""The teacher GEMM shapes of the cfg-2 step (fqss_amd/roofline_cases.py: cold operands rotating over > 256 MiB, launches replayed from a
hipGraph, HIP events on the launch stream) -- for A/B of library variants through FQSS_LIB.  python tools/tgemm_probe.py [kernel-substring]"""
import sys
import torch
sys.path.insert(0, ".")
from fqss_amd import roofline_cases as RC

def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else "k_tgemm"
    cases = [c for c in RC.build(torch.device("cuda", 0)) if pat in c["kernel"]]
    for rep in range(2):
        for c in cases:
            ms = RC.time_case(c)
            print("%-12s %-78s %7.1f us" % (c["kernel"], c["label"][:78], ms * 1e3), flush=True)

if __name__ == "__main__":
    main()
