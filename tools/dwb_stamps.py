"""Where a workgroup of k_dwq_bwd<3, GA, GB> spends its life (library built with -DFQSS_DWB_STAMPS: FQSS_LIB=...):
wave 0's s_memtime deltas between phase boundaries, averaged over the 4096 workgroups of one launch, plus the launch's span."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import roofline_cases as RC  # noqa: E402

dev = torch.device("cuda", 0)
case = [c for c in RC.build(dev) if c["kernel"].startswith("k_dwq_bwd")][0]
import fqss_amd.kernels as K  # noqa: E402
real = K.dwq_bwd
grabbed = {}


def spy(*a, **k):
    out = real(*a, **k)
    grabbed["ws"] = k["before"]["ws"]
    return out


K.dwq_bwd = spy
for i in range(3):
    case["fn"](i)
torch.cuda.synchronize()
w = grabbed["ws"].view(torch.int64).cpu().numpy().reshape(-1, 2)
d = np.zeros((w.shape[0], 6))
for k in range(6):
    src = w[:, 0] if k < 4 else w[:, 1]
    d[:, k] = ((src >> (16 * (k % 4))) & 0xffff) * 16.0
start = ((w[:, 1] >> 32) & 0xffffffff).astype(np.float64) * 256.0
names = ["prologue", "phase 1 (loads + arithmetic)", "barrier", "phase 2", "GB reduction", "final reduction + atomics"]
ghz = 2.3
print(f"{w.shape[0]} workgroups; cycles -> us at {ghz} GHz")
for k, n in enumerate(names):
    print(f"  {n:30s} mean {d[:, k].mean() / ghz / 1e3:6.2f} us   p10 {np.percentile(d[:, k], 10) / ghz / 1e3:6.2f}   p90 {np.percentile(d[:, k], 90) / ghz / 1e3:6.2f}")
life = d.sum(1)
print(f"  workgroup life                 mean {life.mean() / ghz / 1e3:6.2f} us; launch span (first start .. last start) {(start.max() - start.min()) / ghz / 1e3:6.1f} us")
print(f"  time per launch: {min(RC.time_case(case) for _ in range(3)) * 1e3:.2f} us")
