#!/usr/bin/env python3
"""k_gnq_apply (per-element quantizer arithmetic, rounds 1-4) vs k_gnq_apply_t (code-indexed LDS table, round 5) at the cfg-2 shape
(8 x 512 x 3999 codes), statistics from the producer, operands rotating over > 256 MiB, HIP events.
    python tools/r05_gnq_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K   # noqa: E402

B, C, M = 8, 512, 3999
dev = "cuda"
sets = [K.empty_codes((B, C, M), dev).random_(0, 256) for _ in range(24)]
lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
gm, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
xi = sets[0].to(torch.int64)
st = K.CodeStats(torch.stack([xi.sum(dim=(1, 2)), (xi * xi).sum(dim=(1, 2))], 1).reshape(-1).contiguous(), 1)
for v1 in ("1", "0", "1", "0"):
    os.environ["FQSS_GNQ_APPLY_V1"] = v1
    for i in range(10):
        K.gnq_fwd(sets[i % 24], lo, hi, gm, bt, 1e-8, lo, hi, False, stats=st)
    torch.cuda.synchronize()
    n = 96
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [K.gnq_fwd(sets[i % 24], lo, hi, gm, bt, 1e-8, lo, hi, False, stats=st) for i in range(n)]
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * n) * 1e3
    print(f"FQSS_GNQ_APPLY_V1={v1}: {us:.1f} us per launch (hipGraph of {n} launches), {2 * B * C * M / us / 1e6:.2f} TB/s of 1 B in + 1 B out")
    del g, keep
