#!/usr/bin/env python3
"""Row-count sensitivity of the row GEMMs (workgroup-count quantisation on 256 CUs): python tools/tail_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from fqss_amd import kernels as K  # noqa: E402


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = "cuda"
for Ci, Co in ((256, 1024), (1024, 256), (256, 256), (256, 768)):
    for R in (4096, 8000, 8192, 8500, 8704, 12288, 16384):
        x, w, b = torch.randn(R, Ci, device=dev), torch.randn(Co, Ci, device=dev), torch.randn(Co, device=dev)
        g, gw = torch.randn(R, Co, device=dev), torch.zeros(Co, Ci, device=dev)
        xc = torch.randint(0, 256, (R, Ci), device=dev, dtype=torch.uint8)
        lo, hi = torch.tensor([-1.0], device=dev), torch.tensor([1.0], device=dev)
        wc = K.WCodes()
        wc.Ci, wc.Co = Ci, Co
        wc.idx = torch.randint(-127, 128, (Co, Ci), device=dev, dtype=torch.int8)
        wc.dw = torch.rand(Co, device=dev) * 0.01
        print(f"{Ci:5d}->{Co:5d} R={R:6d}  fwd {t(lambda: K.rowlin_fwd(x, w, b)):7.1f}  bwd_x {t(lambda: K.rowlin_bwd_x(g, w)):7.1f}  bwd_x coded {t(lambda: K.qrow_bwd_x(g, wc)):7.1f}"
              f"  bwd_w {t(lambda: K.rowlin_bwd_w(g, x, gw)):7.1f}  bwd_w coded {t(lambda: K.qrow_bwd_w(g, xc, lo, hi, gw)):7.1f} us", flush=True)
