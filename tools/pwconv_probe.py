#!/usr/bin/env python3
"""times the pointwise GEMM path (K.pwconv_fwd six-product, bwd_x, bwd_w) at the narrow shapes of the HTDemucs DConv layers (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fqss_amd import kernels as K


def t(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, K_, Co, M in (("dconv k3 48->6 (K=144)", 4, 144, 6, 110250), ("dconv 1x1 6->96", 4, 6, 96, 110250), ("rewrite k3 48->96 (K=144)", 4, 144, 96, 110250),
                           ("dconv k3 96->12 (K=288)", 4, 288, 12, 27563), ("dconv 1x1 12->192", 4, 12, 192, 27563),
                           ("freq dconv k3 48->6", 2048, 144, 6, 431), ("freq dconv 1x1 6->96", 2048, 6, 96, 431), ("freq rewrite 48->96 k3", 2048, 144, 96, 431),
                           ("dconv k3 384->48 (K=1152)", 4, 1152, 48, 1723), ("dconv 1x1 48->768", 4, 48, 768, 1723)):
    x = K.empty_act((B, K_, M), "cuda"); x.normal_()
    w = torch.randn(Co, K_, 1, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda")
    gz = K.empty_act((B, Co, M), "cuda"); gz.normal_()
    gw = torch.zeros_like(w)
    f = t(lambda: K.pwconv_fwd(x, w, b, six=True))
    bx = t(lambda: K.pwconv_bwd_x(gz, w, K_))
    bw = t(lambda: K.pwconv_bwd_w(gz, x, gw))
    mb = (B * K_ * M + B * Co * M) * 4e-6
    print(f"{name:28s} fwd {f:7.1f} us  bwd_x {bx:7.1f} us  bwd_w {bw:7.1f} us   (in+out {mb:6.0f} MB -> {mb / 5e3 * 1e3:5.1f} us at 5 TB/s)", flush=True)
