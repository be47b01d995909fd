#!/usr/bin/env python3
"""implicit-GEMM stride-1 Conv1d (K.conv1d_s1_*) against frame gather + pointwise GEMM at the HTDemucs k3 shapes (GPU box)"""
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


for name, B, Ci, Co, M, d in (("dconv 48->6", 4, 48, 6, 110250, 1), ("rewrite 48->96", 4, 48, 96, 110250, 1), ("dconv 96->12", 4, 96, 12, 27563, 2),
                              ("rewrite 96->192", 4, 96, 192, 27563, 1), ("freq dconv 48->6", 2048, 48, 6, 431, 2), ("freq rewrite 48->96", 2048, 48, 96, 431, 1),
                              ("dconv 384->48", 4, 384, 48, 1723, 1), ("rewrite 384->768", 4, 384, 768, 1723, 1), ("freq dconv 192->24", 128, 192, 24, 431, 1)):
    x = K.empty_act((B, Ci, M), "cuda"); x.normal_()
    w = torch.randn(Co, Ci * 3, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda")
    gz = K.empty_act((B, Co, M), "cuda"); gz.normal_()
    gw = torch.zeros_like(w)
    geom = K.ConvGeom((1, 3), (1, 1), (0, d), (1, d))
    x4 = x.unsqueeze(2)
    fi = t(lambda: K.conv1d_s1_fwd(x, w, b, 3, d, d))
    wt = w.reshape(Co, Ci, 3).flip(2).permute(1, 0, 2).reshape(Ci, Co * 3).contiguous()
    di = t(lambda: K.conv1d_s1_fwd(gz, wt, None, 3, d, d))
    wi = t(lambda: K.conv1d_s1_bwd_w(gz, x, gw, 3, d, d))
    fg = t(lambda: K.frames_gather(x4, geom))
    f = K.frames_gather(x4, geom)[0]
    fp = t(lambda: K.pwconv_fwd(f, w.unsqueeze(-1), b, six=True))
    bx = t(lambda: K.pwconv_bwd_x(gz, w.unsqueeze(-1), Ci * 3))
    gf = K.pwconv_bwd_x(gz, w.unsqueeze(-1), Ci * 3)
    ol = t(lambda: K.frames_ola(gf, None, (B, Ci, 1, M), geom))
    bw = t(lambda: K.pwconv_bwd_w(gz, f, gw.unsqueeze(-1)))
    print(f"{name:22s} fwd implicit {fi:7.1f} | gather {fg:7.1f} + gemm {fp:7.1f}   dgrad implicit {di:7.1f} | gemm {bx:7.1f} + ola {ol:7.1f}   wgrad implicit {wi:7.1f} | gemm {bw:7.1f}", flush=True)
