#!/usr/bin/env python3
"""times the streaming attention core at the cfg 5 shapes (GPU box)"""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fqss_amd import kernels as K


def t(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, nh, hd = 4, 8, 64
for Lq, Lk in ((3448, 3448), (1723, 1723), (3448, 1723), (1723, 3448)):
    E = nh * hd
    q, k, v, go = (torch.randn(B, L, E, device="cuda") * 0.3 for L in (Lq, Lk, Lk, Lq))
    o, st = K.attn_long_fwd(q, k, v, nh, True)
    f = t(lambda: K.attn_long_fwd(q, k, v, nh, True))
    b = t(lambda: K.attn_long_bwd(q, k, v, o, go, st, nh, True))
    gf = 4.0 * Lq * Lk * hd * B * nh * 1e-9
    print(f"Lq {Lq} Lk {Lk}: fwd {f:.3f} ms ({gf / f:.1f} TF/s)  bwd {b:.3f} ms ({2.5 * gf / b:.1f} TF/s)", flush=True)
