#!/usr/bin/env python3
"""times the streaming attention core at the cfg 5 shapes (and the dual-path shapes of cfg 3 / 4) in its forms: float operands on the bf16
matrix cores (exact 3-piece split; FQSS_ATTN_MFMA=f32 in the environment: the fp32-MFMA kernels of rounds 1-2) and coded operands (GPU box)"""
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


for Lq, Lk, B, nh, hd, bf in ((3448, 3448, 4, 8, 64, True), (1723, 1723, 4, 8, 64, True), (3448, 1723, 4, 8, 64, True), (1723, 3448, 4, 8, 64, True),
                              (250, 250, 64, 8, 32, False), (64, 64, 250, 8, 32, False), (250, 250, 194, 4, 16, False), (194, 194, 250, 4, 16, False)):
    E = nh * hd
    shp = lambda L: (B, L, E) if bf else (L, B, E)
    q, k, v, go = (torch.randn(*shp(L), device="cuda") * 0.3 for L in (Lq, Lk, Lk, Lq))
    o, st = K.attn_long_fwd(q, k, v, nh, bf)
    f = t(lambda: K.attn_long_fwd(q, k, v, nh, bf))
    b = t(lambda: K.attn_long_bwd(q, k, v, o, go, st, nh, bf))
    qc, kc, vc = (torch.randint(0, 256, shp(L), device="cuda", dtype=torch.uint8) for L in (Lq, Lk, Lk))
    rng = [(torch.tensor([-0.9], device="cuda"), torch.tensor([0.8], device="cuda")) for _ in range(3)]
    oc, stc = K.attn_long_fwd_c(qc, kc, vc, rng, nh, bf)
    fc = t(lambda: K.attn_long_fwd_c(qc, kc, vc, rng, nh, bf))
    bc = t(lambda: K.attn_long_bwd_c(qc, kc, vc, rng, oc, go, stc, nh, bf))
    gf = 4.0 * Lq * Lk * hd * B * nh * 1e-9
    print(f"Lq {Lq} Lk {Lk} B {B} nh {nh} hd {hd}: float fwd {f:.3f} ms ({gf / f:.1f} TF/s)  bwd {b:.3f} ms ({2.5 * gf / b:.1f} TF/s) | coded fwd {fc:.3f} ms  bwd {bc:.3f} ms", flush=True)
