"""Timings of the student's codes-only element-wise / norm / depthwise kernels (csrc/fused_q.hip) at cfg-2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd import kernels as K

dev = "cuda"
B, M = 8, 3999


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rep(name, us, nbytes):
    print(f"{name:52s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
slope = torch.tensor([0.25], device=dev)
gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
for C in (512, 128):
    n = B * C * M
    xc = K.empty_codes((B, C, M), dev); xc.random_(0, 256)
    yc = K.empty_codes((B, C, M), dev); yc.random_(0, 256)
    g = K.empty_act((B, C, M), dev); g.normal_()
    gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    gg, gbb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rep(f"gnq_fwd C={C} (codes->codes)", timeit(lambda: K.gnq_fwd(xc, lo, hi, gm, bt, 1e-8, lo, hi, False)), 3 * n)
    _, _, mr = K.gnq_fwd(xc, lo, hi, gm, bt, 1e-8, lo, hi, False)
    rep(f"gnq_bwd C={C}", timeit(lambda: K.gnq_bwd(xc, lo, hi, g, gm, bt, mr, lo, hi, gacc, gg, gbb)), 14 * n)
    rep(f"ewq_fwd C={C} a+b codes", timeit(lambda: K.ewq_fwd(xc, lo, hi, yc, lo, hi, None, 1.0, 0, None, lo, hi, False)), 3 * n)
    rep(f"ewq_fwd C={C} prelu(a)", timeit(lambda: K.ewq_fwd(xc, lo, hi, None, None, None, None, 0.0, 1, slope, lo, hi, False)), 2 * n)
    rep(f"ewq_bwd C={C} a+b codes", timeit(lambda: K.ewq_bwd(xc, lo, hi, yc, lo, hi, None, 1.0, g, 0, None, lo, hi, gacc)), 10 * n)
    rep(f"ewq_bwd C={C} prelu(a)", timeit(lambda: K.ewq_bwd(xc, lo, hi, None, None, None, None, 0.0, g, 1, slope, lo, hi, gacc)), 9 * n)
    if C == 512:
        w3 = torch.randn(C, 1, 3, device=dev)
        for dil in (1, 2, 8, 128):
            rep(f"dwq_fwd C={C} dil={dil}", timeit(lambda: K.dwq_fwd(xc, lo, hi, w3, gm, dil, dil, 1, slope, lo, hi, False)), 2 * n)
            rep(f"dwq_bwd_z C={C} dil={dil}", timeit(lambda: K.dwq_bwd_z(xc, lo, hi, w3, gm, g, dil, dil, 1, slope, lo, hi, gacc, gbb)), 9 * n)
            gw3 = torch.zeros_like(w3)
            rep(f"dwq_bwd_w C={C} dil={dil}", timeit(lambda: K.dwq_bwd_w(g, xc, lo, hi, gw3, dil, dil)), 5 * n)
            rep(f"dwconv_bwd_x C={C} dil={dil} (fp32)", timeit(lambda: K.dwconv_bwd_x(g, w3, dil, dil)), 8 * n)
    z = K.empty_act((B, C, M), dev); z.normal_()
    rep(f"actq_fwd C={C} -> codes", timeit(lambda: K.actq_fwd(z, 0, None, 2, lo, hi, None, want_idx=True)), 5 * n)
    rep(f"actq_bwd C={C} +bias", timeit(lambda: K.actq_bwd(z, g, 0, None, 2, lo, hi, gacc, gbias=gbb, C=C)), 12 * n)
    rep(f"axpby C={C}", timeit(lambda: K.axpby(z, g, 1.0)), 12 * n)
