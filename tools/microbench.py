"""Per-kernel timing at cfg-2 shapes with HIP events (GPU box). Prints us and algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd import kernels as K

dev = "cuda"
B, CF, CB, M = 8, 512, 128, 3999


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def rep(name, us, nbytes, flops=None):
    s = f"{name:58s} {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s"
    if flops:
        s += f"  {flops / us / 1e6:8.1f} TFLOP/s"
    print(s, flush=True)


def act(C):
    t = K.empty_act((B, C, M), dev)
    t.normal_()
    return t


lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
slope = torch.tensor([0.25], device=dev)
gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
for C in (CF, CB):
    n = B * C * M
    z, g = act(C), act(C)
    gb = torch.zeros(C, device=dev)
    rep(f"actq_fwd C={C} QUANT none", timeit(lambda: K.actq_fwd(z, 0, None, 2, lo, hi, None)), 8 * n)
    rep(f"actq_fwd C={C} QUANT prelu", timeit(lambda: K.actq_fwd(z, 1, slope, 2, lo, hi, None)), 8 * n)
    rep(f"actq_fwd C={C} BYPASS prelu", timeit(lambda: K.actq_fwd(z, 1, slope, 0, None, None, None)), 8 * n)
    rep(f"actq_bwd C={C} QUANT none", timeit(lambda: K.actq_bwd(z, g, 0, None, 2, lo, hi, gacc)), 12 * n)
    rep(f"actq_bwd C={C} QUANT prelu", timeit(lambda: K.actq_bwd(z, g, 1, slope, 2, lo, hi, gacc)), 12 * n)
    rep(f"actq_bwd C={C} QUANT prelu +bias", timeit(lambda: K.actq_bwd(z, g, 1, slope, 2, lo, hi, gacc, gbias=gb, C=C)), 12 * n)
    rep(f"actq_bwd C={C} BYPASS none", timeit(lambda: K.actq_bwd(z, g, 0, None, 0, None, None, None)), 12 * n)
    rep(f"actq_bwd C={C} BYPASS relu +bias", timeit(lambda: K.actq_bwd(z, g, 2, None, 0, None, None, None, gbias=gb, C=C)), 12 * n)
    rep(f"axpby C={C}", timeit(lambda: K.axpby(z, g, 1.0)), 12 * n)
    gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rep(f"gn_fwd C={C}", timeit(lambda: K.gn_fwd(z, gm, bt, 1e-8)), 12 * n)
    _, mr = K.gn_fwd(z, gm, bt, 1e-8)
    gg, gbb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rep(f"gn_bwd C={C}", timeit(lambda: K.gn_bwd(g, z, gm, mr, gg, gbb)), 20 * n)
    w3 = torch.randn(C, 1, 3, device=dev)
    for dil in (1, 128):
        rep(f"dwconv_fwd C={C} dil={dil}", timeit(lambda: K.dwconv_fwd(z, w3, gm, dil, dil)), 8 * n)
        rep(f"dwconv_bwd_x C={C} dil={dil}", timeit(lambda: K.dwconv_bwd_x(g, w3, dil, dil)), 8 * n)
        gw3 = torch.zeros_like(w3)
        rep(f"dwconv_bwd_w C={C} dil={dil}", timeit(lambda: K.dwconv_bwd_w(g, z, gw3, dil, dil)), 8 * n)

# codes-only layers
for C in (CF,):
    xc = K.empty_codes((B, C, M), dev); xc.random_(0, 256)
    n = B * C * M
    gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    g = act(C)
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    rep(f"gnq_fwd C={C} (codes->codes)", timeit(lambda: K.gnq_fwd(xc, lo, hi, gm, bt, 1e-8, lo, hi, False)), 3 * n)
    _, _, mr = K.gnq_fwd(xc, lo, hi, gm, bt, 1e-8, lo, hi, False)
    gg, gbb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rep(f"gnq_bwd C={C}", timeit(lambda: K.gnq_bwd(xc, lo, hi, g, gm, bt, mr, lo, hi, gacc, gg, gbb)), 14 * n)
    w3 = torch.randn(C, 1, 3, device=dev)
    for dil in (1, 2, 128):
        rep(f"dwq_fwd C={C} dil={dil}", timeit(lambda: K.dwq_fwd(xc, lo, hi, w3, gm, dil, dil, 1, slope, lo, hi, False)), 2 * n)
        rep(f"dwq_bwd_z C={C} dil={dil}", timeit(lambda: K.dwq_bwd_z(xc, lo, hi, w3, gm, g, dil, dil, 1, slope, lo, hi, gacc, gbb)), 9 * n)
        gw3 = torch.zeros_like(w3)
        rep(f"dwq_bwd_w C={C} dil={dil}", timeit(lambda: K.dwq_bwd_w(g, xc, lo, hi, gw3, dil, dil)), 5 * n)

x128, x512 = act(CB), act(CF)
for (ci, co, xin) in ((128, 512, x128), (512, 128, x512), (128, 1024, x128)):
    w = torch.randn(co, ci, 1, device=dev) * 0.05
    bias = torch.randn(co, device=dev)
    fl = 2.0 * ci * co * B * M
    rep(f"pwconv_fwd {ci}->{co}", timeit(lambda: K.pwconv_fwd(xin, w, bias)), 4 * (ci + co) * B * M, fl)
    gz = act(co)
    rep(f"pwconv_bwd_x {ci}->{co}", timeit(lambda: K.pwconv_bwd_x(gz, w, ci)), 4 * (ci + co) * B * M, fl)
    gw = torch.zeros_like(w)
    rep(f"pwconv_bwd_w {ci}->{co}", timeit(lambda: K.pwconv_bwd_w(gz, xin, gw)), 4 * (ci + co) * B * M, fl)
# plain torch copy as the HBM yardstick
a = torch.empty(B * CF * 4000, device=dev); b = torch.empty_like(a)
rep("torch copy_ 65.5MB (yardstick)", timeit(lambda: b.copy_(a)), 8 * a.numel())
