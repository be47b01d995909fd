"""Streaming rate of the un-fused fake-quant passes (fqss_actq_fwd / bwd) at the dual-path tensor size, in several 2-D views."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd import kernels as K


def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


n = 250 * 194 * 64
lo, hi = torch.tensor([-2.0], device="cuda"), torch.tensor([2.5], device="cuda")
gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
for shape in ((485, 6400), (1, n), (48500, 64), (3031, 1024), (194, 16000)):
    z = torch.randn(*shape, device="cuda")
    g = torch.randn(*shape, device="cuda")
    f_idx = timeit(lambda: K.actq_fwd(z, 0, None, 2, lo, hi, None, want_idx=True))
    f_no = timeit(lambda: K.actq_fwd(z, 0, None, 2, lo, hi, None))
    b = timeit(lambda: K.actq_bwd(z, g, 0, None, 2, lo, hi, gacc))
    print(shape, "fwd+idx %.1f us (%.2f TB/s)  fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (
        f_idx, 9 * n / f_idx * 1e-6, f_no, 8 * n / f_no * 1e-6, b, 12 * n / b * 1e-6))
