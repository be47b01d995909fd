"""A/B timing of the framing-conv weight gradients of cfg 2 (k_gemm_f32, 5 launches per step): FQSS_LIB selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd import kernels as K
dev = "cuda"
B, M, T = 8, 3999, 32000


def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


a = K.empty_act((B, 512, M), dev); a.normal_()
x2 = torch.randn(B, 2, T, device=dev)
gw2 = torch.zeros(512, 2, 16, device=dev)
a16 = K.empty_act((2 * B, 512, M), dev); a16.normal_()
x1 = torch.randn(2 * B, 1, T, device=dev)
gw1 = torch.zeros(512, 1, 16, device=dev)
print(os.environ.get("FQSS_LIB", "default"), "enc wgrad (Ci=2) %.1f us   dec wgrad (16 rows) %.1f us" % (
    timeit(lambda: K.frames_wgrad(a, x2, gw2, 8)), timeit(lambda: K.frames_wgrad(a16, x1, gw1, 8))))
