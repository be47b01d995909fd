import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd import kernels as K
dev = "cuda"
B, M = 8, 3999
x = K.empty_act((B, 128, M), dev); x.normal_()
w = torch.randn(512, 128, 1, device=dev) * 0.05
bias = torch.randn(512, device=dev)
lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
wlo, whi = -torch.ones(512, 1, 1, device=dev) * 0.2, torch.ones(512, 1, 1, device=dev) * 0.2
_, xc = K.actq_fwd(x, 0, None, 2, lo, hi, None, want_idx=True)
wc = K.wq_codes(w, wlo, whi)
gz = K.empty_act((B, 512, M), dev); gz.normal_()
gw = torch.zeros(512, 128, device=dev)
def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x5 = K.empty_act((B, 512, M), dev); x5.normal_()
_, xc5 = K.actq_fwd(x5, 0, None, 2, lo, hi, None, want_idx=True)
w5 = torch.randn(128, 512, 1, device=dev) * 0.05
wc5 = K.wq_codes(w5, -torch.ones(128, 1, 1, device=dev) * 0.2, torch.ones(128, 1, 1, device=dev) * 0.2)
gz1 = K.empty_act((B, 128, M), dev); gz1.normal_()
gw5 = torch.zeros(128, 512, device=dev)
b5 = torch.randn(128, device=dev)
print("x3 fwd 128->512  %.1f us" % timeit(lambda: K.pwconv_fwd(x, w, bias)))
print("x3 fwd 512->128  %.1f us" % timeit(lambda: K.pwconv_fwd(x5, w5, b5)))
print("q fwd 128->512   %.1f us" % timeit(lambda: K.qpw_fwd(xc, wc, bias, lo, hi)))
print("q fwd 512->128   %.1f us" % timeit(lambda: K.qpw_fwd(xc5, wc5, b5, lo, hi)))
print("q dgrad 128->512 %.1f us" % timeit(lambda: K.qpw_bwd_x(gz, wc)))
print("q dgrad 512->128 %.1f us" % timeit(lambda: K.qpw_bwd_x(gz1, wc5)))
print("q wgrad 128->512 %.1f us" % timeit(lambda: K.qpw_bwd_w(gz, xc, lo, hi, gw)))
print("q wgrad 512->128 %.1f us" % timeit(lambda: K.qpw_bwd_w(gz1, xc5, lo, hi, gw5)))
