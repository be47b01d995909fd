"""Fixed cost vs streaming cost of the GEMM-class kernels: time against the number of frames M (B = 8).
The intercept is launch + prologue + epilogue/atomics, the slope is the streaming rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd import kernels as K

dev = "cuda"
lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
slope = torch.tensor([0.25], device=dev)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ones = lambda c: torch.ones(c, 1, 1, device=dev) * 0.2
wc_up = K.wq_codes(torch.randn(512, 128, 1, device=dev) * 0.05, -ones(512), ones(512))
w1 = K.split3_planes(torch.randn(512, 128, device=dev) * 0.1)
w3 = K.split3_planes(torch.randn(256, 512, device=dev) * 0.05)
b1, b3, bu = torch.randn(512, device=dev), torch.randn(256, device=dev), torch.randn(512, device=dev)
ga, be = torch.ones(512, device=dev), torch.zeros(512, device=dev)
print("%6s %10s %10s %10s %10s %10s" % ("M", "q fwd", "q dgrad", "q wgrad", "tgemm T1", "tgemm T3"))
for M in (128, 500, 1000, 2000, 3999, 8000):
    B = 8
    xc = K.empty_codes((B, 128, M), dev); xc.random_(0, 256)
    gz = K.empty_act((B, 512, M), dev); gz.normal_()
    h = K.empty_act((B, 128, M), dev); h.normal_()
    acc = K.empty_act((B, 128, M), dev); acc.normal_()
    y = K.empty_act((B, 512, M), dev); y.normal_()
    gw = torch.zeros(512, 128, device=dev)
    st = K.tstat_buffer(2, B, dev)
    K.tstats(y, st[0])
    t = [timeit(lambda: K.qpw_fwdq(xc, wc_up, bu, None, lo, hi, 512, 1, slope, (lo, hi))),
         timeit(lambda: K.qpw_bwd_x(gz, wc_up)),
         timeit(lambda: K.qpw_bwd_w(gz, xc, lo, hi, gw)),
         timeit(lambda: K.tgemm(w1, h, b1, act=K.ACT_PRELU, slope=slope, stats_out=st[1])),
         timeit(lambda: K.tgemm(w3, y, b3, pro=1, pro_stats=st[0], pro_gamma=ga, pro_beta=be, pro_eps=1e-8, M1=128, r1=h, r2=acc))]
    print("%6d %10.1f %10.1f %10.1f %10.1f %10.1f" % (M, *t), flush=True)
