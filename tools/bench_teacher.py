"""Per-kernel timings of the fused teacher chain (csrc/teacher.hip) at the bench shape (B=8, 4 s @ 8 kHz)."""
import sys, torch
sys.path.insert(0, ".")
from fqss_amd import kernels as K

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def main():
    dev = "cuda"
    B, M = 8, 3999
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)
    h = K.empty_act((B, 128, M), dev); h.copy_(R(B, 128, M))
    acc = K.empty_act((B, 128, M), dev); acc.copy_(R(B, 128, M))
    y1 = K.empty_act((B, 512, M), dev); y1.copy_(R(B, 512, M))
    w1 = K.split3_planes(R(512, 128) * 0.1); w3 = K.split3_planes(R(256, 512) * 0.05)
    b1, b3 = R(512), R(256)
    slope = torch.full((1,), 0.25, device=dev)
    st = K.tstat_buffer(1, B, dev)[0]
    K.tstats(y1, st)
    so2 = K.tstat_buffer(1, B, dev)[0]
    ga, be = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    print("T1 128->512 prelu+stats   %.1f us" % timeit(lambda: K.tgemm(w1, h, b1, act=K.ACT_PRELU, slope=slope, stats_out=so2)))
    print("T1 128->512 plain         %.1f us" % timeit(lambda: K.tgemm(w1, h, b1)))
    print("T3 512->256 GN pro + res  %.1f us" % timeit(lambda: K.tgemm(w3, y1, b3, pro=1, pro_stats=st, pro_gamma=ga, pro_beta=be, pro_eps=1e-8, M1=128, r1=h, r2=acc)))
    print("T3 512->256 plain         %.1f us" % timeit(lambda: K.tgemm(w3, y1, b3)))
    wd, bd = R(512, 1, 3), R(512)
    for dil in (1, 2, 4, 8, 64, 128):
        so = K.tstat_buffer(1, B, dev)[0]
        print("T2 dw dil=%-3d              %.1f us" % (dil, timeit(lambda: K.tdw(y1, st, ga, be, 1e-8, wd, bd, slope, so, dil, dil))))
    print("tstats 512                %.1f us" % timeit(lambda: K.tstats(y1, st)))

if __name__ == "__main__":
    main()
