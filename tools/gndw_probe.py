"""fqss_gndwq_fwd against the two launches it replaces at the cfg-2 shape (8 x 512 x 3999), cold operands, hipGraph replays.
    python tools/gndw_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from fqss_amd import kernels as K
from fqss_amd import roofline_cases as RC

def main():
    dev = torch.device("cuda", 0)
    B, C, M, sets = 8, 512, 3999, 4
    T1 = lambda v: torch.tensor([v], device=dev)
    lo, hi, lo1, hi1, lo2, hi2, slope = T1(-1.7), T1(2.9), T1(-2.2), T1(2.4), T1(-0.6), T1(1.9), T1(0.25)
    xs = [K.empty_codes((B, C, M), dev).random_(0, 256) for _ in range(sets)]
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    w, bias = torch.randn(C, 1, 3, device=dev), torch.randn(C, device=dev) * 0.1
    xi = xs[0].to(torch.int64)
    st = K.CodeStats(torch.stack([xi.sum(dim=(1, 2)), (xi * xi).sum(dim=(1, 2))], 1).reshape(-1).contiguous(), 1)
    for dil in (1, 4, 128):
        def two(i):
            _, yc1, _ = K.gnq_fwd(xs[i % sets], lo, hi, gamma, beta, 1e-8, lo1, hi1, write_out=False, stats=st)
            std = K.new_stats("dwq", B, C, M, dev)
            K.dwq_fwd(yc1, lo1, hi1, w, bias, dil, dil, K.ACT_PRELU, slope, lo2, hi2, write_out=False, stats=std)
        def one(i):
            xc = xs[i % sets]
            _, y1, mr = K.gnq_fwd_deferred(xc)
            d = dict(xc=xc, qmin_x=lo, qmax_x=hi, gamma=gamma, beta=beta, eps=1e-8, qmin=lo1, qmax=hi1, stats=st, yc=y1, mean_rstd=mr, done=False)
            K.gndwq_fwd(d, w, bias, dil, dil, K.ACT_PRELU, slope, lo2, hi2, True)
        for name, fn in (("gnq_fwd + dwq_fwd", two), ("gndwq_fwd", one), ("gnq_fwd + dwq_fwd", two), ("gndwq_fwd", one)):
            ms = RC.time_case(dict(fn=fn))
            print(f"dil {dil:3d}  {name:20s} {ms * 1e3:7.1f} us", flush=True)

if __name__ == "__main__":
    main()
