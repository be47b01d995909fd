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
for _ in range(3):
    K.pwconv_fwd(x, w, bias)            # x3
    K.qpw_fwd(xc, wc, bias, lo, hi)     # q fwd
    K.qpw_bwd_x(gz, wc)                 # q dgrad
    K.qpw_bwd_w(gz, xc, lo, hi, gw)     # q wgrad
torch.cuda.synchronize()
