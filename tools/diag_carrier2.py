import os, sys
os.environ["FQSS_DEBUG_CARRIER"] = "1"
import torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fqss_amd import ops
from fqss_amd.quantization.qat import qat_layers as QL
P = dict(gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8)
for C in (16, 64, 256):
    ln = QL.LayerNormQ(nn.LayerNorm(C), gradient_based=True, act_quant=True).cuda()
    lin = QL.LinearQ(nn.Linear(C, 32), **P).cuda()
    x = torch.randn(6, 20, C, device="cuda", requires_grad=True)
    with torch.no_grad():
        for _ in range(50):
            lin(ln(x))
    with ops.fast_codes(True):
        h = ln(x)
        y = lin(h)
    y.sum().backward()
    print(C, "h has rowq", hasattr(h, "_fqss_rowq"), "h finite", bool(torch.isfinite(h).all()), "y finite", bool(torch.isfinite(y).all()),
          "gw finite", bool(torch.isfinite(lin.linear.weight.grad).all()), "numel", h.numel())
