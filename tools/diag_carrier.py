#!/usr/bin/env python3
"""FQSS_DEBUG_CARRIER=1: which parameter gradients of the tiny HTDemucs step turn NaN under the codes-only dataflow (GPU box)"""
import os, sys
os.environ["FQSS_DEBUG_CARRIER"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_htdemucs import T, _models
from fqss_amd import ops, kernels as K

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hd_tiny_step.npz"))
model, fmodel = _models(g)
mix, src = T(g["mix"]).cuda(), T(g["src"]).cuda()
with torch.no_grad():
    for _ in range(50):
        model(mix)
    fest = fmodel(mix)
with ops.fast_codes(True):
    est = model(mix)
print("est finite:", bool(torch.isfinite(est).all()))
_, _, _, _, gest = K.hd_kd_loss(est.detach(), fest, src, torch.ones(2, device="cuda"), 0.1)
est.backward(gest)
bad = [k for k, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
print(len(bad), "parameters with non-finite gradients")
for k in bad[:40]:
    print("  ", k)
