#!/usr/bin/env python3
"""Which parameter gradients of the cfg-2 step differ in their BITS between two runs from the same state (fp32 atomics left in the step)?
    python tools/r05_determinism.py [tiny]            (FQSS_DETERMINISTIC=1 in the environment: the same under the deterministic mode)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd.data import synth_batch   # noqa: E402
from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize   # noqa: E402
from fqss_amd.runtime import KDTrainStep   # noqa: E402
from fqss_amd.smoke import build_pair   # noqa: E402

dev = torch.device("cuda", 0)
model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
B, T = (2, 8000) if "tiny" in sys.argv else (8, 32000)
x, tgt = synth_batch(B, T, seed=100, device=dev)
step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=0.0, clip=5.0)
step(x, tgt)
with torch.no_grad():
    for _ in range(49):
        model(x)
assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))
step(x, tgt)
runs = []
for r in range(3):
    res = step._fwd_bwd(x, tgt)
    torch.cuda.synchronize()
    runs.append((res["loss"].item(), step.arena.flat_g.clone()))
names = {id(p): n for n, p in model.named_parameters()}
print("FQSS_DETERMINISTIC =", os.environ.get("FQSS_DETERMINISTIC", "0"))
print("loss", [r[0] for r in runs])
bad = {}
for p, o in zip(step.arena.params, step.arena.offsets):
    a, b, c = (r[1][o:o + p.numel()] for r in runs)
    if not (torch.equal(a, b) and torch.equal(a, c)):
        n = names[id(p)]
        key = ".".join(q for q in n.split(".") if not q.isdigit())
        d = max(float((a - b).abs().max()), float((a - c).abs().max())) / max(float(a.abs().max()), 1e-30)
        bad.setdefault(key, []).append(d)
print(f"{sum(len(v) for v in bad.values())} of {len(step.arena.params)} parameter gradients differ in their bits between runs")
for k, v in sorted(bad.items()):
    print(f"  {k}: {len(v)} tensors, worst relative difference {max(v):.2e}")
