#!/usr/bin/env python3
"""eval-mode forward of the quantized ConvTasNet with and without the codes-only dataflow (GPU box): equality and time"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fqss_amd import ops
from fqss_amd.data import synth_batch
from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model, enable_observer
from fqss_amd.smoke import QCFG

torch.manual_seed(0)
m = quantize_model(create_model({"name": "ConvTasNet", "n_src": 2, "kernel_size": 16, "stride": 8}), dict(QCFG)).cuda().train()
x, _ = synth_batch(8, 32000, seed=0, device="cuda")
with torch.no_grad():
    for _ in range(50):
        m(x)
enable_observer(m, False)
m.eval()


def run(fast, n=10):
    with torch.no_grad(), ops.fast_codes(fast):
        y = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            y = m(x)
        torch.cuda.synchronize()
    return y, (time.perf_counter() - t0) / n * 1e3


y0, t0 = run(False)
y1, t1 = run(True)
print(f"plain eval {t0:.2f} ms, codes-only eval {t1:.2f} ms, equal {torch.equal(y0, y1)}, max diff {(y0 - y1).abs().max().item():.3e}")
