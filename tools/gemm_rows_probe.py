#!/usr/bin/env python3
"""Isolated timing of the fp32-MFMA row GEMMs (csrc/gemm.hip fqss_rowlin_*) at the shapes of cfg 3 / cfg 4: achieved TFLOP/s
against the fp32 MFMA peak (157.3 TFLOP/s).  HIP events on torch's current stream (the stream the kernels are launched on)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from fqss_amd import kernels as K  # noqa: E402

PEAK = 157.3
SHAPES = [("dptnet in_proj", 48500, 64, 192), ("dptnet lstm proj", 48500, 64, 1024), ("dptnet linear", 48500, 256, 64),
          ("sepformer in_proj", 8500, 256, 768), ("sepformer ffn0", 8500, 256, 1024), ("sepformer ffn3", 8500, 1024, 256),
          ("sepformer out_proj", 8500, 256, 256)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    out = []
    for name, R, Ci, Co in SHAPES:
        x, w, b = torch.randn(R, Ci, device="cuda"), torch.randn(Co, Ci, device="cuda"), torch.randn(Co, device="cuda")
        g, gw = torch.randn(R, Co, device="cuda"), torch.zeros(Co, Ci, device="cuda")
        fl = 2.0 * R * Ci * Co
        rec = {"shape": name, "R": R, "Ci": Ci, "Co": Co}
        for kind, fn in (("fwd", lambda: K.rowlin_fwd(x, w, b)), ("bwd_x", lambda: K.rowlin_bwd_x(g, w)),
                         ("bwd_w", lambda: K.rowlin_bwd_w(g, x, gw))):
            us = timeit(fn)
            rec[kind] = {"us": round(us, 1), "TFLOPs": round(fl / us * 1e-6, 1), "frac": round(fl / us * 1e-6 / PEAK, 3)}
        out.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
