#!/usr/bin/env python3
"""Golden vectors of the true-integer export wrappers from the REAL reference's TorchWeightFakeQuantize /
TorchActivationFakeQuantize (qat_quant.py:15-56).  Usage: python tools/make_goldens_export.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402
from quantization.qat import qat_quant as RQ  # noqa: E402


def main():
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    d = {}
    for tag, shape, axis in (("w0", (12, 7, 3), 0), ("w1", (6, 10, 5), 1), ("w2d", (9, 16), 0)):
        q = RQ.GradientWeightFakeQuantize(True, shape, ch_out_idx=axis)
        w = MG.keyed_randn("exp." + tag, shape, 0.4)
        q(w)                                                       # observer call: records the per-channel ranges
        with torch.no_grad():
            q.min_range.mul_(0.8); q.max_range.mul_(0.9)           # tighten: clipping is exercised
        with torch.no_grad():                                      # (the export wrappers are an inference-time construct)
            t = RQ.TorchWeightFakeQuantize(q)
            y = t(w)
        d[tag + ".w"], d[tag + ".min"], d[tag + ".max"], d[tag + ".y"] = MG.npy(w), MG.npy(q.min_range), MG.npy(q.max_range), MG.npy(y)
        d[tag + ".scales"], d[tag + ".axis"] = MG.npy(t.scales), np.array(axis)
        d[tag + ".codes"] = np.rint(MG.npy(y) / MG.npy(t.scales).reshape([-1 if i == axis else 1 for i in range(len(shape))])).astype(np.int8)
    for tag, lo, hi in (("a0", -1.3, 1.7), ("a1", 0.0, 6.0), ("a2", -0.5, 0.5), ("a3", 0.2, 3.1)):
        q = RQ.GradientActivationFakeQuantize(True)
        with torch.no_grad():
            q.min_range.fill_(lo); q.max_range.fill_(hi)
        x = MG.keyed_randn("exp." + tag, (3, 11, 50), 1.5) + 0.5 * (lo + hi)
        with torch.no_grad():
            t = RQ.TorchActivationFakeQuantize(q)
            y = t(x)
        d[tag + ".x"], d[tag + ".y"], d[tag + ".range"] = MG.npy(x), MG.npy(y), np.array([lo, hi], dtype=np.float32)
        d[tag + ".scale"], d[tag + ".zero_point"] = np.float64(t.scale), np.array(t.zero_point)
    np.savez_compressed(os.path.join(out, "export.npz"), **d)
    print("export:", len(d), "arrays")


if __name__ == "__main__":
    main()
