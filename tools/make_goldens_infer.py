#!/usr/bin/env python3
"""Golden vectors of the evaluation side from the REAL reference's process.model_infer /
swap_channel_order (process.py:105-194), run on a tiny quantized ConvTasNet in eval mode.  torchmetrics (third party, absent) is
supplied as its published SI-SNR formula.  Usage: python tools/make_goldens_infer.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402  (installs the import shim)
import torchmetrics  # noqa: E402  (the shim's stub)


class ScaleInvariantSignalNoiseRatio:
    def __call__(self, preds, target):
        eps = torch.finfo(preds.dtype).eps
        target = target - torch.mean(target, dim=-1, keepdim=True)
        preds = preds - torch.mean(preds, dim=-1, keepdim=True)
        alpha = (torch.sum(preds * target, dim=-1, keepdim=True) + eps) / (torch.sum(target ** 2, dim=-1, keepdim=True) + eps)
        ts = alpha * target
        val = (torch.sum(ts ** 2, dim=-1) + eps) / (torch.sum((ts - preds) ** 2, dim=-1) + eps)
        return (10 * torch.log10(val)).mean()


import process as RP  # noqa: E402

RP.ScaleInvariantSignalNoiseRatio = ScaleInvariantSignalNoiseRatio
from quantization.qat.models.convtasnetq import ConvTasNetQ  # noqa: E402
from quantization.qat.models.load_model import enable_observer, quantize_model  # noqa: E402


def main():
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    d = {}
    torch.manual_seed(0)
    model = quantize_model(ConvTasNetQ(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1),
                           MG.QCFG)
    MG.cfg1_fill(model, "I.")
    model.train()
    x, tgt = MG.synth_batch(2, 1500, seed=5)
    with torch.no_grad():
        for _ in range(50):
            model(x)
    enable_observer(model, False)
    model.eval()
    for k, v in model.state_dict().items():
        d["sd." + k] = MG.npy(v)
    mix, clean = MG.synth_batch(1, 3100, seed=6)
    mix, clean = mix[0], clean[0]                                # [1, L], [2, L]
    d["mix"], d["clean"] = MG.npy(mix), MG.npy(clean)
    with torch.no_grad():
        d["whole"] = MG.npy(RP.model_infer(model, mix, n_srcs=2, device="cpu"))
        d["chunked"] = MG.npy(RP.model_infer(model, mix, n_srcs=2, segment=1000, overlap=0.25, device="cpu", target=clean))
        d["chunked_nt"] = MG.npy(RP.model_infer(model, mix, n_srcs=2, segment=1000, overlap=0.25, device="cpu"))
    # swap_channel_order and the SI-SNR on their own: estimates = noisy copies of the targets in the WRONG order
    est = torch.stack([clean[1] + 0.05 * MG.keyed_randn("inf.n0", (3100,)), clean[0] + 0.05 * MG.keyed_randn("inf.n1", (3100,))])
    d["swap.in"], d["swap.out"] = MG.npy(est), MG.npy(RP.swap_channel_order(est, clean))
    m = ScaleInvariantSignalNoiseRatio()
    d["sisnr"] = np.array([[float(m(est[p:p + 1], clean[q])) for q in range(2)] for p in range(2)], dtype=np.float32)
    np.savez_compressed(os.path.join(out, "infer.npz"), **d)
    print("infer:", len(d), "arrays;", d["sisnr"])


if __name__ == "__main__":
    main()
