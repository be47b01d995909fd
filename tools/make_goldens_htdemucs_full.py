#!/usr/bin/env python3
"""Full-ARCHITECTURE digests for SURVEY.md §8 row a15 (cfg 5) from the REAL reference (this container only): HTDemucsQ with the
shipped architecture (4 sources, stereo, channels 48, n_fft 4096, depth 4, 5 transformer layers of 8 heads, bottom_channels 512 ->
head_dim 64; 41.6 M parameters), name-keyed deterministic weights (restated on the GPU side: tests/helpers_cfg5.py), B = 1 x 1 s
(`segment` = 1 so the eval-mode teacher is not padded to 10 s).  cfg5_step.npz holds small digests only: teacher output, observer
call 1 with a backward (no quantizer active: every kernel at full width, free of flips) -- output, loss, per-parameter gradient norms
and a few gradient slices -- and the observer ranges after 5 calls.
Usage: python tools/make_goldens_htdemucs_full.py"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens_htdemucs as MGH  # noqa: E402  (installs the shim + spectro / ispectro)
from quantization.qat.models.htdemucsq import HTDemucsQ  # noqa: E402
from quantization.qat.models.load_model import quantize_model  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers_cfg5 import cfg5_kw, cfg5_fill, cfg5_batch  # noqa: E402

npy = MGH.npy


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=1, help="excerpt length; 10 = the BASELINE workload's segment (cfg5_full_step.npz)")
    ap.add_argument("--fname", default="cfg5_step.npz")
    a = ap.parse_args()
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    sub = 1 if a.seconds == 1 else 41            # the 10 s outputs are stored subsampled (digests stay small)
    torch.set_num_threads(int(os.environ.get("FQSS_GOLDEN_THREADS", "8")))
    torch.manual_seed(0)
    d = {}
    model = HTDemucsQ(**cfg5_kw(a.seconds))
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, MGH.QCFG)
    cfg5_fill(model, "S."); cfg5_fill(fmodel, "T.")
    model.train(); fmodel.eval()
    mix, src = cfg5_batch(a.seconds)
    d["mix_sum"], d["src_sumsq"] = np.float64(mix.double().sum()), np.float64((src.double() ** 2).sum())
    d["param_names"] = np.array([k for k, _ in model.named_parameters()])
    d["param_sum"] = np.array([float(p.double().sum()) for _, p in model.named_parameters()])
    with torch.no_grad():
        fest = fmodel(mix)
    d["fest"] = npy(fest).astype(np.float32)[:, :, :, ::7 * sub]
    d["fest_rms"] = np.float64(fest.double().pow(2).mean().sqrt())
    est, _, w, task, kd, loss = MGH.kd_step(model, fmodel, mix, src, weights=(1.0, 1.0, 1.0, 1.0))
    loss.backward()
    d["o1.est"] = npy(est)[..., ::sub]
    d["o1.est_rms"] = np.float64(est.detach().double().pow(2).mean().sqrt())
    d["o1.w"], d["o1.task"], d["o1.kd"], d["o1.loss"] = npy(w), npy(task), npy(kd), npy(loss)
    d["o1.grad_norm"] = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    for k in ("encoder.0.conv.conv2d.weight", "tencoder.3.rewrite.conv1d.weight", "decoder.0.conv_tr.convTr2d.weight",
              "crosstransformer.layers.1.cross_attn.mha.in_proj_weight", "crosstransformer.layers_t.4.self_attn.mha.out_proj.weight",
              "crosstransformer.layers.2.linear1.linear.weight", "channel_upsampler.conv1d.weight", "freq_emb.embedding.embedding.weight"):
        g = dict(model.named_parameters())[k].grad
        d["o1.grad." + k] = npy(g.reshape(-1)[:4096])
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        for _ in range(4):
            model(mix)
    for k, v in model.state_dict().items():
        if k.endswith("_range") and v.numel() == 1:
            d["obs5." + k] = npy(v)
    d["seconds"], d["sub"] = np.array(a.seconds), np.array(sub)
    np.savez_compressed(os.path.join(out, a.fname), **d)
    print("cfg5_step:", len(d), "arrays; loss", float(loss), "fest rms", float(d["fest_rms"]))


if __name__ == "__main__":
    main()
