#!/usr/bin/env python3
"""Model-level golden vectors for SURVEY.md §8 row a15 (cfg 5: HTDemucs) from the REAL reference (this container only).

hd_tiny_step.npz: a tiny HTDemucsQ (2 sources, stereo, channels 8, nfft 2048, depth 4 (the reference only quantizes depth 4: `train_res_dec` is keyed on "decoder.3"), bottom_channels 16, 3 transformer layers of 2
heads; the structure of the full model: both branches, DConv, frequency embedding, self- and cross-attention layers, 8-bit I/O
blocks with the residual decoders) -- float teacher and W8A8 student with name-keyed deterministic weights, 50 observer forwards,
then ONE quantizing KD step of the htdemucs solver (L1 task + SDR-weighted L1 KD, solver.py:333-359): outputs, loss, every gradient.
demucs.spec.spectro / ispectro (demucs~=4.0.0, absent here) are supplied as the torch.stft / torch.istft calls they wrap.
Usage: python tools/make_goldens_htdemucs.py"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
import demucs.spec as _spec  # noqa: E402  (the stub module of the shim)


def spectro(x, n_fft=512, hop_length=None, pad=0):
    *other, length = x.shape
    x = x.reshape(-1, length)
    z = torch.stft(x, n_fft * (1 + pad), hop_length or n_fft // 4, window=torch.hann_window(n_fft).to(x), win_length=n_fft, normalized=True,
                   center=True, return_complex=True, pad_mode="reflect")
    _, freqs, frame = z.shape
    return z.view(*other, freqs, frame)


def ispectro(z, hop_length=None, length=None, pad=0):
    *other, freqs, frames = z.shape
    n_fft = 2 * freqs - 2
    z = z.view(-1, freqs, frames)
    win_length = n_fft // (1 + pad)
    x = torch.istft(z, n_fft, hop_length, window=torch.hann_window(win_length).to(z.real), win_length=win_length, normalized=True,
                    length=length, center=True)
    _, length = x.shape
    return x.view(*other, length)


_spec.spectro, _spec.ispectro = spectro, ispectro

import make_goldens as MG  # noqa: E402
import make_goldens_dptnet as MD  # noqa: E402
from quantization.qat.models.htdemucsq import HTDemucsQ  # noqa: E402
from quantization.qat.models.load_model import enable_observer, quantize_model  # noqa: E402
from quantization.qat import qat_quant as RQ  # noqa: E402

npy, keyed_randn = MG.npy, MG.keyed_randn
TINY_KW = dict(sources=["a", "b"], audio_channels=2, channels=8, nfft=2048, depth=4, bottom_channels=16, t_layers=3, t_heads=2)
QCFG = dict(qat=True, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, in_quant=False, in_act_n_bits=8,
            out_quant=True, out_act_n_bits=8, n_splitter=2, n_combiner=2, observer=True)


def fill(mod, prefix):
    """name-keyed deterministic fill: matrices ~ N(0, 1/fan_in)-ish so activations stay O(1); LayerScale / norm weights near their
    working values; biases small"""
    with torch.no_grad():
        for k, p in mod.named_parameters():
            if k.endswith("min_range") or k.endswith("max_range"):
                continue
            if k.endswith(".scale"):                                   # LayerScale
                p.copy_(0.5 + keyed_randn(prefix + k, tuple(p.shape), 0.1))
            elif p.dim() == 1 and ("norm" in k or ".gn." in k or k.split(".")[-2] in ("1", "4")) and k.endswith("weight"):
                p.copy_(1.0 + keyed_randn(prefix + k, tuple(p.shape), 0.1))
            elif p.dim() == 1:
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 0.05))
            else:
                fan = p[0].numel() if "convTr" not in k and "conv_tr" not in k else p.shape[0] * p[0, 0].numel()
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 1.0 / np.sqrt(max(fan, 1))))


def new_sdr(references, estimates):
    """demucs.evaluate.new_sdr (demucs~=4.0.0): 10 log10(sum ref^2 / sum (ref - est)^2) over (channels, time), delta 1e-7"""
    delta = 1e-7
    num = torch.sum(torch.square(references), dim=(2, 3)) + delta
    den = torch.sum(torch.square(references - estimates), dim=(2, 3)) + delta
    return 10 * torch.log10(num / den)


def kd_step(model, fmodel, mix, sources, kd_lambda=0.1, weights=(1.0, 1.0)):
    """solver.py:325-366 (train branch, optim.loss = l1, kd_lambda > 0)"""
    estimate = model(mix)
    dims = tuple(range(2, sources.dim()))
    with torch.no_grad():
        festimate = fmodel(mix).detach()
        sdrs = new_sdr(sources, festimate)
        sdrqs = new_sdr(sources, estimate.detach())
        w = torch.exp((sdrs - sdrqs) / 10)
    task = torch.nn.functional.l1_loss(estimate, sources, reduction="none").mean(dims).mean(0)
    kd = torch.nn.functional.l1_loss(estimate, festimate, reduction="none").mean(dims)
    kd = torch.mean(w * kd, dim=0)
    loss = (1 - kd_lambda) * task + kd_lambda * kd
    wt = torch.tensor(weights).to(sources)
    loss = (loss * wt).sum() / wt.sum()
    return estimate, festimate, w, task, kd, loss


def main():
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    torch.manual_seed(0)
    d = {}
    model = HTDemucsQ(**TINY_KW)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, QCFG)
    fill(model, "S."); fill(fmodel, "T.")
    model.train(); fmodel.eval()
    B, T = 2, 2100
    src = keyed_randn("hd.src", (B, 2, 2, T), 0.3)
    # band-limit a little so the spectrogram is not flat
    src = torch.nn.functional.avg_pool1d(src.reshape(B * 4, 1, T), 5, 1, 2).reshape(B, 2, 2, T) * 2.0
    mix = src.sum(1)
    d["mix"], d["src"] = npy(mix), npy(src)
    for k, v in fmodel.state_dict().items():
        d["fsd." + k] = npy(v)
    with torch.no_grad():
        fest = fmodel(mix)
    d["fest"] = npy(fest)
    for k, v in model.state_dict().items():
        d["sd0." + k] = npy(v)
    # the first observer call WITH a backward: every quantizer passes its input through (activations: observer branch; weights:
    # the first call only records), so this step checks every backward kernel free of quantizer flips.  Calls 2..50 forward only.
    est1, _, w1, task1, kd1, loss1 = kd_step(model, fmodel, mix, src)
    loss1.backward()
    d["o1.est"], d["o1.w"], d["o1.task"], d["o1.kd"], d["o1.loss"] = npy(est1), npy(w1), npy(task1), npy(kd1), npy(loss1)
    for k, p in model.named_parameters():
        if p.grad is not None:
            d["o1.grad." + k] = npy(p.grad)
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        for _ in range(49):
            est_obs = model(mix)
    d["est_obs"] = npy(est_obs)
    for k, v in model.state_dict().items():
        if k.endswith("_range"):
            d["sd_obs." + k] = npy(v)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, RQ.GradientActivationFakeQuantize):
                m.min_range.mul_(0.93); m.max_range.mul_(0.93)
    for k, v in model.state_dict().items():
        if k.endswith("_range"):
            d["sd." + k] = npy(v)
    est, festimate, w, task, kd, loss = kd_step(model, fmodel, mix, src)
    loss.backward()
    d["est"], d["w"], d["task"], d["kd"], d["loss"] = npy(est), npy(w), npy(task), npy(kd), npy(loss)
    for k, p in model.named_parameters():
        if p.grad is not None:
            d["grad." + k] = npy(p.grad)
    d["nograd"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
    np.savez_compressed(os.path.join(out, "hd_tiny_step.npz"), **d)
    print("hd_tiny_step:", len(d), "arrays; loss", float(loss), "est rms", float(est.pow(2).mean().sqrt()), "fest rms", float(fest.pow(2).mean().sqrt()))


if __name__ == "__main__":
    main()
