#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (/root/reference) for the FQSS QAT hot path.

Runs ONLY in the build container (the reference never travels to the GPU box).  The outputs are
small .npz fixtures committed under tests/golden/ — data only (inputs + expected outputs).

What is pinned (SURVEY.md §8(c) fixture plan F1-F7):
  fq_act.npz      qat_quant.linear_quantize asym  (qat_quant.py:136-147)   y / u8 idx / gx / gmin / gmax
  fq_w.npz        qat_quant.linear_quantize sym   (qat_quant.py:126-135)   per-channel, ch_out_idx 0 and 1
  observer.npz    GradientActivationFakeQuantize  (qat_quant.py:227-242)   51-call EMA sequence
                  GradientWeightFakeQuantize      (qat_quant.py:372-381)   one-shot amax/amin
  process.npz     process.preprocess/postprocess/quantize (process.py:10-52)
  layers.npz      each LayerQ class used by ConvTasNetQ at tiny shapes (qat_layers.py) fwd + bwd
  loss.npz        wsdr.PairwiseWSDR (wsdr.py:46-95) + restated asteroid PIT -> w, kd, task, loss
  tiny_step.npz   tiny ConvTasNetQ: 53 full QAT steps (mysystem.py:124-151 semantics), observer
                  phase + quantizing phase: est, loss, grads, per-layer activations, final state
  cfg1_step.npz   FULL-SIZE ConvTasNetQ at cfg 1 (B=2, T=8000), name-keyed deterministic weights: steps 1, 2,
                  51, 52 -> loss / KD / task / weights / SI-SDRs / clipped grad norm / per-parameter grad norms
  cfg2_step.npz   the same at cfg 2 (B=8, T=32000: the benchmark's size), digests only (--only cfg2)

Usage:  python tools/make_goldens.py [--out tests/golden]
"""
import argparse
import copy
import itertools
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

torch.set_num_threads(1)  # fixed summation order for reproducible goldens

from quantization.qat import qat_quant as RQ  # noqa: E402
from quantization.qat import qat_layers as RL  # noqa: E402
from quantization.qat.models.convtasnetq import ConvTasNetQ  # noqa: E402
from quantization.qat.models.load_model import quantize_model, enable_observer  # noqa: E402
import process as RP  # noqa: E402

# the asteroid env's wsdr.py has no third-party import -> exec-load it standalone
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "ref_wsdr", "/root/reference/train_env/asteroid_librimix/wsdr.py")
RW = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(RW)

EPS = 1e-8
QCFG = {'qat': True, 'gradient_based': True, 'weight_quant': True, 'weight_n_bits': 8,
        'act_quant': True, 'act_n_bits': 8, 'in_quant': False, 'in_act_n_bits': 8,
        'out_quant': True, 'out_act_n_bits': 8, 'n_splitter': 2, 'n_combiner': 2, 'observer': True}


def npy(t):
    return t.detach().cpu().numpy().copy()


def keyed_randn(key, shape, scale=1.0):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return torch.randn(*shape, generator=g) * scale


# --------------------------------------------------------------------------------------------
# restated asteroid-0.6.0 pieces that are NOT in /root/reference ("parity unpinned"
# for the third-party part; the in-repo wsdr.py IS used for the SDR math itself)
# --------------------------------------------------------------------------------------------
def pit_min_mean(pw):
    """asteroid PITLossWrapper(pit_from='pw_mtx') restated: pw[b, est_i, tgt_j] ->
    mean_b min_perm mean_i pw[b, perm[i], i]."""
    n_src = pw.shape[-1]
    perms = list(itertools.permutations(range(n_src)))
    loss_set = torch.stack(
        [sum(pw[:, p[i], i] for i in range(n_src)) / n_src for p in perms], dim=1)
    min_loss, _ = torch.min(loss_set, dim=1)
    return torch.mean(min_loss)


_pw_log = RW.PairwiseWSDR("sisdr", take_log=True)


def pairwise_neg_sisdr(est, tgt):
    return -_pw_log(est, tgt)


def common_step(model, fmodel, x, targets, kd_lambda=0.1):
    """mysystem.py:124-151 restated around the REAL model / wsdr code."""
    est = model(x)
    with torch.no_grad():
        fest = fmodel(x).detach()
        sdrs, sdrqs = [], []
        for b in range(len(fest)):
            sdrs.append(pit_min_mean(pairwise_neg_sisdr(fest[b:b + 1], targets[b:b + 1])))
            sdrqs.append(pit_min_mean(pairwise_neg_sisdr(est[b:b + 1].detach(), targets[b:b + 1])))
        sdrs, sdrqs = torch.stack(sdrs), torch.stack(sdrqs)
        w = 10 ** ((sdrs - sdrqs) / 10)
    kd = -pit_min_mean(RW.pairwise_wsisdr(est, fest, weights=w))
    task = -pit_min_mean(RW.pairwise_wsisdr(est, targets))
    loss = -10 * torch.log10((1 - kd_lambda) * task + kd_lambda * kd + EPS)
    return est, fest, w, kd, task, loss, sdrs, sdrqs


# --------------------------------------------------------------------------------------------
def gen_fq_act(out):
    d = {}
    cases = [(-1.3, 1.7), (0.0, 6.0), (-0.5, 0.5), (-0.0371, 0.2113)]
    for ci, (lo, hi) in enumerate(cases):
        x = keyed_randn(f"fq_act.x{ci}", (3, 7, 101), scale=(hi - lo) * 0.45) + 0.5 * (hi + lo)
        # add exact half-bin edges and out-of-range values
        delta = (hi - lo) / 255.0
        edges = torch.tensor([lo + (k + 0.5) * delta for k in range(0, 255, 17)], dtype=torch.float32)
        x.view(-1)[:edges.numel()] = edges
        x.view(-1)[edges.numel():edges.numel() + 4] = torch.tensor([lo, hi, lo - 1.0, hi + 1.0])
        g = keyed_randn(f"fq_act.g{ci}", tuple(x.shape))
        xr = x.clone().requires_grad_(True)
        mn = torch.tensor([lo], dtype=torch.float32, requires_grad=True)
        mx = torch.tensor([hi], dtype=torch.float32, requires_grad=True)
        y = RQ.linear_quantize(xr, mn, mx, 8, sign=True, sym=False)
        y.backward(g)
        with torch.no_grad():
            dl = (mx - mn) / 255
            idx = torch.clip(torch.round((x - mn) / dl), 0, 255).to(torch.uint8)
        d[f"x{ci}"], d[f"g{ci}"], d[f"range{ci}"] = npy(x), npy(g), np.array([lo, hi], np.float32)
        d[f"y{ci}"], d[f"idx{ci}"] = npy(y), npy(idx)
        d[f"gx{ci}"], d[f"gmin{ci}"], d[f"gmax{ci}"] = npy(xr.grad), npy(mn.grad), npy(mx.grad)
    d["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(out, "fq_act.npz"), **d)


def gen_fq_w(out):
    d = {}
    shapes = [((12, 5, 3), 0), ((6, 9, 1), 0), ((7, 1, 16), 1), ((5, 4, 16), 1)]
    for ci, (shape, axis) in enumerate(shapes):
        w = keyed_randn(f"fq_w.w{ci}", shape, 0.2)
        g = keyed_randn(f"fq_w.g{ci}", shape)
        q = RQ.GradientWeightFakeQuantize(True, shape, n_bits=8, ch_out_idx=axis)
        w0 = q(w)  # observer call: records amax/amin, returns w unquantized
        assert torch.equal(w0, w)
        d[f"obs_min{ci}"], d[f"obs_max{ci}"] = npy(q.min_range), npy(q.max_range)
        # perturb the learned ranges so that clipping and |min|<>|max| branches are exercised
        with torch.no_grad():
            q.min_range.mul_(0.8)
            q.max_range.mul_(0.9)
            q.min_range.view(-1)[0] = -q.max_range.view(-1)[0]  # exact tie |min| == |max|
        wr = w.clone().requires_grad_(True)
        y = q(wr)
        y.backward(g)
        with torch.no_grad():
            a = torch.maximum(q.min_range.abs(), q.max_range.abs())
            dl = 2 * a / 255
            idx = torch.clip(torch.round(w / dl), -128, 127).to(torch.int8)
        d[f"w{ci}"], d[f"g{ci}"], d[f"axis{ci}"] = npy(w), npy(g), np.array(axis)
        d[f"min{ci}"], d[f"max{ci}"] = npy(q.min_range), npy(q.max_range)
        d[f"y{ci}"], d[f"idx{ci}"], d[f"gw{ci}"] = npy(y), npy(idx), npy(wr.grad)
        d[f"gmin{ci}"], d[f"gmax{ci}"] = npy(q.min_range.grad), npy(q.max_range.grad)
    d["n_cases"] = np.array(len(shapes))
    np.savez_compressed(os.path.join(out, "fq_w.npz"), **d)


def gen_observer(out):
    d = {}
    q = RQ.GradientActivationFakeQuantize(True, n_bits=8)
    q.train()
    xs, ys, mins, maxs = [], [], [], []
    for it in range(53):
        x = keyed_randn(f"observer.x{it}", (2, 5, 33), 0.3 + 0.01 * it) + 0.05
        y = q(x)
        xs.append(npy(x)); ys.append(npy(y)); mins.append(npy(q.min_range)); maxs.append(npy(q.max_range))
    d["x"], d["y"], d["min"], d["max"] = map(np.stack, (xs, ys, mins, maxs))
    d["n_iter"] = np.array(q.n_iter)
    np.savez_compressed(os.path.join(out, "observer.npz"), **d)


def gen_process(out):
    d = {}
    x = keyed_randn("process.x", (3, 1, 257), 0.2)
    x.view(-1)[0] = 0.73  # global max, positive
    d["x"] = npy(x)
    d["pre2"] = npy(RP.preprocess(x.clone(), n_splitter=2))
    d["pre1"] = npy(RP.preprocess(x.clone(), n_splitter=1))
    x2 = keyed_randn("process.x2d", (3, 257), 0.2)
    x2.view(-1)[5] = -0.91  # global max is a negative sample
    d["x2d"], d["pre2_2d"] = npy(x2), npy(RP.preprocess(x2.clone(), n_splitter=2))
    q_in = keyed_randn("process.q", (4, 100), 0.7)
    d["q_in"], d["q_out"] = npy(q_in), npy(RP.quantize(q_in))
    z = keyed_randn("process.z", (2, 3, 2, 1, 129), 0.3)
    d["post_in"] = npy(z)
    d["post2"] = npy(RP.postprocess(z.clone(), n_combiner=2))
    z1 = keyed_randn("process.z1", (1, 3, 2, 1, 129), 0.3)
    d["post_in1"], d["post1"] = npy(z1), npy(RP.postprocess(z1.clone(), n_combiner=1))
    np.savez_compressed(os.path.join(out, "process.npz"), **d)


def _fill_module(mod, prefix):
    """deterministic, name-keyed parameter fill (Q6: reference init is RNG-order dependent)."""
    with torch.no_grad():
        for k, p in mod.named_parameters():
            if k.endswith("min_range") or k.endswith("max_range"):
                continue
            if k.endswith("nl.weight"):
                p.copy_(torch.full_like(p, 0.25) + keyed_randn(prefix + k, tuple(p.shape), 0.05))
            elif "groupnorm.weight" in k:
                p.copy_(1.0 + keyed_randn(prefix + k, tuple(p.shape), 0.1))
            else:
                fan = max(1, int(np.prod(p.shape[1:]))) if p.dim() > 1 else 4
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 1.0 / np.sqrt(fan)))


def _run_layer(name, layer, inputs, d, calib=None):
    """calibrate ranges with the reference observer (1 call), then quantizing fwd+bwd."""
    layer.train()
    enable_observer(layer, True)
    with torch.no_grad():
        y_obs = layer(*inputs)  # weights: one-shot observer; activations: EMA step 1
    # make activation ranges meaningful: set them from the observed tensor, slightly tight
    for n, m in layer.named_modules():
        if isinstance(m, RQ.GradientActivationFakeQuantize):
            m.n_iter = m.max_observations  # leave observer phase
    if calib is not None:
        calib(layer, y_obs)
    ins = [t.clone().requires_grad_(True) if t.dtype.is_floating_point else t for t in inputs]
    y = layer(*ins)
    g = keyed_randn(name + ".gout", tuple(y.shape))
    y.backward(g)
    for i, t in enumerate(inputs):
        d[f"{name}.in{i}"] = npy(t)
        if ins[i].grad is not None:
            d[f"{name}.gin{i}"] = npy(ins[i].grad)
    d[f"{name}.out_obs"] = npy(y_obs)
    d[f"{name}.out"], d[f"{name}.gout"] = npy(y), npy(g)
    for k, v in layer.state_dict().items():
        d[f"{name}.sd.{k}"] = npy(v)
    for k, p in layer.named_parameters():
        if p.grad is not None:
            d[f"{name}.grad.{k}"] = npy(p.grad)


def _set_act_range(q, t, tight=0.9):
    with torch.no_grad():
        q.min_range.fill_(float(t.min()) * tight)
        q.max_range.fill_(float(t.max()) * tight)


def gen_layers(out):
    d = {}
    P = {'gradient_based': True, 'weight_quant': True, 'act_quant': True, 'act_n_bits': 8, 'weight_n_bits': 8}
    B, C, H, M = 2, 16, 24, 77

    def calib_single(layer, y):
        _set_act_range(layer.activation_fake_quantize, y)

    x_bn = keyed_randn("layers.x_bn", (B, C, M), 0.8)
    x_hid = keyed_randn("layers.x_hid", (B, H, M), 0.8)

    # Conv1dQ 1x1 (res/skip/bottleneck conv)  qat_layers.py:124-153
    conv = nn.Conv1d(H, C, 1); L = RL.Conv1dQ(conv, **P); _fill_module(L, "pw_q.")
    _run_layer("conv1dq_pw", L, [x_hid], d, calib_single)
    # Conv1dNlQ 1x1 + PReLU  qat_layers.py:188-219
    conv = nn.Conv1d(C, H, 1); L = RL.Conv1dNlQ(conv, nn.PReLU(), **P); _fill_module(L, "pw_nl.")
    _run_layer("conv1dnlq_pw_prelu", L, [x_bn], d, calib_single)
    # Conv1dNlQ 1x1 + ReLU (mask_net.1)
    conv = nn.Conv1d(C, 2 * H, 1); L = RL.Conv1dNlQ(conv, nn.ReLU(), **P); _fill_module(L, "pw_relu.")
    _run_layer("conv1dnlq_pw_relu", L, [x_bn], d, calib_single)
    # depthwise k3 dilated + PReLU (convtasnetq.py:28-30)
    for dil in (1, 4):
        conv = nn.Conv1d(H, H, 3, padding=dil, dilation=dil, groups=H)
        L = RL.Conv1dNlQ(conv, nn.PReLU(), **P); _fill_module(L, f"dw{dil}.")
        _run_layer(f"conv1dnlq_dw_d{dil}", L, [x_hid], d, calib_single)
    # GroupNormQ(1, C, eps 1e-8)  qat_layers.py:438-452
    gn = nn.GroupNorm(1, H, eps=1e-8); L = RL.GroupNormQ(gn, gradient_based=True, act_quant=True)
    _fill_module(L, "gn.")
    _run_layer("groupnormq", L, [x_hid * 1.7 + 0.3], d, calib_single)
    # AddQ / MulQ / NlQ
    L = RL.AddQ(RL.Add(), gradient_based=True, act_quant=True)
    _run_layer("addq", L, [x_bn, keyed_randn("layers.add_b", (B, C, M), 0.5)], d, calib_single)
    L = RL.MulQ(RL.Mul(), gradient_based=True, act_quant=True)
    mask = keyed_randn("layers.mask", (B, 2, H, M), 0.6).abs()
    _run_layer("mulq", L, [mask, x_hid.unsqueeze(1)], d, calib_single)
    L = RL.NlQ(nn.PReLU(), gradient_based=True, act_quant=True); _fill_module(L, "nlq.")
    _run_layer("nlq_prelu", L, [x_bn], d, calib_single)
    # Conv1dEncoderQ n_splitter=2  qat_layers.py:993-1039
    T = 8 * (M - 1) + 16
    enc = nn.Conv1d(1, H, 16, stride=8, bias=False)
    L = RL.Conv1dEncoderQ([enc], n_splitter=2, **P); _fill_module(L, "enc.")
    wav = keyed_randn("layers.wav", (B, 1, T), 0.2)
    _run_layer("conv1dencoderq", L, [RP.preprocess(wav.clone(), n_splitter=2)], d, calib_single)
    # ConvTr1dDecoderQ n_combiner=2 + ResidualErrorBlock  qat_layers.py:1305-1354, 1105-1202
    dec = nn.ConvTranspose1d(H, 1, 16, stride=8, bias=False)
    L = RL.ConvTr1dDecoderQ([dec], n_combiner=2, gradient_based=True, weight_quant=True, weight_n_bits=8,
                            act_quant=True, act_n_bits=8, out_quant=True, out_act_n_bits=8)
    _fill_module(L, "dec.")

    def calib_dec(layer, y):
        _set_act_range(layer.activation_fake_quantize, y[0])
        _set_act_range(layer.activation_fake_quantize_residual, y[1])
        # residual quantizer range: observed on Y - Y_q; re-run pieces of the reference forward
        with torch.no_grad():
            Yq = torch.nn.functional.conv1d(y[0], layer.residual_error_block.residual_encoder.weight, stride=8)
            _set_act_range(layer.residual_error_block.activation_fake_quantize, dec_in - Yq)

    dec_in = keyed_randn("layers.dec_in", (B * 2, H, M), 0.5)
    _run_layer("convtr1ddecoderq", L, [dec_in], d, calib_dec)
    np.savez_compressed(os.path.join(out, "layers.npz"), **d)


def gen_loss(out):
    d = {}
    B, S, T = 3, 2, 400
    tgt = keyed_randn("loss.tgt", (B, S, T), 0.1)
    est = (tgt + keyed_randn("loss.noise_q", (B, S, T), 0.05)).requires_grad_(True)
    fest = tgt + keyed_randn("loss.noise_f", (B, S, T), 0.03)
    # sample 1: swap speakers in the student so PIT picks the other permutation
    with torch.no_grad():
        est[1] = est[1].flip(0)

    class M(nn.Module):
        def __init__(s, v): super().__init__(); s.v = v
        def forward(s, x): return s.v
    e, f, w, kd, task, loss, sdrs, sdrqs = common_step(M(est), M(fest), None, tgt)
    loss.backward()
    d.update(est=npy(est), fest=npy(fest), tgt=npy(tgt), w=npy(w), kd=npy(kd), task=npy(task),
             loss=npy(loss), sdrs=npy(sdrs), sdrqs=npy(sdrqs), gest=npy(est.grad))
    d["pw_neg_sisdr"] = npy(pairwise_neg_sisdr(est.detach(), tgt))
    d["pw_wsisdr"] = npy(RW.pairwise_wsisdr(est.detach(), fest, weights=w))
    np.savez_compressed(os.path.join(out, "loss.npz"), **d)


def synth_batch(B, T, seed=0):
    """cfg-1 style synthetic 2-speaker mixtures: 0.05*randn band-limited by a 5-tap FIR."""
    g = torch.Generator().manual_seed(seed)
    s = 0.05 * torch.randn(B, 2, T + 4, generator=g)
    fir = torch.tensor([0.1, 0.25, 0.3, 0.25, 0.1]).view(1, 1, 5)
    s = torch.nn.functional.conv1d(s.view(B * 2, 1, T + 4), fir).view(B, 2, T)
    return s.sum(1, keepdim=True), s


def gen_tiny_step(out, n_steps=53):
    d = {}
    torch.manual_seed(0)
    kw = dict(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)
    model = ConvTasNetQ(**kw)
    fmodel = copy.deepcopy(model)           # train_utils.py:25  teacher = float copy BEFORE quantization
    model = quantize_model(model, QCFG)     # load_model.py:53-74
    model.train(); fmodel.eval()
    for k, v in model.state_dict().items():
        d[f"sd0.{k}"] = npy(v)
    for k, v in fmodel.state_dict().items():
        d[f"fsd.{k}"] = npy(v)
    d["sd_keys"] = np.array(list(model.state_dict().keys()))
    B, T = 2, 800
    x, tgt = synth_batch(B, T, seed=0)
    d["x"], d["tgt"] = npy(x), npy(tgt)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    record = {1, 2, 50, 51, 52, 53}
    layer_names = [n for n, m in model.named_modules() if isinstance(m, RL.LayerQ)]
    d["layer_names"] = np.array(layer_names)
    for step in range(1, n_steps + 1):
        acts = {}
        hooks = []
        if step in (1, 51):
            for n, m in model.named_modules():
                if isinstance(m, RL.LayerQ):
                    hooks.append(m.register_forward_hook(
                        lambda mod, i, o, n=n: acts.__setitem__(n, (tuple(npy(t) for t in i if torch.is_tensor(t)), npy(o)))))
        opt.zero_grad()
        est, fest, w, kd, task, loss, sdrs, sdrqs = common_step(model, fmodel, x, tgt)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)  # PL gradient_clip_val=5.0
        for h in hooks:
            h.remove()
        if step in record:
            p = f"s{step}."
            d[p + "est"], d[p + "fest"], d[p + "w"] = npy(est), npy(fest), npy(w)
            d[p + "kd"], d[p + "task"], d[p + "loss"], d[p + "gnorm"] = npy(kd), npy(task), npy(loss), npy(gnorm)
            for k, prm in model.named_parameters():
                if prm.grad is not None:
                    d[p + "grad." + k] = npy(prm.grad)
            for n, (ins, o) in acts.items():
                d[p + "act." + n] = o
                for j, t in enumerate(ins):
                    d[p + f"actin{j}." + n] = t
        opt.step()
        if step in record:
            for k, v in model.state_dict().items():
                if k.endswith("min_range") or k.endswith("max_range") or step in (1, 50, 53):
                    d[f"s{step}.post_sd.{k}"] = npy(v)
    np.savez_compressed(os.path.join(out, "tiny_step.npz"), **d)
    print("tiny_step: final loss", float(loss), "keys", len(d))


def cfg1_fill(model, prefix):
    """Name-keyed deterministic fill of every non-range parameter (the reference's own init draws
    from the global RNG in construction order).  The SAME function is restated in tests/helpers_cfg1.py."""
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("min_range") or k.endswith("max_range"):
                continue
            if p.numel() == 1:                                   # PReLU slope
                p.fill_(0.25)
            elif p.dim() == 1 and ("norm" in k.lower() or k.split(".")[-2].isdigit()) and k.endswith("weight"):
                p.copy_(1.0 + keyed_randn(prefix + k, tuple(p.shape), 0.1))      # GroupNorm gain
            elif p.dim() == 1:
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 0.02))           # biases / GroupNorm shift
            else:
                fan = max(1, int(np.prod(p.shape[1:])))
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 1.0 / np.sqrt(fan)))


def gen_cfg1_step(out, n_steps=52, B=2, T=8000, fname="cfg1_step.npz", keep_est=True):
    """F7: the FULL-SIZE ConvTasNetQ (5.1 M parameters) at cfg 1 (B=2, T=8000) -- or, with B=8 / T=32000, at cfg 2, the
    benchmark's own size: steps 1, 2 (observer phase) and 51, 52 (quantizing phase).  Weights come from cfg1_fill, so
    nothing but digests is stored."""
    torch.set_num_threads(8)
    d = {}
    torch.manual_seed(0)
    model = ConvTasNetQ(n_spks=2, kernel_size=16, stride=8)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, QCFG)
    cfg1_fill(fmodel, "T.")
    cfg1_fill(model, "S.")
    model.train(); fmodel.eval()
    d["param_names"] = np.array([k for k, _ in model.named_parameters()])
    d["param_sum"] = np.array([float(p.double().sum()) for _, p in model.named_parameters()])
    d["param_sumsq"] = np.array([float((p.double() ** 2).sum()) for _, p in model.named_parameters()])
    d["tparam_names"] = np.array([k for k, _ in fmodel.named_parameters()])
    d["tparam_sum"] = np.array([float(p.double().sum()) for _, p in fmodel.named_parameters()])
    x, tgt = synth_batch(B, T, seed=0)
    d["x_sum"], d["tgt_sumsq"] = np.float64(x.double().sum()), np.float64((tgt.double() ** 2).sum())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    record = {1, 2, 51, 52}
    for step in range(1, n_steps + 1):
        opt.zero_grad()
        est, fest, w, kd, task, loss, sdrs, sdrqs = common_step(model, fmodel, x, tgt)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        if step in record:
            p = f"s{step}."
            d[p + "w"], d[p + "kd"], d[p + "task"], d[p + "loss"], d[p + "gnorm"] = npy(w), npy(kd), npy(task), npy(loss), npy(gnorm)
            d[p + "sdr_teacher"], d[p + "sdr_student"] = npy(sdrs), npy(sdrqs)
            if step in (1, 51) and keep_est:
                d[p + "est"] = npy(est).astype(np.float32)
            if step in (1, 51):
                d[p + "fest_rms"] = np.float64(fest.double().pow(2).mean().sqrt())
            # per-parameter gradient norms AFTER clipping (what the optimizer sees)
            d[p + "grad_norm"] = np.array([float(q.grad.double().norm()) if q.grad is not None else -1.0
                                           for _, q in model.named_parameters()])
        opt.step()
        if step in record or step % 10 == 0:
            print("cfg1 step", step, "loss", float(loss), flush=True)
    np.savez_compressed(os.path.join(out, fname), **d)
    torch.set_num_threads(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="", help="generate just this fixture (e.g. cfg1)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.only == "cfg1":
        gen_cfg1_step(a.out)
        return
    if a.only == "cfg2":     # ~10 s per reference step on 8 cores
        gen_cfg1_step(a.out, B=8, T=32000, fname="cfg2_step.npz", keep_est=False)
        return
    gen_fq_act(a.out); gen_fq_w(a.out); gen_observer(a.out); gen_process(a.out)
    gen_layers(a.out); gen_loss(a.out); gen_tiny_step(a.out); gen_cfg1_step(a.out)
    for f in sorted(os.listdir(a.out)):
        print(f, os.path.getsize(os.path.join(a.out, f)))


if __name__ == "__main__":
    main()
