#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8 row a13 (cfg 3: DPTNet 2spk W8A8 QAT), generated from the REAL reference.

Runs ONLY in the build container (imports /root/reference through tools/ref_shim.py); the outputs are
data-only .npz fixtures under tests/golden/:

  dpt_layers.npz     the LayerQ classes that only the dual-path models reach, fwd + bwd at tiny shapes:
                     LayerNormQ (qat_layers.py:455-465), LinearQ (:521-536), LSTMQ (:571-600, bidirectional),
                     MultiheadAttentionQ (:865-950), Conv2dQ 1x1, Conv1dNlQ with Tanh / Sigmoid, MulQ (same shape and
                     the [B,1,E,L] x [B,S,E,L] masking broadcast of dptnetq.py:395), Conv1dEncoderQ k=2 s=1 + ReLU with
                     n_splitter=2 (:993-1039), LinearDecoderQ n_combiner=2 + the Linear ResidualErrorBlock
                     (:1256-1296, 1105-1190), overlap_and_add (dptnetq.py:17-58), split/merge_feature (:232-276)
  dpt_tiny_step.npz  a tiny DPTNetQ (1 dual-path layer... see TINY_KW): 53 QAT steps with mysystem.py:124-151 semantics
  cfg3_step.npz      the FULL-SIZE DPTNetQ (6 layers, 2.8 M parameters), B=1, name-keyed weights: digests of steps
                     1, 2, 51, 52 (loss, KD, task, SI-SDRs, clipped grad norm, per-parameter gradient norms)

Usage:  python tools/make_goldens_dptnet.py [--only layers|tiny|cfg3] [--T 8000]
"""
import argparse
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402  (installs the import shim, pins torch to one thread)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from quantization.qat import qat_quant as RQ  # noqa: E402
from quantization.qat import qat_layers as RL  # noqa: E402
from quantization.qat.models import dptnetq as RD  # noqa: E402
from quantization.qat.models.load_model import quantize_model, enable_observer  # noqa: E402
import process as RP  # noqa: E402

npy, keyed_randn = MG.npy, MG.keyed_randn
P = {'gradient_based': True, 'weight_quant': True, 'act_quant': True, 'act_n_bits': 8, 'weight_n_bits': 8}
TINY_KW = dict(n_spks=2, kernel_size=2, enc_dim=16, feature_dim=8, hidden_dim=12, layer=2, segment_size=10)


class First(nn.Module):
    """LSTMQ returns [y], MultiheadAttentionQ returns (y,): unwrap for the generic layer runner"""

    def __init__(self, m, n_in=1):
        super().__init__()
        self.m, self.n_in = m, n_in

    def forward(self, x):
        return self.m(*([x] * self.n_in))[0]


def fill(mod, prefix):
    """name-keyed deterministic fill of every non-range parameter"""
    with torch.no_grad():
        for k, p in mod.named_parameters():
            if k.endswith("min_range") or k.endswith("max_range"):
                continue
            if p.numel() == 1:
                p.fill_(0.25)
            elif p.dim() == 1 and "norm" in k and k.endswith("weight"):
                p.copy_(1.0 + keyed_randn(prefix + k, tuple(p.shape), 0.1))
            elif p.dim() == 1:
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 0.05))
            else:
                fan = max(1, int(np.prod(p.shape[1:])))
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 1.0 / np.sqrt(fan)))


def run_layer(name, layer, inputs, d, tight=0.93):
    """51 reference observer calls on the same input (weight observer: first call; activation EMA: 50 calls), ranges
    tightened so clipping is exercised, then ONE quantizing forward + backward."""
    layer.train()
    enable_observer(layer, True)
    with torch.no_grad():
        for _ in range(50):
            y_obs = layer(*inputs)
    d[f"{name}.out_obs"] = npy(y_obs)
    for k, v in layer.state_dict().items():
        d[f"{name}.sd_obs.{k}"] = npy(v)
    with torch.no_grad():
        for n, m in layer.named_modules():
            if isinstance(m, RQ.GradientActivationFakeQuantize):
                m.min_range.mul_(tight)
                m.max_range.mul_(tight)
    ins = [t.clone().requires_grad_(True) for t in inputs]
    y = layer(*ins)
    g = keyed_randn(name + ".gout", tuple(y.shape))
    y.backward(g)
    for i, t in enumerate(inputs):
        d[f"{name}.in{i}"] = npy(t)
        if ins[i].grad is not None:
            d[f"{name}.gin{i}"] = npy(ins[i].grad)
    d[f"{name}.out"], d[f"{name}.gout"] = npy(y), npy(g)
    for k, v in layer.state_dict().items():
        d[f"{name}.sd.{k}"] = npy(v)
    for k, p in layer.named_parameters():
        if p.grad is not None:
            d[f"{name}.grad.{k}"] = npy(p.grad)


def gen_layers(out):
    d = {}
    S, Bt, C, H = 9, 5, 16, 12
    x = keyed_randn("dpt.x", (S, Bt, C), 0.8)
    # LayerNormQ
    L = RL.LayerNormQ(nn.LayerNorm(C), gradient_based=True, act_quant=True); fill(L, "ln.")
    run_layer("layernormq", L, [x * 1.3 + 0.2], d)
    # LinearQ (2H -> C, the "improved transformer" output projection, dptnetq.py:69)
    L = RL.LinearQ(nn.Linear(2 * H, C), **P); fill(L, "lin.")
    run_layer("linearq", L, [keyed_randn("dpt.xlin", (S, Bt, 2 * H), 0.7).relu()], d)
    # LSTMQ bidirectional, seq-first (dptnetq.py:67)
    L = First(RL.LSTMQ(nn.LSTM(C, H, 1, bidirectional=True), **P)); fill(L, "lstm.")
    run_layer("lstmq", L, [x], d)
    # MultiheadAttentionQ (d_model C, 4 heads, dptnetq.py:64), self-attention
    L = First(RL.MultiheadAttentionQ(nn.MultiheadAttention(C, 4, dropout=0.0), **P), n_in=3); fill(L, "mha.")
    run_layer("mhaq", L, [x], d)
    # Conv2dQ 1x1 (DPT.output[1], dptnetq.py:187)
    L = RL.Conv2dQ(nn.Conv2d(C, 2 * C, 1), **P); fill(L, "c2d.")
    run_layer("conv2dq", L, [keyed_randn("dpt.x4", (2, C, 6, 5), 0.8)], d)
    # gated output convs (dptnetq.py:286-287)
    xc = keyed_randn("dpt.xc", (3, C, 37), 0.9)
    L = RL.Conv1dNlQ(nn.Conv1d(C, C, 1), nn.Tanh(), **P); fill(L, "ctanh.")
    run_layer("conv1dnlq_tanh", L, [xc], d)
    L = RL.Conv1dNlQ(nn.Conv1d(C, C, 1), nn.Sigmoid(), **P); fill(L, "csig.")
    run_layer("conv1dnlq_sigmoid", L, [xc], d)
    # MulQ same shape (dptnetq.py:306) and masking broadcast (dptnetq.py:395)
    L = RL.MulQ(RL.Mul(), gradient_based=True, act_quant=True)
    run_layer("mulq_same", L, [xc, keyed_randn("dpt.xc2", (3, C, 37), 0.6)], d)
    L = RL.MulQ(RL.Mul(), gradient_based=True, act_quant=True)
    run_layer("mulq_mask", L, [keyed_randn("dpt.feat", (3, 1, C, 37), 0.7).abs(), keyed_randn("dpt.mask", (3, 2, C, 37), 0.6).abs()], d)
    # AddQ on rank-3 seq-first tensors and NlQ(PReLU) on rank 4 (dptnetq.py:75, 187)
    L = RL.AddQ(RL.Add(), gradient_based=True, act_quant=True)
    run_layer("addq_seq", L, [x, keyed_randn("dpt.x2", (S, Bt, C), 0.5)], d)
    L = RL.NlQ(nn.PReLU(), gradient_based=True, act_quant=True); fill(L, "nlq.")
    run_layer("nlq_prelu4", L, [keyed_randn("dpt.x4", (2, C, 6, 5), 0.8)], d)
    # encoder: Conv1d(1, N, k=2, s=1, bias=False) + ReLU, n_splitter = 2 (dptnetq.py:116-117, 441)
    T = 61
    enc = nn.Conv1d(1, C, 2, stride=1, bias=False)
    L = RL.Conv1dEncoderQ([enc, nn.ReLU()], n_splitter=2, **P); fill(L, "enc.")
    wav = keyed_randn("dpt.wav", (3, 1, T), 0.2)
    run_layer("conv1dencoderq_k2", L, [RP.preprocess(wav.clone(), n_splitter=2)], d)
    # GroupNormQ(1, N) right after it, on [B, N, L]
    L = RL.GroupNormQ(nn.GroupNorm(1, C, eps=1e-8), gradient_based=True, act_quant=True); fill(L, "gn.")
    run_layer("groupnormq_enc", L, [keyed_randn("dpt.gnx", (3, C, T - 1), 0.5).relu()], d)
    # decoder: Linear(E, W=2, bias=False), n_combiner = 2 (dptnetq.py:136, 450) on [B, S, L, E]
    L = RL.LinearDecoderQ([nn.Linear(C, 2, bias=False)], n_combiner=2, gradient_based=True, weight_quant=True, weight_n_bits=8,
                          act_quant=True, act_n_bits=8, out_quant=True, out_act_n_bits=8)
    fill(L, "dec.")
    run_layer("lineardecoderq", L, [keyed_randn("dpt.decin", (3, 2, 37, C), 0.5).abs()], d)
    # data-movement functions of the model file
    sig = keyed_randn("dpt.ola", (2, 3, 11, 4), 1.0)
    d["ola.in"], d["ola.out_step2"], d["ola.out_step1"] = npy(sig), npy(RD.overlap_and_add(sig, 2)), npy(RD.overlap_and_add(sig[..., :2], 1))
    base = RD.DPT_base(4, 4, 4, num_spk=2, layer=1, segment_size=10)
    for Tn in (37, 40, 45):
        f = keyed_randn(f"dpt.seg{Tn}", (2, 4, Tn), 1.0)
        seg, rest = base.split_feature(f, 10)
        d[f"seg{Tn}.in"], d[f"seg{Tn}.out"], d[f"seg{Tn}.rest"] = npy(f), npy(seg), np.array(rest)
        d[f"seg{Tn}.merged"] = npy(base.merge_feature(seg, rest))
    np.savez_compressed(os.path.join(out, "dpt_layers.npz"), **d)
    print("dpt_layers:", len(d), "arrays")


def gen_tiny_step(out, n_steps=53):
    d = {}
    torch.manual_seed(0)
    model = RD.DPTNetQ(**TINY_KW)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, MG.QCFG)
    fill(model, "S."); fill(fmodel, "T.")
    model.train(); fmodel.eval()
    for k, v in model.state_dict().items():
        d[f"sd0.{k}"] = npy(v)
    for k, v in fmodel.state_dict().items():
        d[f"fsd.{k}"] = npy(v)
    d["sd_keys"] = np.array(list(model.state_dict().keys()))
    B, T = 2, 121
    x, tgt = MG.synth_batch(B, T, seed=3)
    d["x"], d["tgt"] = npy(x), npy(tgt)
    opt = torch.optim.Adam(model.parameters(), lr=4e-4)
    record = {1, 2, 50, 51, 52, 53}
    layer_names = [n for n, m in model.named_modules() if isinstance(m, RL.LayerQ)]
    d["layer_names"] = np.array(layer_names)

    def first(o):
        return o[0] if isinstance(o, (list, tuple)) else o

    for step in range(1, n_steps + 1):
        acts, hooks = {}, []
        if step == 51:      # per-layer inputs / outputs of the first fully quantizing step (teacher-forced G1 inside the network)
            for n, m in model.named_modules():
                if isinstance(m, RL.LayerQ):
                    hooks.append(m.register_forward_hook(
                        lambda mod, i, o, n=n: acts.__setitem__(n, (tuple(npy(t) for t in i if torch.is_tensor(t)), npy(first(o))))))
        opt.zero_grad()
        est, fest, w, kd, task, loss, sdrs, sdrqs = MG.common_step(model, fmodel, x, tgt)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        for h in hooks:
            h.remove()
        if step in record:
            p = f"s{step}."
            d[p + "est"], d[p + "fest"], d[p + "w"] = npy(est), npy(fest), npy(w)
            d[p + "kd"], d[p + "task"], d[p + "loss"], d[p + "gnorm"] = npy(kd), npy(task), npy(loss), npy(gnorm)
            for k, prm in model.named_parameters():
                if prm.grad is not None:
                    d[p + "grad." + k] = npy(prm.grad)
            d[p + "nograd"] = np.array([k for k, prm in model.named_parameters() if prm.grad is None])
            for n, (ins, o) in acts.items():
                d[p + "act." + n] = o
                for j, t in enumerate(ins):
                    d[p + f"actin{j}." + n] = t
        opt.step()
        if step in record:
            for k, v in model.state_dict().items():
                if k.endswith("min_range") or k.endswith("max_range") or step in (1, 50, 53):
                    d[f"s{step}.post_sd.{k}"] = npy(v)
    np.savez_compressed(os.path.join(out, "dpt_tiny_step.npz"), **d)
    print("dpt_tiny_step: final loss", float(loss), "keys", len(d))


def gen_cfg3_step(out, n_steps=52, B=1, T=8000, fname="cfg3_step.npz"):
    torch.set_num_threads(int(os.environ.get("FQSS_GOLDEN_THREADS", "8")))
    d = {}
    torch.manual_seed(0)
    model = RD.DPTNetQ(n_spks=2, kernel_size=2)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, MG.QCFG)
    fill(fmodel, "T."); fill(model, "S.")
    model.train(); fmodel.eval()
    d["B"], d["T"] = np.array(B), np.array(T)
    d["param_names"] = np.array([k for k, _ in model.named_parameters()])
    d["param_sum"] = np.array([float(p.double().sum()) for _, p in model.named_parameters()])
    d["param_sumsq"] = np.array([float((p.double() ** 2).sum()) for _, p in model.named_parameters()])
    d["tparam_names"] = np.array([k for k, _ in fmodel.named_parameters()])
    d["tparam_sum"] = np.array([float(p.double().sum()) for _, p in fmodel.named_parameters()])
    x, tgt = MG.synth_batch(B, T, seed=0)
    d["x_sum"], d["tgt_sumsq"] = np.float64(x.double().sum()), np.float64((tgt.double() ** 2).sum())
    opt = torch.optim.Adam(model.parameters(), lr=4e-4)
    record = {1, 2, 51, 52}
    for step in range(1, n_steps + 1):
        opt.zero_grad()
        est, fest, w, kd, task, loss, sdrs, sdrqs = MG.common_step(model, fmodel, x, tgt)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        if step in record:
            p = f"s{step}."
            d[p + "w"], d[p + "kd"], d[p + "task"], d[p + "loss"], d[p + "gnorm"] = npy(w), npy(kd), npy(task), npy(loss), npy(gnorm)
            d[p + "sdr_teacher"], d[p + "sdr_student"] = npy(sdrs), npy(sdrqs)
            if step in (1, 51):
                d[p + "est"] = npy(est).astype(np.float32)
                d[p + "fest_rms"] = np.float64(fest.double().pow(2).mean().sqrt())
            d[p + "grad_norm"] = np.array([float(q.grad.double().norm()) if q.grad is not None else -1.0
                                           for _, q in model.named_parameters()])
        opt.step()
        if step in record or step % 10 == 0:
            print("cfg3 step", step, "loss", float(loss), flush=True)
    np.savez_compressed(os.path.join(out, fname), **d)
    torch.set_num_threads(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    ap.add_argument("--T", type=int, default=8000)
    ap.add_argument("--fname", default="cfg3_step.npz", help="cfg3 digest file (cfg3_full_step.npz for the BASELINE size: --T 24000)")
    a = ap.parse_args()
    if a.only in ("", "layers"):
        gen_layers(a.out)
    if a.only in ("", "tiny"):
        gen_tiny_step(a.out)
    if a.only in ("", "cfg3"):
        gen_cfg3_step(a.out, T=a.T, fname=a.fname)


if __name__ == "__main__":
    main()
