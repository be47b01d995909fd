#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8 row a14 (cfg 4: Sepformer 2spk W8A8 QAT), generated from the REAL reference.

Runs ONLY in the build container (imports /root/reference through tools/ref_shim.py); data-only .npz fixtures:

  sep_layers.npz     what the Sepformer graph adds to the layer set: ConstQ on the positional encoding + the broadcasting AddQ
                     (sepformerq.py:39-47, 117-118), GroupNormQ on a 4-D [B, F, K, S] tensor (:159, 175), NlQ(ReLU) (:61),
                     ConvTr1dDecoderQ with train_res_dec=True (qat_layers.py:1137-1146, 1194-1202)
  sep_tiny_step.npz  a tiny SepformerQ (TINY_KW), B = 1: 53 QAT steps
  cfg4_step.npz      the FULL-SIZE SepformerQ (25.7 M parameters), B = 1 (the per-GPU batch of the shipped YAML), name-keyed
                     weights: digests of steps 1, 2, 51, 52

Loss: the speechbrain env (speechbrain_librimix_trainer.py:99-115, wsdr.py:13-110) needs speechbrain, which is not installed; at
B = 1 -- the shipped per-GPU batch -- its KD objective is term for term the asteroid env's (mysystem.py:124-151: same zero-mean
SI-SNR ratio, same PIT, same -10 log10((1-l) task + l kd + eps); the per-sample weight vector has one entry and the batch mean is
over one element), so the fixtures use tools/make_goldens.py::common_step.  The speechbrain PIT wrapper / get_mask are third-party
(requirements.txt:15): parity unpinned for them, as recorded in SURVEY.md §8(c).

Usage:  python tools/make_goldens_sepformer.py [--only layers|tiny|cfg4] [--T 16000]
"""
import argparse
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402
import make_goldens_dptnet as MD  # noqa: E402

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from quantization.qat import qat_layers as RL  # noqa: E402
from quantization.qat.models import sepformerq as RS  # noqa: E402
from quantization.qat.models.load_model import quantize_model  # noqa: E402

npy, keyed_randn, fill, run_layer = MG.npy, MG.keyed_randn, MD.fill, MD.run_layer
TINY_KW = dict(n_spks=2, kernel_size=16, stride=8, n_filters=16, n_repeats=1, n_heads=4, chunk_size=10)
LR = 1.5e-4
TINY_FFN = 32


class PosAdd(nn.Module):
    """TransformerBlock's first two lines (sepformerq.py:117-118) with both modules quantized"""

    def __init__(self, F_):
        super().__init__()
        self.pos = RS.PositionalEncoding(F_)
        self.pos.const = RL.ConstQ(self.pos.const, gradient_based=True, act_quant=True)
        self.pos_add = RL.AddQ(RL.Add(), gradient_based=True, act_quant=True)

    def forward(self, x):
        return self.pos_add(x, self.pos(x))


def gen_layers(out):
    d = {}
    P = MD.P
    B, F_, K, S = 2, 16, 6, 5
    L = PosAdd(F_)
    run_layer("pos_add", L, [keyed_randn("sep.xpos", (7, 9, F_), 0.8)], d)            # [B', L, F] batch-first like the reference
    L = RL.GroupNormQ(nn.GroupNorm(1, F_, eps=1e-8), gradient_based=True, act_quant=True); fill(L, "gn4.")
    run_layer("groupnormq_4d", L, [keyed_randn("sep.x4", (B, F_, K, S), 0.9) + 0.2], d)
    L = RL.NlQ(nn.ReLU(), gradient_based=True, act_quant=True)
    run_layer("nlq_relu", L, [keyed_randn("sep.xr", (9, 7, 24), 0.8)], d)
    dec = nn.ConvTranspose1d(F_, 1, 16, stride=8, bias=False)
    L = RL.ConvTr1dDecoderQ([dec], n_combiner=2, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8,
                            out_quant=True, out_act_n_bits=8, train_res_dec=True)
    fill(L, "dec.")
    run_layer("convtr1ddecoderq_trd", L, [keyed_randn("sep.decin", (4, F_, 37), 0.5).abs()], d)
    np.savez_compressed(os.path.join(out, "sep_layers.npz"), **d)
    print("sep_layers:", len(d), "arrays")


def _steps(model, fmodel, x, tgt, n_steps, record, d, full):
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    for step in range(1, n_steps + 1):
        opt.zero_grad()
        est, fest, w, kd, task, loss, sdrs, sdrqs = MG.common_step(model, fmodel, x, tgt)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        if step in record:
            p = f"s{step}."
            d[p + "w"], d[p + "kd"], d[p + "task"], d[p + "loss"], d[p + "gnorm"] = npy(w), npy(kd), npy(task), npy(loss), npy(gnorm)
            d[p + "sdr_teacher"], d[p + "sdr_student"] = npy(sdrs), npy(sdrqs)
            if full:
                if step in (1, 51):
                    d[p + "est"] = npy(est).astype(np.float32)
                d[p + "grad_norm"] = np.array([float(q.grad.double().norm()) if q.grad is not None else -1.0
                                               for _, q in model.named_parameters()])
            else:
                d[p + "est"], d[p + "fest"] = npy(est), npy(fest)
                for k, prm in model.named_parameters():
                    if prm.grad is not None and step in (1, 2, 51):
                        d[p + "grad." + k] = npy(prm.grad)
                d[p + "nograd"] = np.array([k for k, prm in model.named_parameters() if prm.grad is None])
        opt.step()
        if not full and step in (1, 50):
            for k, v in model.state_dict().items():
                d[f"s{step}.post_sd.{k}"] = npy(v)
        if full and (step in record or step % 10 == 0):
            print("cfg4 step", step, "loss", float(loss), flush=True)
    return float(loss)


def gen_tiny_step(out, n_steps=53):
    d = {}
    torch.manual_seed(0)
    model = RS.SepformerQ(**TINY_KW)
    # the feed-forward width is not a SepformerQ argument (MaskGenerator's default 1024): swap in a narrow masker so that the
    # fixture stays small; same classes, same code paths
    model.masker = RS.MaskGenerator(TINY_KW["n_spks"], TINY_KW["n_filters"], n_repeats=TINY_KW["n_repeats"], n_heads=TINY_KW["n_heads"],
                                    chunk_size=TINY_KW["chunk_size"], n_ffn=TINY_FFN)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, MG.QCFG)
    fill(model, "S."); fill(fmodel, "T.")
    model.train(); fmodel.eval()
    for k, v in model.state_dict().items():
        d[f"sd0.{k}"] = npy(v)
    for k, v in fmodel.state_dict().items():
        d[f"fsd.{k}"] = npy(v)
    d["sd_keys"] = np.array(list(model.state_dict().keys()))
    x, tgt = MG.synth_batch(1, 800, seed=5)
    d["x"], d["tgt"] = npy(x), npy(tgt)
    last = _steps(model, fmodel, x, tgt, n_steps, {1, 2, 50, 51, 52, 53}, d, full=False)
    np.savez_compressed(os.path.join(out, "sep_tiny_step.npz"), **d)
    print("sep_tiny_step: final loss", last, "keys", len(d))


def gen_cfg4_step(out, n_steps=52, B=1, T=16000, fname="cfg4_step.npz"):
    torch.set_num_threads(int(os.environ.get("FQSS_GOLDEN_THREADS", "8")))
    d = {}
    torch.manual_seed(0)
    model = RS.SepformerQ(n_spks=2, kernel_size=16, stride=8)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, MG.QCFG)
    fill(fmodel, "T."); fill(model, "S.")
    model.train(); fmodel.eval()
    d["B"], d["T"] = np.array(B), np.array(T)
    d["param_names"] = np.array([k for k, _ in model.named_parameters()])
    d["param_sum"] = np.array([float(p.double().sum()) for _, p in model.named_parameters()])
    d["tparam_names"] = np.array([k for k, _ in fmodel.named_parameters()])
    x, tgt = MG.synth_batch(B, T, seed=0)
    d["x_sum"] = np.float64(x.double().sum())
    _steps(model, fmodel, x, tgt, n_steps, {1, 2, 51, 52}, d, full=True)
    np.savez_compressed(os.path.join(out, fname), **d)
    torch.set_num_threads(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="")
    ap.add_argument("--T", type=int, default=16000)
    ap.add_argument("--fname", default="cfg4_step.npz", help="cfg4 digest file (cfg4_full_step.npz for the BASELINE size: --T 32000)")
    a = ap.parse_args()
    if a.only in ("", "layers"):
        gen_layers(a.out)
    if a.only in ("", "tiny"):
        gen_tiny_step(a.out)
    if a.only in ("", "cfg4"):
        gen_cfg4_step(a.out, T=a.T, fname=a.fname)


if __name__ == "__main__":
    main()
