#!/usr/bin/env python3
"""Golden vectors for the first LayerQ classes of SURVEY.md §8 row a15 (cfg 5: HTDemucs), generated from the REAL reference
(layer level only: the model is not built yet).  hd_layers.npz: LinearNlQ with GELU and ReLU (qat_layers.py:539-561), NlQ(GELU),
Conv1dNlQ 1x1 + GLU(dim=1) (the `rewrite` convs, hdemucsq.py:127, 314), DivQ, EmbeddingQ (qat_layers.py:490-508), Conv1dQ with a
dilated k3 kernel, Conv1dNlQ k8 s4 p2 + GELU (time-branch encoder), Conv1dGnNlQ with GELU and with GLU (DConv, demucsq.py:163-169).
Usage: python tools/make_goldens_htdemucs_layers.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402
import make_goldens_dptnet as MD  # noqa: E402

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from quantization.qat import qat_layers as RL  # noqa: E402

keyed_randn, fill, run_layer, P = MG.keyed_randn, MD.fill, MD.run_layer, MD.P


class Cross(nn.Module):
    """MultiheadAttentionQ(query, key, key)[0] for the generic layer runner"""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, q, k):
        return self.m(q, k, k)[0]


def main():
    out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
    d = {}
    x = keyed_randn("hd.x", (9, 5, 16), 0.9)
    L = RL.LinearNlQ(nn.Linear(16, 24), nn.GELU(), **P); fill(L, "lng.")
    run_layer("linearnlq_gelu", L, [x], d)
    L = RL.LinearNlQ(nn.Linear(16, 24), nn.ReLU(), **P); fill(L, "lnr.")
    run_layer("linearnlq_relu", L, [x], d)
    L = RL.NlQ(nn.GELU(), gradient_based=True, act_quant=True)
    run_layer("nlq_gelu", L, [keyed_randn("hd.xg", (3, 12, 37), 1.2)], d)
    L = RL.Conv1dNlQ(nn.Conv1d(12, 24, 1), nn.GLU(dim=1), **P); fill(L, "glu.")
    run_layer("conv1dnlq_glu", L, [keyed_randn("hd.xc", (3, 12, 37), 0.9)], d)
    L = RL.DivQ(RL.Div(), gradient_based=True, act_quant=True)
    run_layer("divq", L, [keyed_randn("hd.num", (3, 12, 37), 0.8), keyed_randn("hd.den", (3, 12, 37), 0.3).abs() + 0.5], d)
    # general Conv1d geometries of the HTDemucs layers and DConv's conv + GroupNorm + non-linearity blocks
    xt = keyed_randn("hd.xt", (2, 6, 53), 0.9)
    L = RL.Conv1dQ(nn.Conv1d(6, 10, 3, dilation=2, padding=2), **P); fill(L, "ck3.")
    run_layer("conv1dq_k3_d2", L, [xt], d)
    L = RL.Conv1dNlQ(nn.Conv1d(6, 10, 8, stride=4, padding=2), nn.GELU(), **P); fill(L, "ck8.")
    run_layer("conv1dnlq_k8_s4_gelu", L, [xt], d)
    L = RL.Conv1dGnNlQ(nn.Conv1d(6, 12, 3, dilation=1, padding=1), nn.GroupNorm(1, 12), nn.GELU(), **P); fill(L, "cgn1.")
    run_layer("conv1dgnnlq_gelu", L, [xt], d)
    L = RL.Conv1dGnNlQ(nn.Conv1d(6, 12, 1), nn.GroupNorm(1, 12), nn.GLU(1), **P); fill(L, "cgn2.")
    run_layer("conv1dgnnlq_glu", L, [xt], d)
    # 2-D convolutions along frequency, the 3x3 decoder `rewrite`, and the transposed convolutions (hdemucsq.py:96-127, 303-347)
    xf = keyed_randn("hd.xf", (2, 4, 22, 7), 0.9)
    L = RL.Conv2dNlQ(nn.Conv2d(4, 6, (8, 1), (4, 1), (2, 0)), nn.GELU(), **P); fill(L, "c2a.")
    run_layer("conv2dnlq_k8_s4_gelu", L, [xf], d)
    L = RL.Conv2dNlQ(nn.Conv2d(4, 8, 3, 1, 1), nn.GLU(1), **P); fill(L, "c2b.")
    run_layer("conv2dnlq_3x3_glu", L, [xf], d)
    L = RL.Conv2dNlQ(nn.Conv2d(4, 8, 1), nn.GLU(1), **P); fill(L, "c2c.")
    run_layer("conv2dnlq_1x1_glu", L, [xf], d)
    L = RL.ConvTranspose2dNlQ(nn.ConvTranspose2d(4, 6, (8, 1), (4, 1)), nn.GELU(), **P); fill(L, "ct2.")
    run_layer("convtr2dnlq_k8_s4_gelu", L, [keyed_randn("hd.xg2", (2, 4, 5, 7), 0.9)], d)
    L = RL.ConvTranspose1dNlQ(nn.ConvTranspose1d(6, 4, 8, 4), nn.GELU(), **P); fill(L, "ct1.")
    run_layer("convtr1dnlq_k8_s4_gelu", L, [keyed_randn("hd.xg1", (2, 6, 13), 0.9)], d)
    L = RL.ConvTranspose1dQ(nn.ConvTranspose1d(6, 4, 5, 3, padding=1, output_padding=2), **P); fill(L, "ct0.")
    run_layer("convtr1dq_k5_s3_p1", L, [keyed_randn("hd.xg1", (2, 6, 13), 0.9)], d)
    # 8-bit I/O blocks of HTDemucs: widened first convs of both branches, last transposed convs with the residual (LSB) output
    torch.manual_seed(11)
    L = RL.Conv1dEncoderQ(nn.Sequential(nn.Conv1d(2, 6, 8, 4, 2), nn.GELU()), n_splitter=2, **P); fill(L, "e1.")
    run_layer("conv1dencoderq_k8_s4_gelu", L, [keyed_randn("hd.xe1", (2, 4, 53), 0.9)], d)
    L = RL.Conv2dEncoderQ(nn.Sequential(nn.Conv2d(4, 6, (8, 1), (4, 1), (2, 0)), nn.GELU()), n_splitter=2, **P); fill(L, "e2.")
    run_layer("conv2dencoderq_k8_s4_gelu", L, [keyed_randn("hd.xe2", (2, 8, 22, 7), 0.9)], d)
    L = RL.ConvTr1dDecoderQ(nn.Sequential(nn.ConvTranspose1d(6, 4, 8, 4)), n_combiner=2, **P); fill(L, "d1.")
    run_layer("convtr1ddecoderq_stereo", L, [keyed_randn("hd.xd1", (2, 6, 13), 0.9)], d)
    L = RL.ConvTr2dDecoderQ(nn.Sequential(nn.ConvTranspose2d(6, 4, (8, 1), (4, 1))), n_combiner=2, train_res_dec=True, **P); fill(L, "d2.")
    run_layer("convtr2ddecoderq_resdec", L, [keyed_randn("hd.xd2", (2, 6, 5, 7), 0.9)], d)
    # attention of the HTDemucs transformer: batch-first rows, self-attention on a long-ish sequence and cross attention (Lq != Lk)
    L = MD.First(RL.MultiheadAttentionQ(nn.MultiheadAttention(16, 4, dropout=0.0, batch_first=True), **P), 3); fill(L, "mhs.")
    run_layer("mhaq_bf_self", L, [keyed_randn("hd.xs", (2, 45, 16), 0.9)], d)
    L = Cross(RL.MultiheadAttentionQ(nn.MultiheadAttention(16, 4, dropout=0.0, batch_first=True), **P)); fill(L, "mhc.")
    run_layer("mhaq_bf_cross", L, [keyed_randn("hd.xq", (2, 37, 16), 0.9), keyed_randn("hd.xk", (2, 21, 16), 0.9)], d)
    # EmbeddingQ: integer input, so run it by hand with the same protocol as run_layer
    emb = RL.EmbeddingQ(nn.Embedding(11, 16), **P); fill(emb, "emb.")
    idx = torch.tensor([[0, 3, 10, 3], [7, 1, 1, 5]])
    emb.train()
    from quantization.qat.models.load_model import enable_observer
    from quantization.qat import qat_quant as RQ
    enable_observer(emb, True)
    with torch.no_grad():
        for _ in range(50):
            y_obs = emb(idx)
    d["embeddingq.out_obs"] = MG.npy(y_obs)
    with torch.no_grad():
        for m in emb.modules():
            if isinstance(m, RQ.GradientActivationFakeQuantize):
                m.min_range.mul_(0.93); m.max_range.mul_(0.93)
    y = emb(idx)
    g = keyed_randn("embeddingq.gout", tuple(y.shape))
    y.backward(g)
    d["embeddingq.idx"], d["embeddingq.out"], d["embeddingq.gout"] = idx.numpy(), MG.npy(y), MG.npy(g)
    for k, v in emb.state_dict().items():
        d["embeddingq.sd." + k] = MG.npy(v)
    for k, p in emb.named_parameters():
        if p.grad is not None:
            d["embeddingq.grad." + k] = MG.npy(p.grad)
    np.savez_compressed(os.path.join(out, "hd_layers.npz"), **d)
    print("hd_layers:", len(d), "arrays")


if __name__ == "__main__":
    main()
