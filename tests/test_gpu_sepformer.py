"""GPU parity of the Sepformer (SURVEY.md §8 row a14 / cfg 4) product path against the oracle (oracle/sepformer_oracle.py) and the
reference-generated fixtures (tests/golden/sep_*.npz, cfg4_step.npz)."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import oracle.fqss_oracle as O
import oracle.sepformer_oracle as S
from tests.test_gpu_dptnet import QCFG, T, _leave_observer, close, dpt_fill, rnd

pytestmark = pytest.mark.gpu

TINY = dict(n_spks=2, kernel_size=16, stride=8, n_filters=16, n_repeats=1, n_heads=4, chunk_size=10)
TINY_FFN = 32
A = dict(gradient_based=True, act_quant=True)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    yield


@pytest.mark.parametrize("B,X,n,C", [(2, 5, 6, 16), (1, 34, 250, 256), (3, 7, 10, 64)])
def test_gnrows_kernels(B, X, n, C):
    """gLN on a row layout [n][B*X][C] vs F.group_norm on the reference's [B, C, n, X] tensor"""
    from fqss_amd import kernels as K
    x4 = (rnd(B, C, n, X, seed=41) * 1.3 + 0.4).requires_grad_(True)
    ga, be = (1 + 0.1 * rnd(C, seed=42)).requires_grad_(True), (0.1 * rnd(C, seed=43)).requires_grad_(True)
    y4 = F.group_norm(x4, 1, ga, be, 1e-8)
    g4 = rnd(B, C, n, X, seed=44)
    y4.backward(g4)
    to_rows = lambda t: t.detach().permute(2, 0, 3, 1).reshape(n, B * X, C).contiguous().cuda()
    y, ms = K.gnrows_fwd(to_rows(x4), ga.detach().cuda(), be.detach().cuda(), 1e-8, B * X, X, B)
    close(y, to_rows(y4).cpu(), 5e-6)
    gg, gb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gx = K.gnrows_bwd(to_rows(g4), to_rows(x4), ga.detach().cuda(), ms, gg, gb, B * X, X, B)
    close(gx, to_rows(x4.grad).cpu(), 2e-5)
    close(gg, ga.grad, 2e-5)
    close(gb, be.grad, 2e-5)


def _sd(g, name):
    return {k[len(name) + 4:]: T(g[k]) for k in g.files if k.startswith(name + ".sd.")}


def _idx_close(out, ref, lo, hi, max_frac):
    delta = (hi - lo) / 255.0
    a, b = np.rint((out - lo) / delta), np.rint((ref - lo) / delta)
    assert np.abs(a - b).max() <= 1
    assert float(np.mean(a != b)) <= max_frac, float(np.mean(a != b))
    return int((a != b).sum())


def test_sep_layer_pos_add(golden):
    """ConstQ on the positional table + the broadcasting AddQ, on sequence-first rows"""
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat.models.sepformerq import PositionalEncoding
    g = golden("sep_layers")
    name = "pos_add"
    sd = _sd(g, name)

    class PosAdd(nn.Module):
        def __init__(self, F_):
            super().__init__()
            self.pos = PositionalEncoding(F_)
            self.pos.const = QL.ConstQ(self.pos.const, **A)
            self.pos_add = QL.AddQ(QL.Add(), **A)

        def forward(self, x):
            return self.pos_add(x, self.pos(x))

    L = PosAdd(sd["pos.pe"].shape[-1])
    L.load_state_dict(sd, strict=True)
    L = L.cuda().train()
    _leave_observer(L)
    x = T(g[name + ".in0"]).permute(1, 0, 2).contiguous().cuda().requires_grad_(True)       # [L, B', F]
    y = L(x)
    y.backward(T(g[name + ".gout"]).permute(1, 0, 2).contiguous().cuda())
    lo, hi = float(sd["pos_add.activation_fake_quantize.min_range"]), float(sd["pos_add.activation_fake_quantize.max_range"])
    nflip = _idx_close(y.detach().permute(1, 0, 2).cpu().numpy(), g[name + ".out"], lo, hi, 3e-3)
    want = g[name + ".gin0"]
    bad = np.abs(x.grad.permute(1, 0, 2).cpu().numpy() - want) > 2e-4 * np.abs(want).max() + 2e-4 * np.abs(want)
    assert bad.mean() <= 2e-3 + 8.0 * nflip / want.size
    params = dict(L.named_parameters())
    for k in g.files:
        if k.startswith(name + ".grad."):
            w = g[k]
            np.testing.assert_allclose(params[k[len(name) + 6:]].grad.cpu().numpy(), w, rtol=3e-3, atol=(3e-3 + 0.05 * nflip) * (np.abs(w).max() + 1e-6), err_msg=k)


def test_sep_layer_groupnorm_4d_rows_and_reference_layout(golden):
    from fqss_amd.quantization.qat import qat_layers as QL
    g = golden("sep_layers")
    name = "groupnormq_4d"
    sd = _sd(g, name)
    x4 = T(g[name + ".in0"])
    B, C, Kc, S_ = x4.shape
    lo, hi = float(sd["activation_fake_quantize.min_range"]), float(sd["activation_fake_quantize.max_range"])
    for mode in ("rows", "nchw"):
        L = QL.GroupNormQ(nn.GroupNorm(1, C, eps=1e-8), **A)
        L.load_state_dict(sd, strict=True)
        L = L.cuda().train()
        _leave_observer(L)
        if mode == "rows":
            x = x4.permute(2, 0, 3, 1).reshape(Kc, B * S_, C).contiguous().cuda().requires_grad_(True)
            y = L.forward_rows(x, (B * S_, S_, B))
            y.backward(T(g[name + ".gout"]).permute(2, 0, 3, 1).reshape(Kc, B * S_, C).contiguous().cuda())
            out = y.detach().cpu().view(Kc, B, S_, C).permute(1, 3, 0, 2).numpy()
            gin = x.grad.cpu().view(Kc, B, S_, C).permute(1, 3, 0, 2).numpy()
        else:
            x = x4.cuda().requires_grad_(True)
            y = L(x)
            y.backward(T(g[name + ".gout"]).cuda())
            out, gin = y.detach().cpu().numpy(), x.grad.cpu().numpy()
        nflip = _idx_close(out, g[name + ".out"], lo, hi, 3e-3)
        want = g[name + ".gin0"]
        bad = np.abs(gin - want) > 2e-4 * np.abs(want).max() + 2e-4 * np.abs(want)
        assert bad.mean() <= 2e-3 + 8.0 * nflip / want.size, (mode, bad.mean())
        for k in g.files:
            if k.startswith(name + ".grad."):
                w = g[k]
                got = dict(L.named_parameters())[k[len(name) + 6:]].grad.cpu().numpy()
                np.testing.assert_allclose(got, w, rtol=3e-3, atol=(3e-3 + 0.05 * nflip) * (np.abs(w).max() + 1e-6), err_msg=mode + k)


def test_sep_layer_relu_and_trainable_residual_decoder(golden):
    from fqss_amd.quantization.qat import qat_layers as QL
    g = golden("sep_layers")
    for name in ("nlq_relu", "convtr1ddecoderq_trd"):
        sd = _sd(g, name)
        if name == "nlq_relu":
            L = QL.NlQ(nn.ReLU(), **A)
        else:
            ci = sd["convTr1d.weight"].shape[0]
            L = QL.ConvTr1dDecoderQ([nn.ConvTranspose1d(ci, 1, 16, stride=8, bias=False)], n_combiner=2, gradient_based=True,
                                    weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, out_quant=True, out_act_n_bits=8,
                                    train_res_dec=True)
        L.load_state_dict(sd, strict=True)
        L = L.cuda().train()
        _leave_observer(L)
        x = T(g[name + ".in0"]).cuda().requires_grad_(True)
        y = L(x)
        y.backward(T(g[name + ".gout"]).cuda())
        out, ref = y.detach().cpu().numpy(), g[name + ".out"]
        keys = ["activation_fake_quantize"] + (["activation_fake_quantize_residual"] if name != "nlq_relu" else [])
        nflip = 0
        for ch, key in enumerate(keys):
            o, r = (out[ch], ref[ch]) if len(keys) == 2 else (out, ref)
            nflip += _idx_close(o, r, float(sd[key + ".min_range"]), float(sd[key + ".max_range"]), 6e-3)
        want = g[name + ".gin0"]
        bad = np.abs(x.grad.cpu().numpy() - want) > 2e-4 * np.abs(want).max() + 2e-4 * np.abs(want)
        assert bad.mean() <= 2e-3 + 8.0 * nflip / want.size, (name, bad.mean())
        for k in g.files:
            if k.startswith(name + ".grad."):
                w = g[k]
                got = dict(L.named_parameters())[k[len(name) + 6:]].grad.cpu().numpy()
                np.testing.assert_allclose(got, w, rtol=3e-3, atol=(3e-3 + 0.05 * nflip) * (np.abs(w).max() + 1e-6), err_msg=k)


# ------------------------------------------------------------------------------------------------ model
def build_pair(seed=0, tiny=False, **kw):
    from fqss_amd.quantization.qat.models.load_model import quantize_model
    from fqss_amd.quantization.qat.models.sepformerq import MaskGenerator, SepformerQ
    torch.manual_seed(seed)
    model = SepformerQ(**kw)
    if tiny:      # same narrow feed-forward as the fixture generator (tools/make_goldens_sepformer.py)
        model.masker = MaskGenerator(kw["n_spks"], kw["n_filters"], n_repeats=kw["n_repeats"], n_heads=kw["n_heads"],
                                     chunk_size=kw["chunk_size"], n_ffn=TINY_FFN)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, dict(QCFG))
    return model.cuda().train(), fmodel.cuda().eval()


def _tiny_pair(g, prefix="sd0."):
    model, fmodel = build_pair(0, tiny=True, **TINY)
    model.load_state_dict({k[len(prefix):]: T(g[k]) for k in g.files if k.startswith(prefix)}, strict=True)
    fmodel.load_state_dict({k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}, strict=True)
    return model, fmodel


def test_sep_state_dict_layout(golden):
    g = golden("sep_tiny_step")
    model, _ = _tiny_pair(g)
    assert list(model.state_dict().keys()) == list(g["sd_keys"])


def test_sep_teacher_forward_matches_oracle():
    """float path of the full-size network (25.7 M parameters), B = 1, 1 s"""
    _, fmodel = build_pair(1, n_spks=2, kernel_size=16, stride=8)
    x, _ = O.synth_batch(1, 8000, seed=2)
    with torch.no_grad():
        y = fmodel(x.cuda()).cpu()
    ref = S.TeacherSepformer({k: v.cpu() for k, v in fmodel.state_dict().items()})(x)
    assert float((y - ref).norm() / ref.norm()) < 2e-4


def _check_step(g, p, r, model, loss_rel, est_tol, grad_tol):
    np.testing.assert_allclose(r["loss"].item(), g[p + "loss"], rtol=loss_rel, err_msg=p)
    np.testing.assert_allclose(r["kd"].item(), g[p + "kd"], rtol=loss_rel, err_msg=p)
    np.testing.assert_allclose(r["w"].cpu().numpy(), g[p + "w"], rtol=2.3e-4, err_msg=p)
    np.testing.assert_allclose(r["est"].cpu().numpy(), g[p + "est"], rtol=1e-4, atol=est_tol, err_msg=p)
    np.testing.assert_allclose(r["gnorm"].item(), g[p + "gnorm"], rtol=3e-4, err_msg=p)
    coef = min(1.0, 5.0 / (float(g[p + "gnorm"]) + 1e-6))
    n = 0
    for name, prm in model.named_parameters():
        k = p + "grad." + name
        if k in g.files:
            ref = g[k] / coef
            err = np.linalg.norm(prm.grad.cpu().numpy() - ref) / (np.linalg.norm(ref) + 1e-12)
            assert err <= grad_tol or np.linalg.norm(ref) < 1e-7, (k, err)
            n += 1
        else:
            assert float(prm.grad.abs().max()) == 0.0, name
    assert n >= 100


def test_sep_tiny_training_vs_reference_goldens(golden):
    from fqss_amd.runtime import KDTrainStep
    g = golden("sep_tiny_step")
    model, fmodel = _tiny_pair(g)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1.5e-4, clip=5.0)
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    for s in range(1, 54):
        r = step(x, tgt)
        if s == 1:
            _check_step(g, "s1.", r, model, 2e-5, 5e-6, 3e-3)
        elif f"s{s}.loss" in g.files:
            assert abs(r["loss"].item() - float(g[f"s{s}.loss"])) <= 1.5, (s, r["loss"].item(), float(g[f"s{s}.loss"]))


def _forced(g, s):
    from fqss_amd.quantization.qat import qat_quant as QQ
    model, fmodel = _tiny_pair(g, prefix=f"s{s}.post_sd.")
    for m in model.modules():
        if isinstance(m, QQ.GradientWeightFakeQuantize):
            m.observer_mode = False
        if isinstance(m, QQ.GradientActivationFakeQuantize):
            m.n_iter = min(s, m.max_observations)
    return model, fmodel


def test_sep_tiny_step2_from_reference_state(golden):
    from fqss_amd import kernels as K
    from fqss_amd.runtime import KDTrainStep
    g = golden("sep_tiny_step")
    model, fmodel = _forced(g, 1)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1.5e-4, clip=5.0)
    r = step._fwd_bwd(T(g["x"]).cuda(), T(g["tgt"]).cuda())
    step.arena.sumsq.zero_()
    K.sumsq(step.arena.flat_g, step.arena.sumsq)
    r["gnorm"] = step.arena.sumsq.sqrt().float()
    _check_step(g, "s2.", r, model, 5e-5, 2e-5, 4e-3)


def test_sep_tiny_step51_from_reference_state(golden):
    from fqss_amd import kernels as K
    g = golden("sep_tiny_step")
    model, fmodel = _forced(g, 50)
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    with torch.no_grad():
        est, fest = model(x), fmodel(x)
    out, w, sisdr, _ = K.kd_loss(est, fest, tgt, 0.1, want_grad=False)
    np.testing.assert_allclose(fest.cpu().numpy(), g["s51.fest"], rtol=1e-4, atol=5e-6)
    assert abs(out[0].item() - float(g["s51.loss"])) <= 0.2, (out[0].item(), float(g["s51.loss"]))


@pytest.mark.parametrize("fixture", ["cfg4_step", "cfg4_full_step"])
def test_sep_full_size_vs_reference_goldens(golden, fixture):
    """the FULL-SIZE SepformerQ (2 dual-path blocks x 2 x 8 transformer layers) vs digests of the real reference's run.  cfg4_step:
    1 x 2 s; cfg4_full_step: 1 x 4 s = the BASELINE workload of cfg 4 (T = 32000, `make_goldens_sepformer.py --only cfg4 --T 32000`)"""
    from fqss_amd.data import synth_batch
    from fqss_amd.runtime import KDTrainStep
    g = golden(fixture)
    B, T_ = int(g["B"]), int(g["T"])
    model, fmodel = build_pair(0, n_spks=2, kernel_size=16, stride=8)
    dpt_fill(fmodel, "T.")
    dpt_fill(model, "S.")
    assert [k for k, _ in model.named_parameters()] == list(g["param_names"])
    np.testing.assert_allclose([float(p.detach().double().sum()) for _, p in model.named_parameters()], g["param_sum"], rtol=1e-9, atol=1e-9)
    x, tgt = synth_batch(B, T_, seed=0, device="cuda")
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1.5e-4, clip=5.0)
    for s in range(1, 53):
        r = step(x, tgt)
        p = f"s{s}."
        if p + "loss" not in g.files:
            continue
        if s <= 2:
            f = 1.0 if s == 1 else 30.0
            for k in ("loss", "kd", "task"):
                np.testing.assert_allclose(r[k].item(), g[p + k], rtol=2e-5 * f, err_msg=p + k)
            np.testing.assert_allclose(r["gnorm"].item(), g[p + "gnorm"], rtol=(3e-4 if s == 1 else 1e-2), err_msg=p)
            if s == 1:
                ref = g[p + "est"]
                np.testing.assert_allclose(r["est"].cpu().numpy(), ref, rtol=1e-4, atol=2e-4 * float(np.abs(ref).max()))
                coef = min(1.0, 5.0 / (float(g[p + "gnorm"]) + 1e-6))
                bad = []
                for (name, prm), ref_n in zip(model.named_parameters(), g[p + "grad_norm"]):
                    got = float(prm.grad.double().norm())
                    if ref_n < 0:
                        assert got == 0.0, name
                    elif abs(got - ref_n / coef) > (1e-2 if prm.numel() == 1 else 4e-3) * (ref_n / coef) + 1e-6:
                        bad.append((name, got, ref_n / coef))
                assert not bad, bad[:5]
        else:
            # statistical only: from random init at lr 1.5e-4 the REFERENCE's own loss moves by 1.5-3 dB between consecutive steps
            # here (its log: 4.8 dB at step 40, 8.1 at 50, 6.6 at 51, 7.9 at 52) and the fp32 atomics of the weight gradients
            # make every run of this build land somewhere else in that band (observed 4.1 .. 9.6 dB at step 51 over a dozen runs, and
            # one run at 15.0): the bound only says "the same regime as the reference", the tight gates are the step-1/2 digests above
            assert abs(r["loss"].item() - float(g[p + "loss"])) <= 10.0, (s, r["loss"].item(), float(g[p + "loss"]))
            assert r["loss"].item() < float(g["s1.loss"]) - 8.0          # ... while the run as a whole trains (25.5 dB at step 1)


def test_sepformer_backward_segments_match_the_single_pass_backward():
    """gradient buckets for cfg 4 (SepformerQ.fqss_segments): the full-size network's backward as 4 segments -- the intra and inter
    transformer stacks of its two dual-path blocks, the encoder -> mask-multiply edge as a late cut -- against the one-pass backward"""
    from tests.helpers_segments import check_backward_segments
    x, tgt = O.synth_batch(1, 8000, seed=4)
    check_backward_segments(lambda: build_pair(2, n_spks=2, kernel_size=16, stride=8), x.cuda(), tgt.cuda(), nb=4, nseg=4)


def test_sepformer_batched_quantizer_tables_cover_every_weight():
    """QuantTables on the tiny Sepformer: all 72 weight quantizers (convolutions, LinearQ / LinearNlQ, both attention projections, the 1x1
    Conv2dQ) run from the tables -- output bit-identical to the per-layer quantizers, every gradient equal"""
    from tests.helpers_segments import check_batched_tables
    x, tgt = O.synth_batch(1, 4000, seed=3)
    check_batched_tables(lambda: build_pair(0, tiny=True, **TINY), x.cuda(), tgt.cuda(), 72, step_kw=dict(kd_lambda=0.1, clip=0.0))


def test_attention_forms_agree_on_a_quantizing_layer(monkeypatch):
    """B2-style gate for round 3's attention paths at LAYER level (a whole random-init network turns one flipped 8-bit bin into a 6 %
    difference of its output -- SURVEY A.4 -- whichever two forms are compared): one MultiheadAttentionQ (64 features, 4 heads: head_dim
    16; 250 x 40 rows) in its quantizing phase, forward + backward, with the attention core (a) on the u8 CODES of q / k / v
    (fqss_attn_long_fwd_c / _bwd_c behind fqss_mha_prep_fwd_c), (b) on the de-quantized values through the split-bf16 streaming
    kernels, (c) through the LDS-resident kernels of csrc/attn.hip: same output, input gradient and parameter gradients up to fp32 noise
    and the few bins it flips in the layer's own output quantizers."""
    from fqss_amd import kernels as K
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat import qat_quant as QQ
    L, B, E, nh = 250, 40, 64, 4
    x0, g0 = rnd(L, B, E, seed=1).cuda(), rnd(L, B, E, seed=2).cuda()
    res = {}
    for form, (coded, stream) in (("coded", (True, True)), ("split-bf16", (False, True)), ("lds-resident", (False, False))):
        monkeypatch.setattr(K, "ATTN_CODED", coded)
        monkeypatch.setattr(K, "ATTN_STREAM", stream)
        torch.manual_seed(11)
        layer = QL.MultiheadAttentionQ(nn.MultiheadAttention(E, nh)).cuda().train()
        with torch.no_grad():
            layer(x0, x0, x0)                                     # observer call: ranges from the data
        for m in layer.modules():
            if isinstance(m, QQ.GradientActivationFakeQuantize):
                m.n_iter = m.max_observations
        x = x0.clone().requires_grad_(True)
        y = layer(x, x, x)[0]
        y.backward(g0)
        res[form] = (y.detach().clone(), x.grad.clone(), torch.cat([p.grad.reshape(-1) for p in layer.parameters() if p.grad is not None]))
        monkeypatch.undo()
    ref = res["lds-resident"]
    for form in ("coded", "split-bf16"):
        e = [float((a - b).norm() / b.norm()) for a, b in zip(res[form], ref)]
        print(form, "output / input gradient / parameter gradients:", e)
        assert e[0] <= 2e-3 and e[1] <= 5e-3 and e[2] <= 5e-3, (form, e)


@pytest.mark.parametrize("coded_input", [True, False])
def test_relu_behind_a_linear_rides_in_its_quantizer(coded_input, monkeypatch):
    """B2-style gate: LinearQ -> nn.ReLU (the feed-forward pair of the Sepformer layer, sepformerq.py:63) with the ReLU folded into the
    output quantizer's pass (act = ACT_POST_RELU: in the int8 GEMM's epilogue when the input carries codes, else in the quantizer
    pass; the backward in fqss_actq_bwd_colbias) against linear + quantizer followed by a ReLU pass: output bit-identical, gradients
    of the input, the weight, the bias and the ranges within fp32 summation noise"""
    from fqss_amd.quantization.qat import qat_layers as QL
    R, Ci, Co = 4000, 64, 256
    x0, g0 = rnd(R, 1, Ci, seed=1).cuda(), rnd(R, 1, Co, seed=2).cuda()
    res = {}
    for kind in ("fused", "unfused"):
        monkeypatch.setattr(QL, "FUSE_POSTRELU", kind == "fused")
        torch.manual_seed(5)
        ln = QL.LayerNormQ(nn.LayerNorm(Ci), gradient_based=True, act_quant=True).cuda()
        lin = QL.LinearQ(nn.Linear(Ci, Co), gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8).cuda()
        relu = nn.ReLU()

        def fwd(x):
            h = ln(x) if coded_input else x
            if kind == "fused":
                return lin(h, post_relu=True)
            return QL.fq_node(None, lin(h), relu)
        with torch.no_grad():
            for _ in range(50):
                fwd(x0)
        x = x0.clone().requires_grad_(True)
        y = fwd(x)
        y.backward(g0)
        aq = lin.activation_fake_quantize
        res[kind] = (y.detach(), x.grad, lin.linear.weight.grad, lin.linear.bias.grad, aq.min_range.grad, aq.max_range.grad)
        monkeypatch.undo()
    f, u = res["fused"], res["unfused"]
    assert torch.equal(f[0], u[0]) and float((f[0] == 0).float().mean()) > 0.2
    for i, name in enumerate(("dx", "dW", "db", "min", "max"), start=1):
        assert f[i] is not None and u[i] is not None, name
        err = float((f[i] - u[i]).norm() / (u[i].norm() + 1e-12))
        assert err <= 1e-4, (name, err)


def test_linear_then_relu_quantizer_in_one_launch(monkeypatch):
    """B2-style gate: LinearQ -> NlQ(ReLU) -> LinearQ (the feed-forward block of the QUANTIZED Sepformer layer, sepformerq.py:64: the
    reference's quantize_model wraps the ReLU in its own quantizer) with both quantizers in the int8 GEMM's epilogue
    (fqss_qrow_fwdq2) and one backward pass (fqss_actq2_bwd_colbias) against the two modules as they are: output and the codes handed
    to the next linear bit-identical, every gradient (input, weights, biases, all three quantizers' ranges) within fp32 summation noise"""
    from fqss_amd.quantization.qat import qat_layers as QL
    R, Ci, Ch = 4000, 64, 256
    x0, g0 = rnd(R, 1, Ci, seed=1).cuda(), rnd(R, 1, Ci, seed=2).cuda()
    res = {}
    for kind in ("fused", "unfused"):
        monkeypatch.setattr(QL, "FUSE_NLQ2", kind == "fused")
        torch.manual_seed(5)
        ln = QL.LayerNormQ(nn.LayerNorm(Ci), gradient_based=True, act_quant=True).cuda()
        l1 = QL.LinearQ(nn.Linear(Ci, Ch), gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8).cuda()
        nl = QL.NlQ(nn.ReLU(), gradient_based=True, act_quant=True).cuda()
        l2 = QL.LinearQ(nn.Linear(Ch, Ci), gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8).cuda()
        seen = {}

        def fwd(x):
            h = QL.linear_then_relu_q(l1, nl, ln(x))
            seen["h"] = h
            return l2(h)
        with torch.no_grad():
            for _ in range(50):
                fwd(x0)
        x = x0.clone().requires_grad_(True)
        y = fwd(x)
        h = seen["h"]
        codes = getattr(h, "_fqss_rowq", None) or getattr(h, "_fqss_q", None)
        assert codes is not None, "the feed-forward's second linear must find the codes of its input"
        y.backward(g0)
        grads = [x.grad, l1.linear.weight.grad, l1.linear.bias.grad, l2.linear.weight.grad, l2.linear.bias.grad]
        for m in (l1, nl, l2):
            grads += [m.activation_fake_quantize.min_range.grad, m.activation_fake_quantize.max_range.grad]
        res[kind] = (y.detach(), h.detach().clone(), codes.idx.clone().view(-1), grads)
        monkeypatch.undo()
    f, u = res["fused"], res["unfused"]
    assert torch.equal(f[0], u[0]) and torch.equal(f[1], u[1]) and torch.equal(f[2], u[2])
    assert float((f[1] == f[1].min()).float().mean()) > 0.2          # the ReLU bites
    names = ("dx", "dW1", "db1", "dW2", "db2", "min1", "max1", "min_nl", "max_nl", "min2", "max2")
    for name, a, b in zip(names, f[3], u[3]):
        assert a is not None and b is not None, name
        err = float((a - b).norm() / (b.norm() + 1e-12))
        assert err <= 3e-4, (name, err)
