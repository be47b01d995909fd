"""Pins oracle/dptnet_oracle.py against golden vectors produced by the REAL reference
(tools/make_goldens_dptnet.py).  CPU only."""
import numpy as np
import pytest
import torch

import oracle.dptnet_oracle as D
import oracle.fqss_oracle as O

torch.set_num_threads(1)

TINY = dict(n_src=2, kernel_size=2, segment_size=10)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def table(g, name):
    pre = name + ".sd."
    return {k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)}


def strip_m(sd):
    """fixtures of wrapped layers (LSTMQ / MHAQ sit in a `First` wrapper) carry an 'm.' prefix"""
    return {(k[2:] if k.startswith("m.") else k): v for k, v in sd.items()}


def run(g, name, fn, tol=2e-6, gtol=2e-5, wrapped=False):
    sd = table(g, name)
    if wrapped:
        sd = strip_m(sd)
    tab = D.DQTable({"L." + k: v for k, v in sd.items()})
    tab.leave_observer_phase()
    ins = [T(g[f"{name}.in{i}"]).requires_grad_(True) for i in range(8) if f"{name}.in{i}" in g.files]
    y = fn(tab, *ins)
    want = T(g[name + ".out"])
    err = (y - want).abs().max().item()
    scale = want.abs().max().item()
    # a pre-quant value within rounding noise of a bin edge may land in the neighbouring bin: allow a few one-step flips
    step = (scale / 100)
    flips = ((y - want).abs() > tol * max(scale, 1)).float().mean().item()
    assert flips < 2e-3 and err < step, (name, err, flips)
    y.backward(T(g[name + ".gout"]))
    for i, t in enumerate(ins):
        if f"{name}.gin{i}" in g.files:
            w = T(g[f"{name}.gin{i}"])
            assert (t.grad - w).abs().max().item() <= gtol * max(1.0, w.abs().max().item()) + 1e-2 * flips * w.abs().max().item() * 100, (name, "gin", i)
    for k, v in tab.p.items():
        gk = f"{name}.grad." + ("m." if wrapped else "") + k[2:]
        if gk in g.files and not k.endswith(".decoder_bias"):          # an alias of the decoder's bias in the reference
            w = T(g[gk])
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            denom = max(w.abs().max().item(), 1e-3)
            assert (got - w).abs().max().item() / denom < (5e-3 if flips > 0 else 2e-4), (name, k)
    return tab


def test_dpt_layer_fixtures(golden):
    g = golden("dpt_layers")
    run(g, "layernormq", lambda t, x: t.layer_norm_q("L", x))
    run(g, "linearq", lambda t, x: t.linear_q("L", x))
    run(g, "lstmq", lambda t, x: t.lstm_q("L", x), wrapped=True)
    run(g, "mhaq", lambda t, x: t.mha_q("L", x, 4), wrapped=True)
    run(g, "conv2dq", lambda t, x: t.conv2d_q("L", x))
    run(g, "conv1dnlq_tanh", lambda t, x: t.conv1d_nl_q("L", x, "tanh"))
    run(g, "conv1dnlq_sigmoid", lambda t, x: t.conv1d_nl_q("L", x, "sigmoid"))
    run(g, "mulq_same", lambda t, a, b: t._A("L", a * b))
    run(g, "mulq_mask", lambda t, a, b: t._A("L", a * b))
    run(g, "addq_seq", lambda t, a, b: t._A("L", a + b))
    run(g, "nlq_prelu4", lambda t, x: t._nl("L", x))
    run(g, "conv1dencoderq_k2", lambda t, x: t._conv("L", x, nl="relu"))
    run(g, "groupnormq_enc", lambda t, x: t._gn("L", x))
    run(g, "lineardecoderq", lambda t, x: t.linear_decoder_q("L", x, 2))


def test_htdemucs_first_layer_fixtures(golden):
    """row a15, layer level: LinearNlQ (GELU / ReLU), NlQ(GELU), 1x1 Conv1dNlQ + GLU, DivQ, EmbeddingQ vs the reference"""
    g = golden("hd_layers")
    run(g, "linearnlq_gelu", lambda t, x: t.linear_nl_q("L", x, "gelu"))
    run(g, "linearnlq_relu", lambda t, x: t.linear_nl_q("L", x, "relu"))
    run(g, "nlq_gelu", lambda t, x: t._A("L", torch.nn.functional.gelu(x)))
    run(g, "conv1dnlq_glu", lambda t, x: t.conv1d_nl_q("L", x, "glu"))
    run(g, "divq", lambda t, a, b: t.div_q("L", a, b))
    run(g, "conv1dq_k3_d2", lambda t, x: t.conv1d_nl_q("L", x, None, dilation=2, padding=2))
    run(g, "conv1dnlq_k8_s4_gelu", lambda t, x: t.conv1d_nl_q("L", x, "gelu", stride=4, padding=2))
    run(g, "conv1dgnnlq_gelu", lambda t, x: t.conv1d_gn_nl_q("L", x, "gelu", padding=1))
    run(g, "conv1dgnnlq_glu", lambda t, x: t.conv1d_gn_nl_q("L", x, "glu"))
    run(g, "conv2dnlq_k8_s4_gelu", lambda t, x: t.conv2d_nl_q("L", x, "gelu", stride=(4, 1), padding=(2, 0)))
    run(g, "conv2dnlq_3x3_glu", lambda t, x: t.conv2d_nl_q("L", x, "glu", padding=1))
    run(g, "conv2dnlq_1x1_glu", lambda t, x: t.conv2d_nl_q("L", x, "glu"))
    run(g, "convtr2dnlq_k8_s4_gelu", lambda t, x: t.convtr_nl_q("L", x, "gelu", stride=(4, 1)))
    run(g, "convtr1dnlq_k8_s4_gelu", lambda t, x: t.convtr_nl_q("L", x, "gelu", stride=4))
    run(g, "convtr1dq_k5_s3_p1", lambda t, x: t.convtr_nl_q("L", x, None, stride=3, padding=1, output_padding=2))
    run(g, "mhaq_bf_self", lambda t, x: t.mha_q("L", x, 4, batch_first=True), wrapped=True)
    run(g, "mhaq_bf_cross", lambda t, x, k: t.mha_q("L", x, 4, key=k, batch_first=True), wrapped=True)
    run(g, "conv1dencoderq_k8_s4_gelu", lambda t, x: t.conv1d_nl_q("L", x, "gelu", stride=4, padding=2))
    run(g, "conv2dencoderq_k8_s4_gelu", lambda t, x: t.conv2d_nl_q("L", x, "gelu", stride=(4, 1), padding=(2, 0)))
    run(g, "convtr1ddecoderq_stereo", lambda t, x: t.convtr_decoder_q("L", x, 2, stride=4))
    run(g, "convtr2ddecoderq_resdec", lambda t, x: t.convtr_decoder_q("L", x, 2, True, stride=(4, 1)))
    tab = D.DQTable({"L." + k: v for k, v in table(g, "embeddingq").items()})
    tab.leave_observer_phase()
    y = tab.embedding_q("L", T(g["embeddingq.idx"]))
    y.backward(T(g["embeddingq.gout"]))
    assert (y.detach() - T(g["embeddingq.out"])).abs().max().item() < 1e-6
    for k, v in tab.p.items():
        if "embeddingq.grad." + k[2:] in g.files:
            np.testing.assert_allclose(v.grad.numpy(), g["embeddingq.grad." + k[2:]], rtol=1e-5, atol=1e-6, err_msg=k)


def test_dpt_observer_ranges(golden):
    """50 observer calls on one input: every range of the layer (incl. the MHA quantizers whose outputs are discarded)"""
    g = golden("dpt_layers")
    for name, fn, wrapped in (("mhaq", lambda t, x: t.mha_q("L", x, 4), True), ("lstmq", lambda t, x: t.lstm_q("L", x), True),
                              ("lineardecoderq", lambda t, x: t.linear_decoder_q("L", x, 2), False)):
        pre = name + ".sd_obs."
        sd = {k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)}
        sd = strip_m(sd) if wrapped else sd
        init = {}
        for k, v in sd.items():
            if k.endswith("min_range"):
                init["L." + k] = torch.full_like(v, -0.5)
            elif k.endswith("max_range"):
                init["L." + k] = torch.full_like(v, 0.5)
            else:
                init["L." + k] = v
        tab = D.DQTable(init)
        x = T(g[name + ".in0"])
        with torch.no_grad():
            for _ in range(50):
                y = fn(tab, x)
        np.testing.assert_allclose(y.numpy(), g[name + ".out_obs"], rtol=1e-5, atol=1e-6)
        for k, v in sd.items():
            if k.endswith("_range"):
                np.testing.assert_allclose(tab.p["L." + k].detach().numpy(), v.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


def test_dpt_data_movement(golden):
    g = golden("dpt_layers")
    sig = T(g["ola.in"])
    assert torch.equal(D.overlap_and_add(sig, 2), T(g["ola.out_step2"]))
    assert torch.equal(D.overlap_and_add(sig[..., :2], 1), T(g["ola.out_step1"]))
    for Tn in (37, 40, 45):
        seg, rest = D.split_feature(T(g[f"seg{Tn}.in"]), 10)
        assert rest == int(g[f"seg{Tn}.rest"]) and torch.equal(seg, T(g[f"seg{Tn}.out"]))
        a, b = D.merge_halves(seg)
        m = a + b
        m = m[:, :, :-rest] if rest > 0 else m
        assert torch.equal(m, T(g[f"seg{Tn}.merged"]))


def _tiny(g):
    sd = {k[4:]: T(g[k]) for k in g.files if k.startswith("sd0.")}
    fsd = {k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}
    return D.StudentDPTNetQ(sd, **TINY), D.TeacherDPTNet(fsd, **TINY)


def _cmp_step(g, p, r, s, est_tol, loss_rel, grad_tol):
    np.testing.assert_allclose(r["fest"].numpy(), g[p + "fest"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["est"].detach().numpy(), g[p + "est"], rtol=0, atol=est_tol)
    assert abs(float(r["loss"].detach()) - float(g[p + "loss"])) < loss_rel * abs(float(g[p + "loss"]))
    assert abs(float(r["kd"].detach()) - float(g[p + "kd"])) < loss_rel * abs(float(g[p + "kd"]))
    nograd = set(g[p + "nograd"].tolist())
    for k, v in s.p.items():
        if p + "grad." + k in g.files:
            w = g[p + "grad." + k]
            assert np.abs(v.grad.numpy() - w).max() <= grad_tol * max(np.abs(w).max(), 1e-4), k
        else:
            assert k in nograd and (v.grad is None or float(v.grad.abs().max()) == 0.0), k


def test_dpt_tiny_training_matches_reference(golden):
    """53 free-running QAT steps of the tiny DPTNetQ: step 1 (float network, first observer call) to rounding; the run as a
    whole statistically (a weight bin flipped by Adam's sign-like first update moves the tiny net by percents)"""
    g = golden("dpt_tiny_step")
    s, t = _tiny(g)
    tr = O.Trainer(s, t, lr=4e-4)
    x, tgt = T(g["x"]), T(g["tgt"])
    for step in range(1, 54):
        r = tr.step(x, tgt)
        if step == 1:
            assert abs(float(r["gnorm"]) - float(g["s1.gnorm"])) < 1e-4 * float(g["s1.gnorm"])
            _cmp_step(g, "s1.", r, s, 3e-6, 2e-5, 2e-4)
        if step in (2, 50, 51, 53):
            assert abs(float(r["loss"].detach()) - float(g[f"s{step}.loss"])) < 0.5, (step, float(r["loss"]), float(g[f"s{step}.loss"]))
    for k in g.files:      # the observer EMA saw nearly the same activations for 50 steps
        if k.startswith("s50.post_sd.") and k.endswith("_range") and g[k].size == 1:
            got, want = float(s.p[k[len("s50.post_sd."):]].detach().reshape(-1)[0]), float(g[k].reshape(-1)[0])
            assert abs(got - want) < 0.05 * max(abs(want), 0.05), (k, got, want)


def _forced(g, step):
    sd = {k[len(f"s{step}.post_sd."):]: T(g[k]) for k in g.files if k.startswith(f"s{step}.post_sd.")}
    fsd = {k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}
    return D.StudentDPTNetQ(sd, **TINY), D.TeacherDPTNet(fsd, **TINY)


def test_dpt_tiny_teacher_forced_step2(golden):
    """step 2 from the reference's own state after step 1: the first forward with fake-quantized weights (all 13 weight
    quantizer kinds of the model), activations still observed"""
    g = golden("dpt_tiny_step")
    s, t = _forced(g, 1)
    for q in s.wq.values():
        q.observer = False
    for q in s.aq.values():
        q.n_iter = 1
    r = O.kd_step(s, t, T(g["x"]), T(g["tgt"]))
    r["loss"].backward()
    gn = torch.nn.utils.clip_grad_norm_(s.parameters(), 5.0)
    assert abs(float(gn) - float(g["s2.gnorm"])) < 2e-4 * float(g["s2.gnorm"])
    _cmp_step(g, "s2.", r, s, 5e-6, 3e-5, 5e-4)


def test_dpt_tiny_teacher_forced_step51(golden):
    """first fully quantizing step from the reference's own state after 50 steps"""
    g = golden("dpt_tiny_step")
    s, t = _forced(g, 50)
    s.leave_observer_phase()
    r = O.kd_step(s, t, T(g["x"]), T(g["tgt"]))
    r["loss"].backward()
    assert abs(float(r["loss"].detach()) - float(g["s51.loss"])) < 0.02
    want, got = g["s51.est"], r["est"].detach().numpy()
    assert np.abs(got - want).max() < 0.03 * np.abs(want).max()
