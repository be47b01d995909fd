"""GPU parity of the product path (modules -> ops -> C ABI -> HIP) against reference-generated
goldens and the oracle: per-LayerQ (gate G1, teacher-forced), tiny-model training (G2 observer phase
within 1e-5 / 1e-3 dB, G3 quantizing phase statistically), and full-size (cfg-2) properties."""
import copy

import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle.fqss_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    yield


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _mods():
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat import qat_quant as QQ
    return QL, QQ


P = dict(gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8)
A = dict(gradient_based=True, act_quant=True)


def _build_layer(name, g):
    QL, _ = _mods()
    sd = {k[len(name) + 4:]: T(g[k]) for k in g.files if k.startswith(name + ".sd.")}
    shp = lambda k: tuple(sd[k].shape)
    if name == "conv1dq_pw":
        co, ci, _ = shp("conv1d.weight"); L = QL.Conv1dQ(nn.Conv1d(ci, co, 1), **P)
    elif name in ("conv1dnlq_pw_prelu", "conv1dnlq_pw_relu"):
        co, ci, _ = shp("conv1d.weight")
        L = QL.Conv1dNlQ(nn.Conv1d(ci, co, 1), nn.PReLU() if name.endswith("prelu") else nn.ReLU(), **P)
    elif name.startswith("conv1dnlq_dw_d"):
        d = int(name.rsplit("d", 1)[1]); c = shp("conv1d.weight")[0]
        L = QL.Conv1dNlQ(nn.Conv1d(c, c, 3, padding=d, dilation=d, groups=c), nn.PReLU(), **P)
    elif name == "groupnormq":
        L = QL.GroupNormQ(nn.GroupNorm(1, shp("groupnorm.weight")[0], eps=1e-8), **A)
    elif name == "addq":
        L = QL.AddQ(QL.Add(), **A)
    elif name == "mulq":
        L = QL.MulQ(QL.Mul(), **A)
    elif name == "nlq_prelu":
        L = QL.NlQ(nn.PReLU(), **A)
    elif name == "conv1dencoderq":
        co = shp("conv1d.weight")[0]
        L = QL.Conv1dEncoderQ([nn.Conv1d(1, co, 16, stride=8, bias=False)], n_splitter=2, **P)
    elif name == "convtr1ddecoderq":
        ci = shp("convTr1d.weight")[0]
        L = QL.ConvTr1dDecoderQ([nn.ConvTranspose1d(ci, 1, 16, stride=8, bias=False)], n_combiner=2, gradient_based=True,
                                weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, out_quant=True, out_act_n_bits=8)
    else:
        raise KeyError(name)
    L.load_state_dict(sd, strict=True)
    return L.cuda().train()


LAYER_NAMES = ["conv1dq_pw", "conv1dnlq_pw_prelu", "conv1dnlq_pw_relu", "conv1dnlq_dw_d1", "conv1dnlq_dw_d4", "groupnormq",
               "addq", "mulq", "nlq_prelu", "conv1dencoderq", "convtr1ddecoderq"]


def _leave_observer(L):
    _, QQ = _mods()
    for m in L.modules():
        if isinstance(m, QQ.GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
        if isinstance(m, QQ.GradientWeightFakeQuantize):
            m.observer_mode = False


def _idx_stats(y, y_ref, lo, hi):
    """bin-index agreement between two quantized outputs of the same quantizer"""
    delta = (hi - lo) / 255.0
    a = np.rint((y - lo) / delta)
    b = np.rint((y_ref - lo) / delta)
    return float(np.mean(a != b)), float(np.abs(a - b).max())


@pytest.mark.parametrize("name", LAYER_NAMES)
def test_layer_goldens_teacher_forced(golden, name):
    """G1: fed the reference's recorded input, every LayerQ reproduces the reference output with
    <= 2e-3 of the bin indices off by one (tiny tensors: one flip = 3e-4) and matching gradients."""
    g = golden("layers")
    L = _build_layer(name, g)
    _leave_observer(L)
    ins, i = [], 0
    while f"{name}.in{i}" in g.files:
        ins.append(T(g[f"{name}.in{i}"]).cuda().requires_grad_(True))
        i += 1
    y = L(*ins)
    y.backward(T(g[f"{name}.gout"]).cuda())
    out, ref = y.detach().cpu().numpy(), g[f"{name}.out"]
    sd = L.state_dict()
    qs = [k[:-len(".min_range")] for k in sd if k.endswith("activation_fake_quantize.min_range") or k.endswith("_residual.min_range")]
    lo, hi = float(sd["activation_fake_quantize.min_range"]), float(sd["activation_fake_quantize.max_range"])
    if name == "convtr1ddecoderq":
        for ch, key in enumerate(("activation_fake_quantize", "activation_fake_quantize_residual")):
            frac, dmax = _idx_stats(out[ch], ref[ch], float(sd[key + ".min_range"]), float(sd[key + ".max_range"]))
            assert dmax <= 1 and frac <= 5e-3, (ch, frac, dmax)
    else:
        frac, dmax = _idx_stats(out, ref, lo, hi)
        assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
    nflip = int((np.abs(out - ref) > 1e-6).sum())
    for i, x in enumerate(ins):
        if f"{name}.gin{i}" in g.files:
            gin = x.grad.cpu().numpy()
            bad = np.abs(gin - g[f"{name}.gin{i}"]) > (1e-4 + 1e-4 * np.abs(g[f"{name}.gin{i}"]))
            assert bad.mean() <= 2e-3 + 4.0 * nflip / gin.size, (i, bad.mean())
    for k in g.files:
        if k.startswith(name + ".grad."):
            p = dict(L.named_parameters())[k[len(name) + 6:]]
            ref_g = g[k]
            tol = 2e-3 * (np.abs(ref_g).max() + 1e-6) + 0.02 * nflip * np.abs(ref_g).max()
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref_g, rtol=2e-3, atol=tol, err_msg=k)


TINY = dict(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)


def _tiny_pair(g, prefix="sd0."):
    from fqss_amd.smoke import build_pair
    model, fmodel = build_pair("cuda", 0, **TINY)
    model.load_state_dict({k[len(prefix):]: T(g[k]) for k in g.files if k.startswith(prefix)}, strict=True)
    fmodel.load_state_dict({k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}, strict=True)
    return model, fmodel


def test_state_dict_layout(golden):
    g = golden("tiny_step")
    model, _ = _tiny_pair(g)
    assert list(model.state_dict().keys()) == list(g["sd_keys"])


def test_tiny_training_vs_reference_goldens(golden):
    """53 QAT steps of the tiny ConvTasNetQ on the GPU vs the REAL reference's run.

    Gate hierarchy (SURVEY.md A.4 -- the fake-quantized net is chaotic at bin level: from step 2 the
    weights sit on 8-bit grids, so 1e-7 differences flip weight bins and Adam amplifies them; the
    reference's own trajectory differs by 0.58 dB at step 50 between this container's CPU and the GPU
    box's EPYC, tools/diag_tiny.py):
      steps 1-2   G2: loss/KD 1e-5 relative, est 1e-4, SI-SDR 1e-3 dB, EVERY parameter gradient within
                  2e-3 relative, grad-norm 1e-4, observer EMA ranges 2e-5;
      steps 50-53 G3-ii: statistical -- loss within the reference's own cross-machine spread."""
    from fqss_amd.runtime import KDTrainStep
    g = golden("tiny_step")
    model, fmodel = _tiny_pair(g)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0)
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    first_loss = None
    for s in range(1, 54):
        r = step(x, tgt)
        p = f"s{s}."
        first_loss = first_loss if first_loss is not None else r["loss"].item()
        if p + "loss" not in g.files:
            continue
        est = r["est"].cpu().numpy()
        if s <= 2:
            np.testing.assert_allclose(r["loss"].item(), g[p + "loss"], rtol=1e-5, err_msg=p)
            np.testing.assert_allclose(r["kd"].item(), g[p + "kd"], rtol=1e-5, err_msg=p)
            # w = 10**(dSI-SDR/10): the 1e-3 dB SI-SDR tolerance is ln(10)/10 * 1e-3 = 2.3e-4 relative on w
            np.testing.assert_allclose(r["w"].cpu().numpy(), g[p + "w"], rtol=2.3e-4, err_msg=p)
            np.testing.assert_allclose(r["sisdr"].cpu().numpy().mean(), float(O.si_sdr_db(T(g[p + "est"]), T(g["tgt"]))),
                                       atol=1e-3, err_msg=p)
            np.testing.assert_allclose(est, g[p + "est"], rtol=1e-4, atol=2e-6, err_msg=p)
            np.testing.assert_allclose(r["gnorm"].item(), g[p + "gnorm"], rtol=1e-4, err_msg=p)
            coef = min(1.0, 5.0 / (float(g[p + "gnorm"]) + 1e-6))     # goldens hold the clipped gradients
            n_checked = 0
            for name, prm in model.named_parameters():
                k = p + "grad." + name
                if k in g.files:
                    ref = g[k] / coef
                    err = np.linalg.norm(prm.grad.cpu().numpy() - ref) / (np.linalg.norm(ref) + 1e-12)
                    assert err <= 2e-3, (k, err)
                    n_checked += 1
                else:
                    assert float(prm.grad.abs().max()) == 0.0, name      # reference: grad is None
            assert n_checked >= 30
            for k in g.files:
                if k.startswith(p + "post_sd.") and (k.endswith("min_range") or k.endswith("max_range")):
                    got = model.state_dict()[k[len(p + "post_sd."):]].cpu().numpy()
                    if "weight_fake_quantize" in k:
                        # learned by Adam from step 2 on; Adam normalises, so a range gradient at the
                        # 1e-8 cancellation floor can move a range by a fraction of lr in either direction
                        np.testing.assert_allclose(got, g[k], rtol=0, atol=2e-4 * s, err_msg=k)
                    else:
                        np.testing.assert_allclose(got, g[k], rtol=2e-5, atol=1e-7, err_msg=k)   # EMA observer
        else:
            # chaotic region: the oracle itself (same math, other CPU) lands 0.6-1.2 dB from these goldens
            assert abs(r["loss"].item() - float(g[p + "loss"])) <= 3.5, (s, r["loss"].item(), float(g[p + "loss"]))  # dB
            sis = float(O.si_sdr_db(torch.from_numpy(est), T(g["tgt"])))
            sis_ref = float(O.si_sdr_db(T(g[p + "est"]), T(g["tgt"])))
            assert abs(sis - sis_ref) <= 3.5, (s, sis, sis_ref)
    assert first_loss > 20.0 and r["loss"].item() < 3.0      # the QAT loop trains (23.5 dB -> ~0 dB like the reference)


def test_tiny_step51_end_to_end_from_reference_state(golden):
    """first QUANTIZING step, started from the reference's own state after 50 steps: free-running
    forward through all 8-bit quantizers; tiny net => few cascaded flips => loss within 0.05 dB."""
    g = golden("tiny_step")
    from fqss_amd import kernels as K
    model, fmodel = _tiny_pair(g, prefix="s50.post_sd.")
    _leave_observer(model)
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    with torch.no_grad():
        est = model(x)
        fest = fmodel(x)
    out, w, sisdr, _ = K.kd_loss(est, fest, tgt, 0.1, want_grad=False)
    np.testing.assert_allclose(fest.cpu().numpy(), g["s51.fest"], rtol=1e-4, atol=2e-6)
    assert abs(out[0].item() - float(g["s51.loss"])) <= 0.05, (out[0].item(), float(g["s51.loss"]))
    sis_ref = float(O.si_sdr_db(T(g["s51.est"]), T(g["tgt"])))
    assert abs(float(sisdr.mean()) - sis_ref) <= 0.1


def test_tiny_step51_teacher_forced(golden):
    """G1 inside the real network: state after 50 steps, feed each LayerQ the reference's recorded
    input of step 51 (first quantizing step) and compare bin indices."""
    from fqss_amd.quantization.qat import qat_layers as QL
    g = golden("tiny_step")
    model, _ = _tiny_pair(g, prefix="s50.post_sd.")
    _leave_observer(model)
    tot, bad = 0, 0
    worst = 0.0
    with torch.no_grad():
        for name in g["layer_names"]:
            name = str(name)
            if name.endswith("residual_error_block"):
                continue   # called with the decoder's tensors; covered through `decoder`
            mod = dict(model.named_modules())[name]
            ins, j = [], 0
            while f"s51.actin{j}.{name}" in g.files:
                ins.append(T(g[f"s51.actin{j}.{name}"]).cuda())
                j += 1
            out = mod(*ins).cpu().numpy()
            ref = g[f"s51.act.{name}"]
            sd = mod.state_dict()
            keys = ["activation_fake_quantize"] + (["activation_fake_quantize_residual"] if name == "decoder" else [])
            for ch, key in enumerate(keys):
                o, r = (out[ch], ref[ch]) if name == "decoder" else (out, ref)
                frac, dmax = _idx_stats(o, r, float(sd[key + ".min_range"]), float(sd[key + ".max_range"]))
                tot += o.size
                bad += frac * o.size
                worst = max(worst, dmax)
    assert worst <= 1, worst
    assert bad / tot <= 1e-3, (bad, tot)


def test_teacher_forward_matches_oracle():
    """float path (BYPASS kernels) at a mid size"""
    from fqss_amd.smoke import build_pair
    kw = dict(n_spks=2, kernel_size=16, stride=8, n_filters=64, bn_chan=32, hid_chan=64, n_blocks=3, n_repeats=2)
    _, fmodel = build_pair("cuda", 1, **kw)
    x, _ = O.synth_batch(3, 4000, seed=2)
    with torch.no_grad():
        y = fmodel(x.cuda()).cpu()
    ref = O.TeacherConvTasNet({k: v.cpu() for k, v in fmodel.state_dict().items()}, layers_per_stack=3)(x)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-4, atol=1e-6)


def test_full_size_step_properties():
    """cfg 2 (B=8, T=32000, full ConvTasNet): size-independent properties of one observer-phase step
    and one quantizing step -- finite loss, every activation on its 8-bit grid, gradients finite,
    linearity of the combiner, and the observer ranges bracketing the data."""
    from fqss_amd.quantization.qat import qat_quant as QQ
    from fqss_amd.runtime import KDTrainStep
    from fqss_amd.smoke import build_pair
    model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
    assert len(model.state_dict()) == 948
    x, tgt = O.synth_batch(8, 32000, seed=0)
    x, tgt = x.cuda(), tgt.cuda()
    step = KDTrainStep(model, fmodel)
    r = step(x, tgt)
    assert torch.isfinite(r["loss"]).item() and torch.isfinite(step.arena.flat_g).all().item()
    assert r["est"].shape == (8, 2, 32000)
    # jump to the quantizing phase
    for m in model.modules():
        if isinstance(m, QQ.GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
            assert (m.max_range >= m.min_range).item()
    grid_checked = []

    def hook(mod, inp, out):
        q = mod.activation_fake_quantize
        if isinstance(q, QQ.GradientActivationFakeQuantize) and out.dim() == 3 and len(grid_checked) < 6:
            # evaluated on the CPU like the reference: torch's GPU `x / 255` is x * (1/255), not IEEE division
            from fqss_amd import ops
            # inside KDTrainStep the student runs codes-only: intermediate fp32 tensors are carriers -> decode
            lo, hi, o = q.min_range.detach().cpu(), q.max_range.detach().cpu(), ops.real(out).detach()[:2].cpu()
            delta = (hi - lo) / 255
            c = torch.round((o - lo) / delta)
            assert (c >= 0).all() and (c <= 255).all()
            assert torch.equal(delta * c + lo, o)          # exactly on the 8-bit grid
            grid_checked.append(1)

    from fqss_amd.quantization.qat.qat_layers import LayerQ
    hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, LayerQ)]
    r = step(x, tgt)
    for h in hs:
        h.remove()
    assert len(grid_checked) == 6
    assert torch.isfinite(r["loss"]).item() and torch.isfinite(step.arena.flat_g).all().item()
    assert float(step.arena.flat_g.abs().max()) > 0


@pytest.mark.parametrize("fixture,B,T_", [("cfg1_step", 2, 8000), ("cfg2_step", 8, 32000)])
def test_full_size_vs_reference_goldens(golden, fixture, B, T_, n_steps=52, device="cuda"):
    """F7 of SURVEY 8(c): the FULL-SIZE ConvTasNetQ (5.1 M parameters, 24 TCN blocks) at cfg 1 (B=2, T=8000) and at
    cfg 2 (B=8, T=32000 -- the benchmark's own size) against digests of the real reference's run from the same
    name-keyed weights (tests/helpers_cfg1.py).
      step 1      (float arithmetic, observers recording): G2 tolerances -- loss / KD / task 1e-5 relative, est 1e-4,
                  SDR weights 2.3e-4 (= 1e-3 dB), clipped global gradient norm 1e-4, EVERY per-parameter gradient norm 2e-3;
      step 2      (weights quantized after Adam's sign-like first update): 2e-4 / 5e-3 / median gradient norm 1e-2;
      steps 51-52 (all 200 activation quantizers live; chaotic at bin level): statistical, 2 dB (typically 0.1-0.3)."""
    from tests.helpers_cfg1 import cfg1_fill
    from fqss_amd.data import synth_batch
    from fqss_amd.runtime import KDTrainStep
    from fqss_amd.smoke import build_pair
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", fixture + ".npz")):
        pytest.skip(fixture + ".npz not generated (tools/make_goldens.py --only cfg2 takes ~1 h of reference CPU time)")
    g = golden(fixture)
    model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
    cfg1_fill(fmodel, "T.")
    cfg1_fill(model, "S.")
    names = [k for k, _ in model.named_parameters()]
    assert names == list(g["param_names"]) and [k for k, _ in fmodel.named_parameters()] == list(g["tparam_names"])
    np.testing.assert_allclose([float(p.detach().double().sum()) for _, p in model.named_parameters()], g["param_sum"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([float(p.detach().double().sum()) for _, p in fmodel.named_parameters()], g["tparam_sum"], rtol=1e-9, atol=1e-9)
    x, tgt = synth_batch(B, T_, seed=0, device=device)
    np.testing.assert_allclose(float(x.double().sum()), float(g["x_sum"]), rtol=1e-9)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0)
    for s in range(1, n_steps + 1):        # (tests/test_cpu_backend.py runs steps 1-2 of cfg 1 through the CPU backend)
        r = step(x, tgt)
        p = f"s{s}."
        if p + "loss" not in g.files:
            continue
        if s <= 2:
            # Step 2 is NOT a tight gate at this size: Adam's first update is -lr*g/(|g|+eps) = -+1e-3 for every weight
            # whatever |g| is, so the SIGN of each near-zero gradient (noise level) decides a 2e-3 move, and the weights
            # are then rounded to their 8-bit grids.  The CPU oracle itself is 2.5e-5 (loss) off the reference there;
            # measured here: loss 1.3e-5, KD 7e-5, grad-norm 1.7e-3, per-tensor gradient norms median 2e-3.
            f = 1.0 if s == 1 else 20.0
            for k in ("loss", "kd", "task"):
                np.testing.assert_allclose(r[k].item(), g[p + k], rtol=1e-5 * f, err_msg=p + k)
            np.testing.assert_allclose(r["w"].cpu().numpy(), g[p + "w"], rtol=2.3e-4 * f, err_msg=p)
            np.testing.assert_allclose(r["gnorm"].item(), g[p + "gnorm"], rtol=(1e-4 if s == 1 else 5e-3), err_msg=p)
            if s == 1 and p + "est" in g.files:
                ref = g[p + "est"]
                np.testing.assert_allclose(r["est"].cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()), err_msg=p)
            coef = min(1.0, 5.0 / (float(g[p + "gnorm"]) + 1e-6))     # the fixture holds the clipped gradients' norms
            bad, rel = [], []
            for (name, prm), ref in zip(model.named_parameters(), g[p + "grad_norm"]):
                got = float(prm.grad.double().norm())
                if ref < 0:
                    assert got == 0.0, name                          # reference: grad is None
                    continue
                # A PReLU slope's gradient is ONE cancelling sum over the 8 M elements of its tensor: sum|terms| ~ 200
                # for a result of 4e-3..1.5 (condition number up to 5e4; tools/diag_slope_grad.py: the kernel equals an
                # fp64 recomputation from its own inputs to 1e-8), so fp32-level (1e-6) differences upstream move it by
                # ~2e-4 ABSOLUTE.  Scalars therefore get an absolute floor; tensors keep the relative bound.
                tol = 1e-2 if prm.numel() == 1 else 2e-3
                floor = 1e-3 if prm.numel() == 1 else 1e-9
                rel.append(abs(got - ref / coef) / (ref / coef + 1e-12))
                if abs(got - ref / coef) > tol * (ref / coef) + floor:
                    bad.append((name, got, ref / coef))
            if s == 1:
                assert not bad, bad[:5]                               # EVERY parameter (measured: max 2.3e-4)
            else:
                assert float(np.median(rel)) <= 2e-2, float(np.median(rel))
        else:
            # chaotic at bin level and the weight-gradient atomics add run-to-run noise: typically 0.1-0.3 dB, bound 2 dB
            assert abs(r["loss"].item() - float(g[p + "loss"])) <= 2.0, (s, r["loss"].item(), float(g[p + "loss"]))
            np.testing.assert_allclose(10 * np.log10(r["w"].cpu().numpy()), 10 * np.log10(g[p + "w"]), atol=2.0, err_msg=p)


def test_hipgraph_replay_matches_eager(golden):
    """the captured step (two hipGraphs) must reproduce the eager step: same state in, same loss /
    parameters out (the only run-to-run noise left is the fp32 atomics of the weight gradients)"""
    from fqss_amd.runtime import KDTrainStep
    g = golden("tiny_step")
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    runs = []
    for use_graph in (False, True):
        model, fmodel = _tiny_pair(g, prefix="s50.post_sd.")
        _leave_observer(model)
        step = KDTrainStep(model, fmodel)
        losses = [step(x, tgt)["loss"].item()]          # eager step (all Adam clocks start)
        if use_graph:
            step.capture(x, tgt, warmup=1)
        else:
            step(x, tgt)
        for _ in range(3):
            losses.append(step(x, tgt)["loss"].item())
        runs.append((losses, step.arena.flat_p.clone()))
    (l0, p0), (l1, p1) = runs
    np.testing.assert_allclose(l0[0], l1[0], rtol=1e-6)
    np.testing.assert_allclose(l0[1:], l1[1:], atol=0.05)          # dB; chaotic after the first quantized update
    assert float((p0 - p1).abs().max()) < 5e-3


def test_batched_quant_tables_match_per_module_path(golden):
    """multi-tensor weight fake-quant / range-flush kernels (runtime.QuantTables) vs the per-module
    autograd path: same state, same batch -> same loss and the same flat gradient"""
    from fqss_amd.runtime import KDTrainStep
    g = golden("tiny_step")
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    grads, losses = [], []
    for batched in (False, True):
        model, fmodel = _tiny_pair(g, prefix="s50.post_sd.")
        _leave_observer(model)
        step = KDTrainStep(model, fmodel)
        if not batched:
            step._quant_tables = lambda: None
        r = step._fwd_bwd(x, tgt)
        assert (step.tables is not None) == batched
        losses.append(r["loss"].item())
        grads.append(step.arena.flat_g.clone())
    np.testing.assert_allclose(losses[0], losses[1], rtol=1e-6)
    ref = grads[0]
    err = float((grads[0] - grads[1]).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()) + 1e-7, err


def test_fused_teacher_chain_matches_oracle_and_module_path():
    """runtime.TeacherRunner (3 fused kernels per block, bf16 3x3-split GEMMs) vs the oracle's float
    teacher and vs the module-by-module HIP path"""
    from fqss_amd.runtime import TeacherRunner
    from fqss_amd.smoke import build_pair
    for kw, B, T_ in ((dict(n_spks=2, kernel_size=16, stride=8, n_filters=64, bn_chan=32, hid_chan=64, n_blocks=3, n_repeats=2), 3, 4000),
                      (dict(n_spks=2, kernel_size=16, stride=8), 2, 8000)):
        _, fmodel = build_pair("cuda", 1, **kw)
        tr = TeacherRunner(fmodel)
        assert tr.ok
        x, _ = O.synth_batch(B, T_, seed=2)
        y = tr(x.cuda()).cpu()
        with torch.no_grad():
            y_mod = fmodel(x.cuda()).cpu()
        ref = O.TeacherConvTasNet({k: v.cpu() for k, v in fmodel.state_dict().items()},
                                  layers_per_stack=kw.get("n_blocks", 8))(x)
        scale = float(ref.abs().max())
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=2e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(y.numpy(), y_mod.numpy(), rtol=2e-4, atol=2e-5 * scale)
        # SI-SDR of the two teacher outputs against each other must be essentially infinite (> 80 dB)
        assert float(O.si_sdr_db(y, ref)) > 70.0


@pytest.mark.parametrize("name", ["bn1d_train", "bn2d_train", "bn1d_eval", "bn2d_eval"])
def test_batchnormq_layer_goldens(golden, name):
    """BatchNormQ (qat_layers.py:472-486; VERDICT r03 missing #3: the last stub of the quantization.qat surface) on csrc/batchnorm.hip,
    fed the reference's recorded input: BatchNorm1d [B, C, M] and BatchNorm2d [B, C, H, W], training mode (batch statistics; the
    running estimates and num_batches_tracked after the call must be the reference's) and eval mode (running statistics): output bins
    (<= 2e-3 off by one), input gradient, gamma / beta / range gradients (tools/make_goldens_bn.py -> tests/golden/bn_layers.npz)"""
    QL, QQ = _mods()
    g = golden("bn_layers")
    sd1 = {k[len(name) + 5:]: T(g[k]) for k in g.files if k.startswith(name + ".sd1.")}
    C = sd1["batchnorm.weight"].shape[0]
    bn = nn.BatchNorm1d(C) if name.startswith("bn1d") else nn.BatchNorm2d(C, momentum=0.3, eps=1e-3)
    L = QL.BatchNormQ(bn, gradient_based=True, act_quant=True)
    assert list(L.state_dict().keys()) == [k[len(name) + 5:] for k in g.files if k.startswith(name + ".sd1.")]
    L.load_state_dict(sd1, strict=True)
    L = L.cuda().train(name.endswith("train"))
    _leave_observer(L)
    x = T(g[name + ".in0"]).cuda().requires_grad_(True)
    y = L(x)
    y.backward(T(g[name + ".gout"]).cuda())
    out, ref = y.detach().cpu().numpy(), g[name + ".out"]
    lo, hi = float(sd1["activation_fake_quantize.min_range"]), float(sd1["activation_fake_quantize.max_range"])
    frac, dmax = _idx_stats(out, ref, lo, hi)
    assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
    nflip = int((np.abs(out - ref) > 1e-6).sum())
    gin, gref = x.grad.cpu().numpy(), g[name + ".gin0"]
    bad = np.abs(gin - gref) > (1e-4 + 1e-4 * np.abs(gref))
    assert bad.mean() <= 2e-3 + 4.0 * nflip / gin.size, bad.mean()
    for k in g.files:
        if k.startswith(name + ".grad."):
            p = dict(L.named_parameters())[k[len(name) + 6:]]
            ref_g = g[k]
            tol = 2e-3 * (np.abs(ref_g).max() + 1e-6) + 0.02 * nflip * np.abs(ref_g).max()
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref_g, rtol=2e-3, atol=tol, err_msg=k)
    after = L.state_dict()
    for k in ("batchnorm.running_mean", "batchnorm.running_var", "batchnorm.num_batches_tracked"):
        np.testing.assert_allclose(after[k].cpu().numpy(), g[f"{name}.sd2.{k}"], rtol=2e-6, atol=1e-7, err_msg=k)
