"""GPU parity (G0): HIP fake-quant kernels vs the oracle and the reference-generated goldens.
Integer bin indices and de-quantised values must be BIT-EXACT given the same pre-quant input."""
import numpy as np
import pytest
import torch

import oracle.fqss_oracle as O
from tests import oracle_c as OC

pytestmark = pytest.mark.gpu

K = None


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    global K
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from fqss_amd import kernels
    K = kernels
    yield


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def obs_ws():
    w = torch.empty(2, dtype=torch.int32, device="cuda")
    K.obs_reset(w)
    return w


def test_golden_fq_act_bit_exact(golden):
    g = golden("fq_act")
    for i in range(int(g["n_cases"])):
        x = dev(g[f"x{i}"])
        lo, hi = dev(g[f"range{i}"][:1]), dev(g[f"range{i}"][1:])
        y, idx = K.actq_fwd(x, K.ACT_NONE, None, K.Q_QUANT, lo, hi, None, want_idx=True, dense_idx=True)
        assert np.array_equal(idx.cpu().numpy(), g[f"idx{i}"])
        assert np.array_equal(y.cpu().numpy(), g[f"y{i}"])
        gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
        gx = K.actq_bwd(x, dev(g[f"g{i}"]), K.ACT_NONE, None, K.Q_QUANT, lo, hi, gacc)
        assert np.array_equal(gx.cpu().numpy(), g[f"gx{i}"])
        ga = gacc.view(-1, 3).sum(0).cpu().numpy()
        np.testing.assert_allclose(ga[0], g[f"gmin{i}"][0], rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(ga[1], g[f"gmax{i}"][0], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(1, 1), (3, 7, 101), (4, 33, 3999), (16, 1, 32000), (2, 1024, 999)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_actq_vs_oracle(shape, act):
    gen = torch.Generator().manual_seed(hash((shape, act)) & 0xFFFF)
    z = torch.randn(*shape, generator=gen) * 0.7 + 0.1
    g = torch.randn(*shape, generator=gen)
    lo, hi = torch.tensor([-0.9]), torch.tensor([1.3])
    slope = torch.tensor([0.25])
    # oracle
    zr = z.clone().requires_grad_(True)
    lo_r, hi_r, sl_r = lo.clone().requires_grad_(True), hi.clone().requires_grad_(True), slope.clone().requires_grad_(True)
    t = zr if act == 0 else (torch.nn.functional.prelu(zr, sl_r) if act == 1 else torch.relu(zr))
    y = O.act_quantize(t, lo_r, hi_r)
    y.backward(g)
    idx_ref = O.act_indices(t.detach(), lo, hi)
    # hip, padded-row layout (vector path) and dense layout (scalar path when rows are unaligned)
    for padded in (True, False):
        zd = K.empty_act(shape, "cuda") if padded else torch.empty(shape, device="cuda")
        zd.copy_(z)
        gd = K.empty_act(shape, "cuda") if padded else torch.empty(shape, device="cuda")
        gd.copy_(g)
        sd = slope.cuda() if act == 1 else None
        out, idx = K.actq_fwd(zd, act, sd, K.Q_QUANT, lo.cuda(), hi.cuda(), None, want_idx=True, dense_idx=True)
        assert torch.equal(idx.cpu(), idx_ref)
        assert torch.equal(out.cpu(), y.detach())
        gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
        gz = K.actq_bwd(zd, gd, act, sd, K.Q_QUANT, lo.cuda(), hi.cuda(), gacc)
        np.testing.assert_allclose(gz.cpu().numpy(), zr.grad.numpy(), rtol=1e-6, atol=1e-7)
        ga = gacc.view(-1, 3).sum(0).cpu().numpy()
        scale = float(g.abs().sum()) * 1e-6 + 1e-5
        np.testing.assert_allclose(ga[0], lo_r.grad.item(), rtol=2e-4, atol=scale)
        np.testing.assert_allclose(ga[1], hi_r.grad.item(), rtol=2e-4, atol=scale)
        if act == 1:
            np.testing.assert_allclose(ga[2], sl_r.grad.item(), rtol=2e-4, atol=scale)


def test_gacc_flush_is_deterministic_and_rezeroes():
    z, g = torch.randn(8, 128, 3999, device="cuda"), torch.randn(8, 128, 3999, device="cuda")
    lo, hi, sl = torch.tensor([-1.0], device="cuda"), torch.tensor([1.5], device="cuda"), torch.tensor([0.25], device="cuda")
    outs = []
    for _ in range(3):
        gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
        K.actq_bwd(z, g, K.ACT_PRELU, sl, K.Q_QUANT, lo, hi, gacc)
        o = [torch.zeros(1, device="cuda") for _ in range(3)]
        K.gacc_flush(gacc, *o)
        assert float(gacc.abs().max()) == 0.0
        outs.append(torch.cat(o).cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])   # bitwise run-to-run reproducible


def test_actq_bias_rowsum():
    B, C, M = 3, 20, 517
    z = torch.randn(B, C, M)
    g = torch.randn(B, C, M)
    gb = torch.zeros(C, device="cuda")
    gz = K.actq_bwd(z.cuda(), g.cuda(), K.ACT_RELU, None, K.Q_BYPASS, None, None, None, gbias=gb, C=C)
    ref = g * (z > 0)
    np.testing.assert_allclose(gz.cpu().numpy(), ref.numpy())
    np.testing.assert_allclose(gb.cpu().numpy(), ref.sum((0, 2)).numpy(), rtol=1e-5, atol=1e-4)


def test_observer_sequence_golden(golden):
    g = golden("observer")
    lo = torch.tensor([-0.5], device="cuda")
    hi = torch.tensor([0.5], device="cuda")
    ws = obs_ws()
    for it in range(50):
        x = dev(g["x"][it])
        y = K.actq_fwd(x, K.ACT_NONE, None, K.Q_OBSERVE, lo, hi, ws)
        K.observer_ema(lo, hi, ws, 0.9)
        assert np.array_equal(y.cpu().numpy(), g["y"][it])
        assert np.array_equal(lo.cpu().numpy(), g["min"][it]), it
        assert np.array_equal(hi.cpu().numpy(), g["max"][it]), it
    for it in range(50, 53):
        y = K.actq_fwd(dev(g["x"][it]), K.ACT_NONE, None, K.Q_QUANT, lo, hi, None)
        assert np.array_equal(y.cpu().numpy(), g["y"][it])


def test_golden_fq_w(golden):
    g = golden("fq_w")
    for i in range(int(g["n_cases"])):
        axis = int(g[f"axis{i}"])
        w = dev(g[f"w{i}"])
        lo, hi = dev(g[f"min{i}"]), dev(g[f"max{i}"])
        omin, omax = torch.empty_like(lo), torch.empty_like(hi)
        K.wq_observe(w, axis, omin, omax)
        assert np.array_equal(omin.cpu().numpy(), g[f"obs_min{i}"])
        assert np.array_equal(omax.cpu().numpy(), g[f"obs_max{i}"])
        y, idx = K.wq_fwd(w, axis, lo, hi, want_idx=True)
        assert np.array_equal(idx.cpu().numpy(), g[f"idx{i}"])
        assert np.array_equal(y.cpu().numpy(), g[f"y{i}"])
        gw, gmin, gmax = K.wq_bwd(w, dev(g[f"g{i}"]), axis, lo, hi)
        assert np.array_equal(gw.cpu().numpy(), g[f"gw{i}"])
        np.testing.assert_allclose(gmin.cpu().numpy(), g[f"gmin{i}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(gmax.cpu().numpy(), g[f"gmax{i}"], rtol=1e-4, atol=1e-6)


def test_fq_w_full_size_vs_c_oracle():
    for shape, axis in [((512, 128, 1), 0), ((512, 1, 3), 0), ((1024, 128, 1), 0), ((512, 2, 16), 0), ((512, 1, 16), 1)]:
        gen = torch.Generator().manual_seed(shape[0] + axis)
        w = torch.randn(*shape, generator=gen) * 0.1
        g = torch.randn(*shape, generator=gen)
        rs = [1] * 3
        rs[axis] = shape[axis]
        lo = -(torch.rand(*rs, generator=gen) * 0.3 + 0.01)
        hi = torch.rand(*rs, generator=gen) * 0.3 + 0.01
        y, idx = K.wq_fwd(w.cuda(), axis, lo.cuda(), hi.cuda(), want_idx=True)
        y_ref, idx_ref = OC.w_fwd(w.numpy(), lo.numpy(), hi.numpy(), axis)
        assert np.array_equal(idx.cpu().numpy(), idx_ref)
        assert np.array_equal(y.cpu().numpy(), y_ref)
        gw, gmin, gmax = K.wq_bwd(w.cuda(), g.cuda(), axis, lo.cuda(), hi.cuda())
        gw_r, gmin_r, gmax_r = OC.w_bwd(w.numpy(), g.numpy(), lo.numpy(), hi.numpy(), axis)
        assert np.array_equal(gw.cpu().numpy(), gw_r)
        np.testing.assert_allclose(gmin.cpu().numpy().ravel(), gmin_r, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gmax.cpu().numpy().ravel(), gmax_r, rtol=1e-4, atol=1e-5)


def test_full_size_properties():
    """cfg-2 sized tensor (8x512x3999): idempotence of the quantizer and index range."""
    x = torch.randn(8, 512, 3999, device="cuda")
    lo, hi = torch.tensor([-2.0], device="cuda"), torch.tensor([2.5], device="cuda")
    y, idx = K.actq_fwd(x, K.ACT_NONE, None, K.Q_QUANT, lo, hi, None, want_idx=True, dense_idx=True)
    y2, idx2 = K.actq_fwd(y, K.ACT_NONE, None, K.Q_QUANT, lo, hi, None, want_idx=True, dense_idx=True)
    assert torch.equal(idx, idx2)            # fq(fq(x)) has the same bins
    assert int((y2 - y).abs().max() * 1e7) <= 3   # and the same values up to 1 ulp of (delta*c+lo)
    delta = (hi - lo) / 255
    assert torch.equal(y, delta * idx.float() + lo)
    # observer min/max == torch min/max (exact)
    ws = obs_ws()
    K.minmax(x, ws)
    lo2, hi2 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    K.observer_ema(lo2, hi2, ws, 0.0)
    assert lo2.item() == x.min().item() and hi2.item() == x.max().item()


def test_fast_division_is_bitwise_ieee():
    """the kernels' 3-instruction division (Markstein correction with y = RN(1/delta)) must equal the IEEE
    division bit for bit: brute force over 2^28 numerators per divisor, incl. exact half-bin edges"""
    from fqss_amd import _lib
    n = 1 << 26
    mism = torch.zeros(1, dtype=torch.int64, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(0)
    for lo, hi in [(-1.3, 1.7), (0.0, 6.0), (-0.5, 0.5), (-0.0371, 0.2113), (-3.1e-4, 7.7e-3), (-117.0, 351.0)]:
        delta = float((torch.tensor(hi) - torch.tensor(lo)) / 255.0)
        for rep in range(4):
            a = (torch.rand(n, device="cuda", generator=gen) * 1.2 - 0.1) * (hi - lo)       # covers [-0.1, 1.1] x range
            if rep == 0:   # exact bin edges (k + 0.5) * delta and their neighbours
                k = torch.arange(0, 256, device="cuda", dtype=torch.float32)
                edges = (k + 0.5) * delta
                a[:256] = edges
                a[256:512] = torch.nextafter(edges, torch.full_like(edges, 1e9))
                a[512:768] = torch.nextafter(edges, torch.full_like(edges, -1e9))
            _lib.call("fqss_selftest_div", a.data_ptr(), n, delta, mism.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(mism.item()) == 0


def test_public_ste_helpers_match_the_reference_formulas():
    """round_ste / floor_ste / grad_sign / grad_scale / clip_ste (qat_quant.py:88-107: `(f(x) - x).detach() + x`): value f(x) bit for
    bit (torch.round is half-to-even), gradient of x (times the scale)"""
    from fqss_amd.quantization.qat import qat_quant as QQ
    gen = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(1000, generator=gen) * 3, torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 0.0, -0.0, 1.0, -2.0])])
    g = torch.randn(x.numel(), generator=gen)
    cases = [(QQ.round_ste, lambda t: (torch.round(t) - t).detach() + t),
             (QQ.floor_ste, lambda t: (torch.floor(t) - t).detach() + t),
             (lambda t: QQ.grad_sign(t, 0.3), lambda t: (torch.sign(t) - t * 0.3).detach() + t * 0.3),
             (lambda t: QQ.grad_scale(t, 0.25), lambda t: (t - t * 0.25).detach() + t * 0.25),
             (lambda t: QQ.clip_ste(t, -1.2, 0.7), lambda t: (torch.clip(t, min=-1.2, max=0.7) - t).detach() + t)]
    for ours, ref in cases:
        xd = x.clone().cuda().requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        y, yr = ours(xd), ref(xr)
        y.backward(g.cuda())
        yr.backward(g)
        # values: the reference's `(f - x) + x` re-rounds f(x) through x's magnitude; the helper returns f(x) itself
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=0, atol=4e-7 * 8)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-7, atol=0)
    assert torch.equal(QQ.round_ste(x.cuda()).cpu(), torch.round(x))
    assert torch.equal(QQ.floor_ste(x.cuda()).cpu(), torch.floor(x))
