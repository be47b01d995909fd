"""GPU parity of every linear/streaming kernel behind the C ABI against torch fp32 on the CPU
(the same ATen ops the reference calls).  Tolerances are fp32 summation-order noise (gate G1)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle.fqss_oracle as O

pytestmark = pytest.mark.gpu
K = None


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    global K
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from fqss_amd import kernels
    K = kernels
    yield


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def padded(t):
    d = K.empty_act(tuple(t.shape), "cuda")
    d.copy_(t)
    return d


def close(a, b, rtol=1e-4, atol=1e-4, msg=""):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol, err_msg=msg)


PW_SHAPES = [(2, 16, 24, 77), (1, 128, 512, 333), (2, 512, 128, 999), (3, 20, 36, 130), (1, 128, 1024, 64)]


@pytest.mark.parametrize("B,Ci,Co,M", PW_SHAPES)
@pytest.mark.parametrize("pad", [True, False])
def test_pwconv(B, Ci, Co, M, pad):
    x, w, b = rnd(B, Ci, M, seed=1), rnd(Co, Ci, 1, seed=2, scale=Ci ** -0.5), rnd(Co, seed=3)
    gz = rnd(B, Co, M, seed=4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv1d(xr, wr, b)
    y.backward(gz)
    conv = padded if pad else (lambda t: t.cuda())
    xd, gzd = conv(x), conv(gz)
    z = K.pwconv_fwd(xd, w.cuda(), b.cuda())
    close(z, y, msg="fwd")
    z2 = K.pwconv_fwd(xd, w.cuda(), None)
    close(z2, F.conv1d(x, w), msg="fwd nobias")
    close(K.pwconv_fwd(xd, w.cuda(), b.cuda(), six=True), y, msg="fwd, six products")
    gx = K.pwconv_bwd_x(gzd, w.cuda(), Ci)
    close(gx, xr.grad, msg="dgrad")
    gw = torch.zeros(Co, Ci, 1, device="cuda")
    K.pwconv_bwd_w(gzd, xd, gw)
    close(gw, wr.grad, rtol=2e-4, atol=2e-3 * (B * M) ** 0.5 / 30, msg="wgrad")


def test_pwconv_exact_integers():
    """asymmetric integer operands: the MFMA fragment maps must be exact (no transposed tiles)"""
    B, Ci, Co, M = 1, 128, 256, 256
    x = torch.arange(Ci * M, dtype=torch.float32).reshape(1, Ci, M) % 7 - 3
    w = (torch.arange(Co * Ci, dtype=torch.float32).reshape(Co, Ci, 1) % 5 - 2) + torch.eye(Co, Ci).unsqueeze(-1)
    y = F.conv1d(x, w)
    z = K.pwconv_fwd(x.cuda(), w.cuda(), None)
    assert torch.equal(z.cpu(), y)
    gx = K.pwconv_bwd_x(y.cuda(), w.cuda(), Ci)
    assert torch.equal(gx.cpu(), torch.einsum("oc,bom->bcm", w[:, :, 0], y))


@pytest.mark.parametrize("B,Ci,Co,M,taps,dil,pad", [(2, 48, 6, 1000, 3, 1, 1), (3, 16, 96, 431, 3, 2, 2), (1, 8, 40, 77, 3, 1, 0), (2, 4, 130, 300, 5, 3, 6),
                                                     (4, 48, 96, 5000, 3, 2, 2), (1, 384, 48, 1723, 3, 1, 1), (2, 12, 7, 33, 3, 2, 4),
                                                     (5, 96, 12, 431, 3, 2, 2), (3, 24, 24, 130, 3, 1, 1), (2, 16, 2, 4099, 3, 4, 4)])
def test_conv1d_stride1_implicit_gemm(B, Ci, Co, M, taps, dil, pad):
    """stride-1 Conv1d without a frame image (fqss_conv1d_s1_fwd / _bwd_w, the data gradient as the same entry on the flipped,
    transposed weight) against F.conv1d and its autograd"""
    x, w, b = rnd(B, Ci, M, seed=1), rnd(Co, Ci, taps, seed=2, scale=(Ci * taps) ** -0.5), rnd(Co, seed=3)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv1d(xr, wr, b, padding=pad, dilation=dil)
    gz = rnd(*y.shape, seed=4)
    y.backward(gz)
    for conv in (padded, lambda t: t.cuda()):
        xd, gzd, wd = conv(x), conv(gz), w.cuda().reshape(Co, Ci * taps)
        close(K.conv1d_s1_fwd(xd, wd, b.cuda(), taps, dil, pad), y, msg="fwd")
        wt = w.cuda().flip(2).permute(1, 0, 2).reshape(Ci, Co * taps).contiguous()
        close(K.conv1d_s1_fwd(gzd, wt, None, taps, dil, dil * (taps - 1) - pad), xr.grad, msg="dgrad")
        if gzd.stride(-2) % 4 == 0:
            gw = torch.zeros(Co, Ci * taps, device="cuda")
            K.conv1d_s1_bwd_w(gzd, xd, gw, taps, dil, pad)
            close(gw.reshape(Co, Ci, taps), wr.grad, rtol=2e-4, atol=2e-3 * (B * M) ** 0.5 / 30, msg="wgrad")


@pytest.mark.parametrize("rows,cols", [(37, 1000), (5, 4099), (2048, 64)])
def test_gelu_in_the_quantizer_pass(rows, cols):
    """act = ACT_GELU: fq(GELU(z)) and its backward in ONE pass each way = the GELU map (k_unary_fwd / _bwd) followed by the quantizer
    pass, bit for bit; all three quantizer modes"""
    from fqss_amd import ops
    z, g = rnd(rows, cols, seed=1, scale=1.5).cuda(), rnd(rows, cols, seed=2).cuda()
    lo, hi = torch.tensor([-0.15], device="cuda"), torch.tensor([2.2], device="cuda")
    t = K.unary_fwd(z, K.UNARY_GELU)
    for qmode in (ops.Q_QUANT, ops.Q_BYPASS):
        y = K.actq_fwd(z, K.ACT_GELU, None, qmode, lo, hi, None)
        assert torch.equal(y, K.actq_fwd(t, K.ACT_NONE, None, qmode, lo, hi, None)), qmode
        ga, gb = (torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda") for _ in range(2))
        gz = K.actq_bwd(z, g, K.ACT_GELU, None, qmode, lo, hi, ga)
        gt = K.actq_bwd(t, g, K.ACT_NONE, None, qmode, lo, hi, gb)
        assert torch.equal(gz, K.unary_bwd(gt, z, K.UNARY_GELU))
        if qmode == ops.Q_QUANT:
            gmn, gmx, hmn, hmx = (torch.zeros(1, device="cuda") for _ in range(4))
            K.gacc_flush(ga, gmn, gmx, None)
            K.gacc_flush(gb, hmn, hmx, None)
            close(gmn, hmn, rtol=1e-6, atol=1e-6)
            close(gmx, hmx, rtol=1e-6, atol=1e-6)
    obs_a, obs_b = (torch.tensor([-1, 0], dtype=torch.int32, device="cuda") for _ in range(2))
    K.actq_fwd(z, K.ACT_GELU, None, ops.Q_OBSERVE, lo, hi, obs_a)
    K.actq_fwd(t, K.ACT_NONE, None, ops.Q_OBSERVE, lo, hi, obs_b)
    assert torch.equal(obs_a, obs_b)


@pytest.mark.parametrize("R,Ci,Co", [(1000, 64, 192), (8500, 256, 1024), (130, 1024, 256), (77, 32, 40)])
def test_rowlin_fwd_on_presplit_weight_planes(R, Ci, Co, monkeypatch):
    """fqss_rowlin_fwd_w3 (weight as three bf16 planes, split once) = fqss_rowlin_fwd, bit for bit; and kernels.rowlin_fwd picks it
    (opt-in: FQSS_W3_CACHE=1) for a frozen parameter under no_grad, re-splitting after the parameter is written to"""
    monkeypatch.setattr(K, "W3_CACHE", True)
    x, b = rnd(R, Ci, seed=1).cuda(), rnd(Co, seed=3).cuda()
    w = torch.nn.Parameter(rnd(Co, Ci, seed=2, scale=Ci ** -0.5).cuda())
    ref = K.rowlin_fwd(x, w, b)                         # grad mode: the on-the-fly split
    assert getattr(w, "_fqss_w3", None) is None
    with torch.no_grad():
        z = K.rowlin_fwd(x, w, b)
        assert w._fqss_w3[0] == w._version and torch.equal(z, ref)
        w.mul_(0.5)
        z2 = K.rowlin_fwd(x, w, b)
    assert w._fqss_w3[0] == w._version and torch.equal(z2, K.rowlin_fwd(x, w, b))


@pytest.mark.parametrize("rows,cols", [(37, 1000), (2048, 256)])
def test_relu_behind_the_quantizer_in_its_pass(rows, cols):
    """act = ACT_POST_RELU: relu(fq(z)) and its backward in ONE pass each way = the quantizer pass followed by a ReLU pass, bit for bit"""
    from fqss_amd import ops
    z, g = rnd(rows, cols, seed=1, scale=1.5).cuda(), rnd(rows, cols, seed=2).cuda()
    lo, hi = torch.tensor([-0.9], device="cuda"), torch.tensor([1.7], device="cuda")
    for qmode in (ops.Q_QUANT, ops.Q_OBSERVE):
        oa, ob = (torch.tensor([-1, 0], dtype=torch.int32, device="cuda") for _ in range(2))
        y1 = K.actq_fwd(z, K.ACT_NONE, None, qmode, lo, hi, ob)
        y = K.actq_fwd(z, K.ACT_POST_RELU, None, qmode, lo, hi, oa)
        assert torch.equal(y, K.actq_fwd(y1, K.ACT_RELU, None, ops.Q_BYPASS, lo, hi, None)) and torch.equal(oa, ob)
        ga, gb = (torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda") for _ in range(2))
        gz = K.actq_bwd(z, g, K.ACT_POST_RELU, None, qmode, lo, hi, ga)
        g1 = K.actq_bwd(y1, g, K.ACT_RELU, None, ops.Q_BYPASS, lo, hi, None)
        assert torch.equal(gz, K.actq_bwd(z, g1, K.ACT_NONE, None, qmode, lo, hi, gb))
        if qmode == ops.Q_QUANT:
            r = [torch.zeros(1, device="cuda") for _ in range(4)]
            K.gacc_flush(ga, r[0], r[1], None)
            K.gacc_flush(gb, r[2], r[3], None)
            close(r[0], r[2], rtol=3e-4, atol=1e-6)      # (fp32 per-thread partial sums, grouped differently by the two launch shapes)
            close(r[1], r[3], rtol=3e-4, atol=1e-6)


@pytest.mark.parametrize("B,C,M", [(3, 5, 1001), (2, 48, 4100), (1, 7, 64)])
def test_glu_in_the_quantizer_pass(B, C, M):
    """fqss_gluq_fwd / _bwd: fq(GLU(z)) in one pass each way = k_glu_fwd / _bwd around the quantizer pass, bit for bit (outputs, input
    gradient, observer extrema; the range partials within fp32 summation noise)"""
    from fqss_amd import ops
    z, g = padded(rnd(B, 2 * C, M, seed=1, scale=1.5)), padded(rnd(B, C, M, seed=2))
    lo, hi = torch.tensor([-0.4], device="cuda"), torch.tensor([1.1], device="cuda")
    t = K.glu_fwd(z)
    for qmode in (ops.Q_QUANT, ops.Q_OBSERVE):
        oa, ob = (torch.tensor([-1, 0], dtype=torch.int32, device="cuda") for _ in range(2))
        y = K.gluq_fwd(z, qmode, lo, hi, oa)
        assert torch.equal(y, K.actq_fwd(t, K.ACT_NONE, None, qmode, lo, hi, ob)) and torch.equal(oa, ob)
        ga, gb = (torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda") for _ in range(2))
        gz = K.gluq_bwd(z, g, qmode, lo, hi, ga)
        gt = K.actq_bwd(t, g, K.ACT_NONE, None, qmode, lo, hi, gb)
        assert torch.equal(gz, K.glu_bwd(z, gt))
        if qmode == ops.Q_QUANT:
            r = [torch.zeros(1, device="cuda") for _ in range(4)]
            K.gacc_flush(ga, r[0], r[1], None)
            K.gacc_flush(gb, r[2], r[3], None)
            close(r[0], r[2], rtol=3e-4, atol=1e-6)
            close(r[1], r[3], rtol=3e-4, atol=1e-6)


def test_pwconv_split_gemms_against_fp64():
    """the channel-first pointwise GEMMs on the bf16 matrix cores (nine exact products, six products, and the batched k_gemm_x3 forms
    of the two gradients) against fp64: all at the level of an fp32 GEMM (torch's own result measured beside them)"""
    B, Ci, Co, M = 3, 384, 768, 1336
    x, w, gz = rnd(B, Ci, M, seed=11), rnd(Co, Ci, 1, seed=12, scale=Ci ** -0.5), rnd(B, Co, M, seed=13)
    xd, wd, gzd = x.cuda(), w.cuda(), gz.cuda()
    x64, w64, gz64 = xd.double(), wd.double()[:, :, 0], gzd.double()
    def err(a, ref):
        return float((a.double() - ref).norm() / ref.norm())
    y64 = torch.einsum("oc,bcm->bom", w64, x64)
    e9, e6, et = err(K.pwconv_fwd(xd, wd, None), y64), err(K.pwconv_fwd(xd, wd, None, six=True), y64), err(F.conv1d(xd, wd), y64)
    gx64 = torch.einsum("oc,bom->bcm", w64, gz64)
    egx, egxt = err(K.pwconv_bwd_x(gzd, wd, Ci), gx64), err(torch.einsum("oc,bom->bcm", wd[:, :, 0], gzd), gx64)
    gw = torch.zeros(Co, Ci, 1, device="cuda")
    K.pwconv_bwd_w(gzd, xd, gw)
    gw64 = torch.einsum("bom,bcm->oc", gz64, x64)
    egw, egwt = err(gw[:, :, 0], gw64), err(torch.einsum("bom,bcm->oc", gzd, xd), gw64)
    print(f"fwd: nine {e9:.2e} six {e6:.2e} torch {et:.2e}; dgrad {egx:.2e} torch {egxt:.2e}; wgrad {egw:.2e} torch {egwt:.2e}")
    assert e9 <= 3e-7 and e6 <= 6e-7 and egx <= 6e-7 and egw <= 1e-6


@pytest.mark.parametrize("B,C,M,dil", [(2, 32, 77, 1), (2, 32, 77, 4), (1, 512, 999, 128), (3, 7, 130, 2)])
def test_dwconv(B, C, M, dil):
    x, w, b, gz = rnd(B, C, M, seed=1), rnd(C, 1, 3, seed=2), rnd(C, seed=3), rnd(B, C, M, seed=4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv1d(xr, wr, b, padding=dil, dilation=dil, groups=C)
    y.backward(gz)
    for conv in (padded, lambda t: t.cuda()):
        xd, gzd = conv(x), conv(gz)
        close(K.dwconv_fwd(xd, w.cuda(), b.cuda(), dil, dil), y, atol=1e-5)
        close(K.dwconv_bwd_x(gzd, w.cuda(), dil, dil), xr.grad, atol=1e-5)
        gw = torch.zeros(C, 1, 3, device="cuda")
        K.dwconv_bwd_w(gzd, xd, gw, dil, dil)
        close(gw, wr.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,C,M", [(2, 24, 77), (2, 512, 999), (3, 5, 130), (1, 1, 64)])
def test_groupnorm(B, C, M):
    x = rnd(B, C, M, seed=1) * 1.7 + 0.3
    gm, bt, gz = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, M, seed=4)
    xr, gr, br = x.clone().requires_grad_(True), gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    y = F.group_norm(xr, 1, gr, br, 1e-8)
    y.backward(gz)
    for conv in (padded, lambda t: t.cuda()):
        xd, gzd = conv(x), conv(gz)
        z, mr = K.gn_fwd(xd, gm.cuda(), bt.cuda(), 1e-8)
        close(z, y, rtol=1e-5, atol=2e-6)
        gg, gb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        gx = K.gn_bwd(gzd, xd, gm.cuda(), mr, gg, gb)
        close(gx, xr.grad, rtol=1e-4, atol=1e-5)
        close(gg, gr.grad, rtol=1e-4, atol=1e-3)
        close(gb, br.grad, rtol=1e-4, atol=1e-3)


def test_elementwise():
    a, b = rnd(3, 20, 131, seed=1), rnd(3, 20, 131, seed=2)
    for conv in (padded, lambda t: t.cuda()):
        assert torch.equal(K.axpby(conv(a), conv(b), 1.0).cpu(), a + b)
        assert torch.equal(K.axpby(conv(a), conv(b), -1.0).cpu(), a - b)
        mask, feat, gz = rnd(2, 2, 24, 77, seed=3), rnd(2, 24, 77, seed=4), rnd(2, 2, 24, 77, seed=5)
        z = K.mul_bcast_fwd(conv(mask), conv(feat))
        assert torch.equal(z.cpu(), mask * feat.unsqueeze(1))
        gm, gf = K.mul_bcast_bwd(conv(gz), conv(mask), conv(feat))
        assert torch.equal(gm.cpu(), gz * feat.unsqueeze(1))
        close(gf, (gz * mask).sum(1), rtol=1e-6, atol=1e-6)


def test_splitter_golden(golden):
    g = golden("process")
    out = K.splitter2(torch.from_numpy(g["x"]).cuda())
    assert np.array_equal(out.cpu().numpy(), g["pre2"])
    out = K.splitter2(torch.from_numpy(g["x2d"]).cuda())
    assert np.array_equal(out.cpu().numpy(), g["pre2_2d"])
    x = rnd(8, 1, 32000, seed=9, scale=0.1)
    assert torch.equal(K.splitter2(x.cuda()).cpu(), O.split(x, 2))


@pytest.mark.parametrize("N,Ci,Co,M,K_,S", [(2, 2, 24, 77, 16, 8), (3, 1, 40, 130, 16, 8), (2, 2, 512, 999, 16, 8), (2, 1, 16, 50, 32, 16)])
def test_frames_conv_and_wgrad(N, Ci, Co, M, K_, S):
    T = (M - 1) * S + K_
    x, w, gz = rnd(N, Ci, T, seed=1), rnd(Co, Ci, K_, seed=2, scale=0.2), rnd(N, Co, M, seed=3)
    wr = w.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = F.conv1d(xr, wr, None, stride=S)
    y.backward(gz)
    z = K.frames_conv_fwd(x.cuda(), w.cuda(), S)
    close(z, y, rtol=1e-5, atol=1e-5)
    other = padded(rnd(N, Co, M, seed=8))            # the addend form (decoder input gradient + the residual block's): one fp32 add
    assert torch.equal(K.frames_conv_fwd(x.cuda(), w.cuda(), S, add=other), z + other)
    gw = torch.zeros(Co, Ci, K_, device="cuda")
    K.frames_wgrad(padded(gz), x.cuda(), gw, S)
    close(gw, wr.grad, rtol=1e-4, atol=2e-3)
    # the conv's input gradient is the transposed conv of gz (Ci == 1 path of the residual encoder)
    if Ci == 1:
        gx = K.ola_convtr_fwd(padded(gz), w.cuda().reshape(Co, K_), S)
        close(gx, xr.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,C,M,K_,S", [(4, 24, 77, 16, 8), (2, 512, 999, 16, 8), (1, 8, 1, 16, 8), (2, 16, 63, 16, 8), (2, 16, 64, 32, 16),
                                        (2, 30, 253, 16, 8), (1, 7, 757, 32, 16)])
def test_ola_convtr(N, C, M, K_, S):
    x, w = rnd(N, C, M, seed=1), rnd(C, 1, K_, seed=2, scale=0.2)
    g = rnd(N, 1, (M - 1) * S + K_, seed=3)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv_transpose1d(xr, wr, None, stride=S)
    y.backward(g)
    outs = []
    for conv in (padded, lambda t: t.cuda()):      # padded rows: the matrix-core form for (16, 8); dense rows of odd length: one frame per lane
        out = K.ola_convtr_fwd(conv(x), w.cuda(), S)
        close(out, y, rtol=1e-4, atol=1e-4)
        outs.append(out)
    close(outs[0], outs[1], rtol=1e-5, atol=1e-5)  # (16, 8) on aligned rows: matrix-core form, another summation order
    # the coded-input form (student decoder) and the masking form (teacher decoder) against their un-fused chains, bit for bit
    gen = torch.Generator().manual_seed(N + C + M)
    codes = torch.randint(0, 256, (N, C, M), generator=gen, dtype=torch.uint8)
    lo, hi = torch.tensor([-0.83]).cuda(), torch.tensor([1.21]).cuda()
    xc = K.empty_codes((N, C, M), "cuda")
    xc.copy_(codes)
    assert torch.equal(K.ola_convtr_fwd_q(xc, lo, hi, w.cuda(), S), K.ola_convtr_fwd(K.decode(xc, lo, hi), w.cuda(), S))
    # ... and its weight gradient straight from the codes against the generic GEMM on the decoded operand
    gw_q, gw_r = torch.zeros(C, 1, K_, device="cuda"), torch.zeros(C, 1, K_, device="cuda")
    assert K.frames_wgrad1_q(xc, lo, hi, g.cuda(), gw_q, S)
    os.environ["FQSS_FRAMES_WGRAD1"] = "0"
    try:
        K.frames_wgrad(K.decode(xc, lo, hi), g.cuda(), gw_r, S)
    finally:
        del os.environ["FQSS_FRAMES_WGRAD1"]
    close(gw_q, gw_r, rtol=1e-4, atol=1e-5 * float(gw_r.abs().max()) + 1e-5)
    if N % 2 == 0:
        mask, feat = padded(rnd(N // 2, 2, C, M, seed=5)), padded(rnd(N // 2, C, M, seed=6))
        ref = K.ola_convtr_fwd(K.mul_bcast_fwd(mask, feat).reshape(N, C, M), w.cuda(), S)
        assert torch.equal(K.ola_convtr_mul_fwd(mask, feat, w.cuda(), S), ref)
    # decoder backward = framing conv of the output gradient with the same taps
    gx = K.frames_conv_fwd(g.cuda(), w.cuda().reshape(C, 1, K_), S)
    close(gx, xr.grad, rtol=1e-5, atol=1e-5)
    gw = torch.zeros(C, 1, K_, device="cuda")
    K.frames_wgrad(padded(x), g.cuda(), gw, S)
    close(gw, wr.grad, rtol=1e-4, atol=2e-3)


def test_waveform_kernels_next_to_a_second_stream():
    """The kernels of the network's waveform side, launched while a second stream runs the float teacher (how KDTrainStep runs them),
    return bit for bit what they return alone.  Round 2 had a faster decoder kernel that passed every other test and failed this
    property (tools/stress_streams.py tells the story); the student's encoder then differed between two runs of the same step."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stress_streams", os.path.join(root, "tools", "stress_streams.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    counts = mod.main()
    torch.cuda.empty_cache()
    assert counts and all(v == 0 for v in counts.values()), counts


def test_kd_loss_golden_and_oracle(golden):
    g = golden("loss")
    est, fest, tgt = (torch.from_numpy(g[k]).cuda() for k in ("est", "fest", "tgt"))
    out, w, sisdr, gest = K.kd_loss(est, fest, tgt, 0.1)
    np.testing.assert_allclose(w.cpu().numpy(), g["w"], rtol=1e-5)
    np.testing.assert_allclose(out[0].item(), g["loss"], rtol=1e-5)      # KD loss: 1e-5 relative (north-star tolerance)
    np.testing.assert_allclose(out[3].item(), g["kd"], rtol=1e-5)
    np.testing.assert_allclose(out[2].item(), g["task"], rtol=1e-5)
    np.testing.assert_allclose(gest.cpu().numpy(), g["gest"], rtol=2e-4, atol=1e-8)
    np.testing.assert_allclose(sisdr.cpu().numpy(), -g["sdrqs"], atol=1e-3)   # SI-SDR: 1e-3 dB
    # cfg-2 sized batch vs the oracle
    x, s = O.synth_batch(8, 32000, seed=3)
    e = (s + 0.3 * rnd(8, 2, 32000, seed=5, scale=0.05)).requires_grad_(True)
    f = s + 0.2 * rnd(8, 2, 32000, seed=6, scale=0.05)
    loss, kd, task, w_r, sdrs, sdrqs = O.kd_loss(e, f, s)
    loss.backward()
    out, w, sisdr, gest = K.kd_loss(e.detach().cuda(), f.cuda(), s.cuda(), 0.1)
    np.testing.assert_allclose(out[0].item(), loss.item(), rtol=1e-5)
    np.testing.assert_allclose(w.cpu().numpy(), w_r.numpy(), rtol=2e-5)
    np.testing.assert_allclose(sisdr.cpu().numpy(), -sdrqs.numpy(), atol=1e-3)
    np.testing.assert_allclose(gest.cpu().numpy(), e.grad.numpy(), rtol=5e-4, atol=1e-9)


def test_kd_loss_per_sample_speechbrain_objective():
    """fqss_kd_loss_per_sample against the oracle's restatement of the speechbrain env's objective (log per sample, thresholded mean,
    the reference's weight broadcast): loss, weights and dL/d est at B = 1 and B = 2; B = 3 is refused like the reference's broadcast"""
    x, s = O.synth_batch(3, 16000, seed=4)
    e_all = s + 0.3 * rnd(3, 2, 16000, seed=7, scale=0.05)
    e_all[1] = s[1, [1, 0]] + 0.05 * rnd(2, 16000, seed=9, scale=0.05)       # a swapped, much easier sample
    f_all = s + 0.2 * rnd(3, 2, 16000, seed=8, scale=0.05)
    for B in (1, 2):
        e, f, t = e_all[:B], f_all[:B], s[:B]
        _, per, _ = O.kd_loss_speechbrain(e, f, t)
        ths = [None, -30.0, 1e9] + ([0.5 * (per[0] + per[1]).item()] if B == 2 else [])
        for th in ths:
            er = e.clone().requires_grad_(True)
            loss, per_r, w_r = O.kd_loss_speechbrain(er, f, t, threshold=th)
            loss.backward()
            out, w, sisdr, gest = K.kd_loss(e.cuda(), f.cuda(), t.cuda(), 0.1, per_sample=True, threshold=th)
            np.testing.assert_allclose(out[0].item(), loss.item(), rtol=1e-5, err_msg=f"B={B} th={th}")
            np.testing.assert_allclose(w.cpu().numpy(), w_r.numpy(), rtol=2e-5)
            gr = er.grad.numpy()      # near its zero crossings the oracle's fp32 autograd carries noise of ~1e-5 of the gradient's scale
            np.testing.assert_allclose(gest.cpu().numpy(), gr, rtol=5e-4, atol=2e-5 * float(np.abs(gr).max()), err_msg=f"B={B} th={th}")
    with pytest.raises(_lib_error(), match="B must be 1 or n_src"):
        K.kd_loss(e_all.cuda(), f_all.cuda(), s.cuda(), 0.1, per_sample=True)


def test_gradient_gemms_two_piece_split(monkeypatch):
    """The dgrad / wgrad q-GEMMs form EXACT products by default (the fp32 gradient operand in three bf16 pieces): 1.6e-7 .. 3.3e-7 of the
    result's norm against fp64 at the cfg-2 layer shapes.  FQSS_GRAD_PIECES=2 opts into a two-piece operand (truncated head +
    round-to-nearest remainder, 16-17 significant bits, -0.4 ms per cfg-2 step): 5e-6 .. 8e-6 -- bounded here so the option stays honest;
    nothing else in the suite, and never the benchmark, runs on it."""
    torch.manual_seed(0)
    B, M = 2, 3999
    for Ci, Co in ((128, 512), (512, 256)):
        ones = torch.ones(Co, 1, 1, device="cuda") * 0.2
        wc = K.wq_codes(torch.randn(Co, Ci, 1, device="cuda") * 0.05, -ones, ones)
        gz = K.empty_act((B, Co, M), "cuda")
        gz.copy_(torch.randn(B, Co, M, device="cuda") * torch.exp(torch.randn(B, Co, M, device="cuda")) * 1e-4)      # heavy-tailed, like real gradients
        xc = K.empty_codes((B, Ci, M), "cuda").random_(0, 256)
        lo, hi = torch.tensor([-1.0], device="cuda"), torch.tensor([2.0], device="cuda")
        out = {}
        for pieces in ("3", "2"):
            monkeypatch.setenv("FQSS_GRAD_PIECES", pieces)         # (the default, unset, is 3)
            gw = torch.zeros(Co, Ci, device="cuda")
            K.qpw_bwd_w(gz, xc, lo, hi, gw)
            out[pieces] = (K.qpw_bwd_x(gz, wc)[..., :M].clone(), gw)
        monkeypatch.delenv("FQSS_GRAD_PIECES")
        # fp64 references
        w64 = (wc.idx.double() * wc.dw.double()[:, None])                              # [Co, Ci]
        gx64 = torch.einsum("oc,bom->bcm", w64, gz[..., :M].double())
        x64 = xc[..., :M].double() * ((hi - lo).double() / 255.0) + lo.double()
        gw64 = torch.einsum("bom,bcm->oc", gz[..., :M].double(), x64)
        for k, ref in ((0, gx64), (1, gw64)):
            a3, a2 = out["3"][k].double(), out["2"][k].double()
            nrm, rms = float(ref.norm()), float(ref.pow(2).mean().sqrt())
            e3, e2 = float((a3 - ref).norm()) / nrm, float((a2 - ref).norm()) / nrm
            print(f"Ci {Ci} Co {Co} {'dgrad' if k == 0 else 'wgrad'}: |2p-3p|/|ref| {float((a2 - a3).norm()) / nrm:.2e}  max/rms "
                  f"{float((a2 - a3).abs().max()) / rms:.2e}  vs fp64: 3 pieces {e3:.2e}, 2 pieces {e2:.2e}")
            assert e3 <= 1e-6 and e2 <= 1.5e-5, (Ci, Co, k, e3, e2)
            assert float((a2 - a3).abs().max()) <= 5e-4 * rms, (Ci, Co, k, float((a2 - a3).abs().max()) / rms)


@pytest.mark.parametrize("case", ["block", "segment", "ragged", "many"])
def test_grouped_weight_gradients(case):
    """fqss_qpw_bwd_w_group (round 5, VERDICT r04 next #1d): the weight gradients of several quantized 1x1 convolutions in ONE launch
    -- 32 teams of 8 workgroups over the (layer, tile group, 64-frame stage) work list, tiles cut by a team boundary reduced through
    slab slots in part order, NO float atomics -- against fp64 (3e-6 of the norm: exact products, fp32 accumulation like
    fqss_qpw_bwd_w, but over chains of up to 18 k terms per workgroup where that kernel's 64-way split-n adds 1 k-term chains), against the per-layer kernel, accumulating into a non-zero gw, and bit-identical from run to run (the
    per-layer kernel's 2.1 M float atomics are not).  Cases: one TCN block (conv1 128 -> 512 + the res | skip pair, cfg-2 shapes),
    a backward segment of three blocks, ragged shapes (Co / Ci / M off the tile sizes, B = 3, a job smaller than one tile, a pair
    whose halves differ), and 29 jobs (two launches).  Reference: autograd of F.conv1d in Conv1dQ / Conv1dNlQ, qat_layers.py:137-146."""
    dev = "cuda"
    g = torch.Generator().manual_seed({"block": 1, "segment": 2, "ragged": 3, "many": 4}[case])
    if case == "block":
        shapes = [(8, 128, 512, 0, 3999), (8, 512, 128, 128, 3999)]
    elif case == "segment":
        shapes = [(8, 128, 512, 0, 3999), (8, 512, 128, 128, 3999)] * 3
    elif case == "ragged":
        shapes = [(3, 48, 80, 0, 777), (3, 16, 32, 0, 50), (2, 144, 64, 32, 1030), (1, 256, 200, 0, 63), (3, 128, 128, 128, 64), (2, 512, 128, 0, 4001)]
    else:
        shapes = [(2, 32 + 16 * (i % 5), 64 + 32 * (i % 3), 0 if i % 4 else 64, 500 + 97 * i) for i in range(29)]
    lo, hi = torch.tensor([-1.3], device=dev), torch.tensor([2.1], device=dev)
    jobs, refs, gws = [], [], []
    for (B, Ci, Co1, Co2, M) in shapes:
        mk = lambda C: K.empty_act((B, C, M), dev).copy_((torch.randn(B, C, M, generator=g) * torch.exp(torch.randn(B, C, M, generator=g)) * 1e-3).to(dev))
        g1, g2 = mk(Co1), (mk(Co2) if Co2 else None)
        xc = K.empty_codes((B, Ci, M), dev)
        xc.copy_(torch.randint(0, 256, (B, Ci, M), generator=g, dtype=torch.uint8).to(dev))
        gw0 = (torch.randn(Co1 + Co2, Ci, generator=g) * 1e-2).to(dev)          # gw is ACCUMULATED
        x64 = xc[..., :M].double() * ((hi - lo).double() / 255.0) + lo.double()
        gz64 = torch.cat([g1[..., :M], g2[..., :M]], 1).double() if Co2 else g1[..., :M].double()
        refs.append(torch.einsum("bom,bcm->oc", gz64, x64))
        jobs.append((g1, g2, xc))
        gws.append(gw0)
    runs = []
    q = K.WgradQueue()
    for rep in range(3):
        outs = [w.clone() for w in gws]
        for (g1, g2, xc), gw in zip(jobs, outs):
            q.push(g1, g2, xc, lo, hi, gw)
        q.flush()
        torch.cuda.synchronize()
        runs.append(outs)
        assert int(q.ws[:65536].view(torch.int32).abs().max()) == 0          # the arrival tickets are left zero
    for j, (ref, gw0) in enumerate(zip(refs, gws)):
        got = runs[0][j].double() - gw0.double()
        err = float((got - ref).norm()) / float(ref.norm())
        assert err <= 3e-6, (case, j, shapes[j], err)
        assert torch.equal(runs[0][j], runs[1][j]) and torch.equal(runs[0][j], runs[2][j]), (case, j, "not reproducible")
        # the per-layer kernel (split-n + float atomics): same value to fp32 summation-order noise
        B, Ci, Co1, Co2, M = shapes[j]
        one = torch.zeros(Co1 + Co2, Ci, device=dev)
        g1, g2, xc = jobs[j]
        if Co2:
            K.qpw_bwd_w2(g1, g2, xc, lo, hi, one)
        else:
            K.qpw_bwd_w(g1, xc, lo, hi, one)
        assert float((one.double() - got).norm()) <= 4e-6 * float(ref.norm()), (case, j)
    # one gw pushed twice (a layer applied twice in one backward): the second contribution runs on the per-layer kernel at once, the
    # sum is both (the queue used to refuse this at flush, the per-layer path always accumulated it)
    twice = gws[0].clone()
    q.push(jobs[0][0], jobs[0][1], jobs[0][2], lo, hi, twice)
    q.push(jobs[0][0], jobs[0][1], jobs[0][2], lo, hi, twice)
    assert len(q.jobs) == 1
    q.flush()
    torch.cuda.synchronize()
    got2 = twice.double() - gws[0].double()
    assert float((got2 - 2 * refs[0]).norm()) <= 4e-6 * float(refs[0].norm())


def test_grouped_weight_gradients_beside_a_memory_hog():
    """Deterministic kernel-level regression for the round-5 store hazard (the only other test of it is a two-process run
    at rtol 2e-2 that fails only some of the time): k_qwgrad_group hands tiles cut by a team boundary over through slab slots written
    with hand-written 16-byte stores (csrc/qgemm.hip st16_sc1).  Without the wait states behind such a store the NEXT slab address
    landed in the slab in place of data whenever the memory pipe was slow to fetch the store data (profiles/r05_store_hazard.txt).
    Here the grouped launch of a three-block backward segment (cfg-2 shapes) runs 30 times while
    two other streams keep HBM saturated with 1 GB copies; the kernel uses no atomics, so every run must reproduce the bits of a run
    on an idle GPU -- not a tolerance.  tools/r06_store_hazard.sh runs this very test against a library built WITHOUT the wait states (wide and narrow tiles: FQSS_WGRAD_WIDE
    is read once per process)."""
    dev = "cuda"
    g = torch.Generator().manual_seed(2)
    shapes = [(8, 128, 512, 0, 3999), (8, 512, 128, 128, 3999)] * 3
    lo, hi = torch.tensor([-1.3], device=dev), torch.tensor([2.1], device=dev)
    jobs = []
    for (B, Ci, Co1, Co2, M) in shapes:
        mk = lambda C: K.empty_act((B, C, M), dev).copy_((torch.randn(B, C, M, generator=g) * 1e-3).to(dev))
        xc = K.empty_codes((B, Ci, M), dev)
        xc.copy_(torch.randint(0, 256, (B, Ci, M), generator=g, dtype=torch.uint8).to(dev))
        jobs.append((mk(Co1), mk(Co2) if Co2 else None, xc, Co1 + Co2, Ci))
    q = K.WgradQueue()

    def run():
        outs = [torch.zeros(co, ci, device=dev) for *_, co, ci in jobs]
        for (g1, g2, xc, _, _), gw in zip(jobs, outs):
            q.push(g1, g2, xc, lo, hi, gw)
        q.flush()
        return outs

    alone = run()
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in alone)
    src, dst = torch.empty(2, 1 << 28, device=dev), torch.empty(2, 1 << 28, device=dev)        # 2 x 1 GiB each way
    hogs = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = 0
    for rep in range(30):
        for k, hs in enumerate(hogs):
            with torch.cuda.stream(hs):
                for _ in range(3):
                    dst[k].copy_(src[k], non_blocking=True)
        outs = run()
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(a, b)) for a, b in zip(alone, outs))
    assert bad == 0, f"{bad} of {30 * len(jobs)} weight gradients differ from the idle-GPU run"


def test_grouped_row_weight_gradients():
    """fqss_qrow_bwd_w_group (round 5): the coded weight gradients (+ bias column sums) of several row-major linears in one launch per
    tile shape -- the Sepformer layer's four linears, DPTNet's 64-wide ones (a second tile shape), ragged sizes, 40 jobs (two launches)
    -- against fp64 and against fqss_qrow_bwd_wb one by one.  Reference: autograd of F.linear in LinearQ / MultiheadAttentionQ
    (qat_layers.py:521-568, 889-901)."""
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    shapes = [(8500, 256, 768), (8500, 256, 256), (8500, 256, 1024), (8500, 1024, 256), (16500, 64, 256), (16500, 64, 64), (777, 48, 36),
              (130, 128, 200)] + [(1000 + 37 * i, 64 + 32 * (i % 4), 32 + 32 * (i % 7)) for i in range(32)]
    lo, hi = torch.tensor([-1.3], device=dev), torch.tensor([2.1], device=dev)
    q = K.RowWgradQueue()
    refs, outs, biases, jobs = [], [], [], []
    for k, (R, Ci, Co) in enumerate(shapes):
        gz = (torch.randn(R, Co, generator=g) * torch.exp(torch.randn(R, Co, generator=g)) * 1e-3).to(dev)
        xc = torch.randint(0, 256, (R, Ci), generator=g, dtype=torch.uint8).to(dev)
        gw0, gb0 = (torch.randn(Co, Ci, generator=g) * 1e-3).to(dev), (torch.randn(Co, generator=g) * 1e-3).to(dev)
        x64 = xc.double() * ((hi - lo).double() / 255.0) + lo.double()
        refs.append((gz.double().t() @ x64, gz.double().sum(0)))
        gw, gb = gw0.clone(), (gb0.clone() if k % 3 != 2 else None)
        q.push(gz, xc, lo, hi, gw, gb)
        outs.append((gw, gb, gw0, gb0))
        jobs.append((gz, xc))
    q.flush()
    torch.cuda.synchronize()
    for k, ((rw, rb), (gw, gb, gw0, gb0)) in enumerate(zip(refs, outs)):
        ew = float(((gw.double() - gw0.double()) - rw).norm()) / float(rw.norm())
        assert ew <= 2e-6, (k, shapes[k], ew)
        if gb is not None:
            eb = float(((gb.double() - gb0.double()) - rb).norm()) / float(rb.norm())
            assert eb <= 2e-6, (k, shapes[k], eb)
        one, oneb = torch.zeros_like(gw), torch.zeros(gw.shape[0], device=dev)
        K.qrow_bwd_w(jobs[k][0], jobs[k][1], lo, hi, one, oneb)
        assert float((one.double() - (gw.double() - gw0.double())).norm()) <= 2e-6 * float(rw.norm()), k


@pytest.mark.parametrize("B,Ci,Co,M", [(2, 384, 96, 2757), (1, 432, 192, 1000), (3, 32, 48, 130), (2, 768, 384, 431), (1, 3456, 768, 216)])
def test_pointwise_conv_float_input_coded_weight(B, Ci, Co, M):
    """fqss_pwconv_fwd_wq (round 5, k_qgemm<4>): z = (dw Wi) x + b for a FLOAT input and a fake-quantized weight given as int8 codes --
    x in three exact bf16 pieces x one exact plane of codes -- against fp64 on the de-quantized weight (fp32-grade: 2e-6 of the maximum),
    and its data gradient on the same codes (fqss_qpw_bwd_x).  The frame-path convolutions of HTDemucs' student (qat_layers.py:188-293)."""
    dev = "cuda"
    g = torch.Generator().manual_seed(Ci + Co + M)
    w = (torch.randn(Co, Ci, 1, generator=g) * 0.05).to(dev)
    ones = torch.ones(Co, 1, 1, device=dev) * 0.2
    wc = K.wq_codes(w, -ones, ones)
    x = K.empty_act((B, Ci, M), dev).copy_((torch.randn(B, Ci, M, generator=g) * torch.exp(torch.randn(B, Ci, M, generator=g) * 0.5)).to(dev))
    bias = (torch.randn(Co, generator=g) * 0.1).to(dev)
    z = K.pwconv_fwd_wq(x, wc, bias)
    assert z is not None
    wq64 = wc.idx.double() * wc.dw.double()[:, None]
    ref = torch.einsum("oc,bcm->bom", wq64, x[..., :M].double()) + bias.double()[None, :, None]
    err = float((z[..., :M].double() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-6, err
    gz = K.empty_act((B, Co, M), dev).copy_((torch.randn(B, Co, M, generator=g) * 1e-2).to(dev))
    if Co <= 1024 and Co % 16 == 0:
        gx = K.qpw_bwd_x(gz, wc)
        refx = torch.einsum("oc,bom->bcm", wq64, gz[..., :M].double())
        assert float((gx[..., :M].double() - refx).abs().max()) <= 2e-6 * float(refx.abs().max())


def test_pit_sisdr_loss_teacher_free():
    """fqss_pit_sisdr_loss (kd_lambda = 0, mysystem.py:153-156) against the oracle's neg_sisdr_pit: loss 1e-5, per-sample SI-SDR 1e-3 dB,
    dL/d est; one sample has its sources swapped so both permutations are exercised"""
    x, s = O.synth_batch(4, 16000, seed=5)
    e = s + 0.3 * rnd(4, 2, 16000, seed=7, scale=0.05)
    e[2] = s[2, [1, 0]] + 0.1 * rnd(2, 16000, seed=9, scale=0.05)
    er = e.clone().requires_grad_(True)
    loss = O.neg_sisdr_pit(er, s)
    loss.backward()
    out, sisdr, gest = K.pit_sisdr_loss(e.cuda(), s.cuda())
    np.testing.assert_allclose(out[0].item(), loss.item(), rtol=1e-5)
    per = torch.stack([-O.neg_sisdr_pit(e[b:b + 1], s[b:b + 1]) for b in range(4)])
    np.testing.assert_allclose(sisdr.cpu().numpy(), per.numpy(), atol=1e-3)
    gr = er.grad.numpy()
    np.testing.assert_allclose(gest.cpu().numpy(), gr, rtol=5e-4, atol=2e-5 * float(np.abs(gr).max()))


def _lib_error():
    from fqss_amd import _lib
    return _lib.FqssError


def test_adam_clip_vs_torch():
    n = 100003
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.05)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    acc = torch.zeros(1, device="cuda", dtype=torch.float64)
    step = torch.zeros(1, device="cuda", dtype=torch.int32)
    gn = torch.zeros(1, device="cuda")
    for it in range(5):
        gi = g * (1 + it)
        pr.grad = gi.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([pr], 5.0)
        opt.step()
        acc.zero_()
        K.sumsq(gi.cuda(), acc)
        K.adam_clip(p, gi.cuda(), m, v, acc, step, gn, 5.0, 1.0, 1e-3)
        np.testing.assert_allclose(gn.item(), ref_norm.item(), rtol=1e-6)
        np.testing.assert_allclose(p.cpu().numpy(), pr.detach().numpy(), rtol=1e-6, atol=1e-7)
    assert step.item() == 5


# ---------------------------------------------------------------------------------------------
# q-GEMMs: grid-valued pointwise conv on the bf16 matrix cores
# ---------------------------------------------------------------------------------------------
def _q_setup(B, Ci, Co, M, seed=0):
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn(Co, Ci, 1, generator=gen) * Ci ** -0.5
    wlo = -(torch.rand(Co, 1, 1, generator=gen) * 0.2 + 0.05)
    whi = torch.rand(Co, 1, 1, generator=gen) * 0.2 + 0.05
    xlo, xhi = torch.tensor([-0.731]), torch.tensor([1.913])
    x = torch.randn(B, Ci, M, generator=gen) * 0.8 + 0.4
    bias = torch.randn(Co, generator=gen)
    wq = O.weight_quantize(w, wlo, whi)                 # reference fake-quant (fp32)
    xq = O.act_quantize(x, xlo, xhi)
    return w, wlo, whi, xlo, xhi, x, bias, wq, xq


@pytest.mark.parametrize("B,Ci,Co,M", [(2, 16, 32, 77), (1, 128, 512, 333), (2, 512, 128, 999), (1, 128, 1024, 130), (3, 48, 80, 64)])
def test_qgemm_fwd_bwd(B, Ci, Co, M):
    w, wlo, whi, xlo, xhi, x, bias, wq, xq = _q_setup(B, Ci, Co, M, seed=Ci + Co)
    wc = K.wq_codes(w.cuda().contiguous(), wlo.cuda(), whi.cuda())
    assert torch.equal(wc.idx.cpu(), O.weight_indices(w, wlo, whi)[:, :, 0])
    assert torch.equal(wc.idxT.cpu(), wc.idx.cpu().t())
    _, xc = K.actq_fwd(padded(x), K.ACT_NONE, None, K.Q_QUANT, xlo.cuda(), xhi.cuda(), None, want_idx=True)
    assert torch.equal(xc.cpu(), O.act_indices(x, xlo, xhi))
    z = K.qpw_fwd(xc, wc, bias.cuda(), xlo.cuda(), xhi.cuda())
    ref64 = F.conv1d(xq.double(), wq.double(), bias.double())
    ref32 = F.conv1d(xq, wq, bias)
    err_q = (z.cpu().double() - ref64).abs().max().item()
    err_32 = (ref32.double() - ref64).abs().max().item()
    assert err_q <= max(2e-6 * ref64.abs().max().item(), 0.75 * err_32 + 1e-7), (err_q, err_32)   # at least as exact as fp32 ATen
    # backward
    gz = rnd(B, Co, M, seed=11)
    gx = K.qpw_bwd_x(padded(gz), wc)
    gx_ref = torch.einsum("oc,bom->bcm", wq[:, :, 0].double(), gz.double())
    close(gx.cpu().double(), gx_ref, rtol=1e-5, atol=2e-6 * float(gx_ref.abs().max()))
    gw = torch.zeros(Co, Ci, device="cuda")
    K.qpw_bwd_w(padded(gz), xc, xlo.cuda(), xhi.cuda(), gw)
    gw_ref = torch.einsum("bom,bcm->oc", gz.double(), xq.double())
    close(gw.cpu().double(), gw_ref, rtol=1e-5, atol=3e-6 * float(gw_ref.abs().max()))


@pytest.mark.parametrize("B,Ci,Co1,Co2,M", [(2, 64, 32, 48, 77), (2, 512, 128, 128, 501), (1, 128, 512, 16, 260)])
def test_qgemm_pair_matches_single_layers(B, Ci, Co1, Co2, M):
    """the paired q-GEMMs (two layers on one input, concatenated codes) against the two single-layer launches"""
    w1, wlo1, whi1, xlo, xhi, x, b1, _, _ = _q_setup(B, Ci, Co1, M, seed=1)
    w2, wlo2, whi2, _, _, _, b2, _, _ = _q_setup(B, Ci, Co2, M, seed=2)
    cu = lambda t: t.cuda().contiguous()
    wc1, wc2 = K.wq_codes(cu(w1), cu(wlo1), cu(whi1)), K.wq_codes(cu(w2), cu(wlo2), cu(whi2))
    pc = K.WCodes()
    pc.Co, pc.Ci = Co1 + Co2, Ci
    pc.idx = torch.cat([wc1.idx, wc2.idx], 0).contiguous()
    pc.idxT = torch.cat([wc1.idxT, wc2.idxT], 1).contiguous()
    pc.dw, pc.rw = torch.cat([wc1.dw, wc2.dw]), torch.cat([wc1.rw, wc2.rw])
    _, xc = K.actq_fwd(padded(x), K.ACT_NONE, None, K.Q_QUANT, cu(xlo), cu(xhi), None, want_idx=True)
    z1, z2 = K.qpw_fwd2(xc, pc, cu(b1), cu(b2), cu(xlo), cu(xhi), Co1)
    assert torch.equal(z1.cpu(), K.qpw_fwd(xc, wc1, cu(b1), cu(xlo), cu(xhi)).cpu())      # exact integer sums: bit for bit
    assert torch.equal(z2.cpu(), K.qpw_fwd(xc, wc2, cu(b2), cu(xlo), cu(xhi)).cpu())
    g1, g2 = rnd(B, Co1, M, seed=5), rnd(B, Co2, M, seed=6)
    gx = K.qpw_bwd_x2(padded(g1), padded(g2), pc)
    ref = K.qpw_bwd_x(padded(g1), wc1).cpu().double() + K.qpw_bwd_x(padded(g2), wc2).cpu().double()
    close(gx.cpu().double(), ref, rtol=1e-5, atol=2e-6 * float(ref.abs().max()))
    gw = torch.zeros(Co1 + Co2, Ci, device="cuda")
    K.qpw_bwd_w2(padded(g1), padded(g2), xc, cu(xlo), cu(xhi), gw)
    gw1, gw2 = torch.zeros(Co1, Ci, device="cuda"), torch.zeros(Co2, Ci, device="cuda")
    K.qpw_bwd_w(padded(g1), xc, cu(xlo), cu(xhi), gw1)
    K.qpw_bwd_w(padded(g2), xc, cu(xlo), cu(xhi), gw2)
    ref = torch.cat([gw1, gw2], 0).cpu().double()
    close(gw.cpu().double(), ref, rtol=1e-5, atol=3e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("B,Ci,Co1,Co2,M,act", [(2, 64, 32, 0, 77, 1), (2, 512, 128, 128, 501, 0), (1, 128, 512, 0, 260, 1), (2, 32, 64, 32, 100, 0),
                                                (3, 128, 512, 0, 1000, 1), (2, 128, 128, 0, 70, 0), (8, 128, 512, 0, 3999, 1)])   # k_qfwd_k128 (Ci = 128, Co % 128 == 0)
def test_qgemm_fused_output_quantizer(B, Ci, Co1, Co2, M, act):
    """fqss_qpw_fwdq: same z as the plain forward, and output codes bit-equal to the stand-alone quantizer on that z"""
    cu = lambda t: t.cuda().contiguous()
    w1, wlo1, whi1, xlo, xhi, x, b1, _, _ = _q_setup(B, Ci, Co1, M, seed=1)
    wc = K.wq_codes(cu(w1), cu(wlo1), cu(whi1))
    b2 = None
    if Co2:
        w2, wlo2, whi2, _, _, _, b2, _, _ = _q_setup(B, Ci, Co2, M, seed=2)
        wc2 = K.wq_codes(cu(w2), cu(wlo2), cu(whi2))
        pc = K.WCodes()
        pc.Co, pc.Ci = Co1 + Co2, Ci
        pc.idx, pc.idxT = torch.cat([wc.idx, wc2.idx], 0).contiguous(), torch.cat([wc.idxT, wc2.idxT], 1).contiguous()
        pc.dw, pc.rw = torch.cat([wc.dw, wc2.dw]), torch.cat([wc.rw, wc2.rw])
        wc = pc
    _, xc = K.actq_fwd(padded(x), K.ACT_NONE, None, K.Q_QUANT, cu(xlo), cu(xhi), None, want_idx=True)
    slope = torch.tensor([0.2], device="cuda") if act == 1 else None
    r1 = (torch.tensor([-1.1], device="cuda"), torch.tensor([1.7], device="cuda"))
    r2 = (torch.tensor([-0.4], device="cuda"), torch.tensor([2.9], device="cuda")) if Co2 else None
    res = K.qpw_fwdq(xc, wc, cu(b1), cu(b2) if Co2 else None, cu(xlo), cu(xhi), Co1, act, slope, r1, r2)
    zs, ycs = (res[:2], res[2:]) if Co2 else (res[:1], res[1:])
    refz = K.qpw_fwd2(xc, wc, cu(b1), cu(b2), cu(xlo), cu(xhi), Co1) if Co2 else (K.qpw_fwd(xc, wc, cu(b1), cu(xlo), cu(xhi)),)
    for z, yc, zr, r in zip(zs, ycs, refz, (r1, r2)):
        assert torch.equal(z.cpu(), zr.cpu())
        _, ref_idx = K.actq_fwd(zr, act, slope, K.Q_QUANT, r[0], r[1], None, want_idx=True)
        assert torch.equal(yc.cpu()[..., :M], ref_idx.cpu()[..., :M])


@pytest.mark.parametrize("B,Ci,Co1,Co2,M,which", [(2, 512, 128, 128, 501, 3), (2, 64, 32, 32, 77, 1), (1, 128, 64, 96, 260, 2), (3, 32, 32, 64, 1000, 3)])
def test_pair_forward_with_fused_adds(B, Ci, Co1, Co2, M, which):
    """fqss_qpw_fwdq_add: the AddQ behind output 1 (residual add) and / or output 2 (skip sum) evaluated in the pair GEMM's epilogue
    -- z and output codes unchanged, sum codes bit-equal to fqss_ewq_fwd on the same operands (convtasnetq.py:41, 110)"""
    cu = lambda t: t.cuda().contiguous()
    w1, wlo1, whi1, xlo, xhi, x, b1, _, _ = _q_setup(B, Ci, Co1, M, seed=1)
    w2, wlo2, whi2, _, _, _, b2, _, _ = _q_setup(B, Ci, Co2, M, seed=2)
    wc1, wc2 = K.wq_codes(cu(w1), cu(wlo1), cu(whi1)), K.wq_codes(cu(w2), cu(wlo2), cu(whi2))
    pc = K.WCodes()
    pc.Co, pc.Ci = Co1 + Co2, Ci
    pc.idx, pc.idxT = torch.cat([wc1.idx, wc2.idx], 0).contiguous(), torch.cat([wc1.idxT, wc2.idxT], 1).contiguous()
    pc.dw, pc.rw = torch.cat([wc1.dw, wc2.dw]), torch.cat([wc1.rw, wc2.rw])
    _, xc = K.actq_fwd(padded(x), K.ACT_NONE, None, K.Q_QUANT, cu(xlo), cu(xhi), None, want_idx=True)
    dev = torch.device("cuda")
    t = lambda v: torch.tensor([v], device=dev)
    r1, r2 = (t(-1.1), t(1.7)), (t(-0.4), t(2.9))
    gen = torch.Generator().manual_seed(9)
    adds = []
    for i, Co in enumerate((Co1, Co2)):
        if not (which >> i) & 1:
            adds.append(None)
            continue
        ac = K.empty_codes((B, Co, M), dev)
        ac.copy_(torch.randint(0, 256, (B, Co, M), generator=gen, dtype=torch.uint8))
        lo, hi = -2.3 + i, 1.9 + i
        ylo, yhi = (float(v) for v in (r1, r2)[i])
        adds.append((ac, t(lo), t(hi), t(lo + ylo + 0.8), t(hi + yhi - 1.5)))     # the sum saturates at both ends in places
    plain = K.qpw_fwdq(xc, pc, cu(b1), cu(b2), cu(xlo), cu(xhi), Co1, K.ACT_NONE, None, r1, r2)
    res = K.qpw_fwdq(xc, pc, cu(b1), cu(b2), cu(xlo), cu(xhi), Co1, K.ACT_NONE, None, r1, r2, adds=tuple(adds))
    assert len(res) == 5
    for a, b in zip(plain[:2], res[:2]):
        assert torch.equal(a, b)
    for a, b in zip(plain[2:], res[2:4]):
        assert torch.equal(a[..., :M], b[..., :M])
    for ad, sc, yc, r in zip(adds, res[4], res[2:4], (r1, r2)):
        if ad is None:
            assert sc is None
            continue
        _, ref = K.ewq_fwd(ad[0], ad[1], ad[2], yc, r[0], r[1], None, 1.0, K.ACT_NONE, None, ad[3], ad[4], write_out=False)
        assert torch.equal(sc[..., :M], ref[..., :M])
        assert 0 in ref[..., :M].unique().tolist() and 255 in ref[..., :M].unique().tolist()


@pytest.mark.parametrize("B,C,M,n,bottom_prod", [(8, 128, 3999, 5, True), (2, 32, 1000, 3, False), (3, 16, 77, 24, True)])
def test_add_chain_backward(B, C, M, n, bottom_prod):
    """fqss_add_chain_bwd: the backward of n chained AddQ layers in one launch against n launches of fqss_ewq_bwd_p -- producers' gz
    and the bottom gradient bit for bit, range partials and bias gradients to summation order (convtasnetq.py:107-111)"""
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(B * 1000 + n)
    t = lambda v: torch.tensor([v], device=dev)

    def codes():
        c = K.empty_codes((B, C, M), dev)
        c.copy_(torch.randint(0, 256, (B, C, M), generator=gen, dtype=torch.uint8))
        return c
    levels = []
    a = codes()
    ra = (t(-1.3), t(2.1))
    za = padded(torch.randn(B, C, M, generator=gen) * 1.2)
    for l in range(n):
        rb, rq = (t(-0.9 - 0.1 * l), t(1.4 + 0.05 * l)), (t(-1.6 - 0.02 * l), t(2.3 + 0.03 * l))
        levels.append(dict(ac=a, amin=ra[0], amax=ra[1], bc=codes(), bmin=rb[0], bmax=rb[1], qmin=rq[0], qmax=rq[1],
                           z=padded(torch.randn(B, C, M, generator=gen) * 0.9)))
        a, ra = codes(), rq            # (the chain's codes need not be consistent with each other for the backward's arithmetic)
    g = padded(torch.randn(B, C, M, generator=gen))
    new = lambda: (torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev), torch.zeros(C, device=dev))
    # reference: level by level, top first
    ref_out, ref_acc = [None] * n, [None] * n
    gl = g
    pa_ref = None
    for l in range(n - 1, -1, -1):
        lv = levels[l]
        gacc, (pg, pb) = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev), new()
        prod_a = None
        if l == 0 and bottom_prod:
            pa_ref = new()
            prod_a = (za, K.ACT_NONE, None, pa_ref[0], pa_ref[1])
        gz, gza, gzb = K.ewq_bwd_p(lv["ac"], lv["amin"], lv["amax"], lv["bc"], lv["bmin"], lv["bmax"], 1.0, gl, K.ACT_NONE, None, lv["qmin"],
                                   lv["qmax"], gacc, C, prod_a=prod_a, prod_b=(lv["z"], K.ACT_NONE, None, pg, pb))
        ref_out[l], ref_acc[l] = gzb, (gacc, pg, pb)
        gl = gza if prod_a is not None else gz
    ref_bottom = gl
    # the chain
    accs = [(torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev),) + new() for _ in range(n)]
    pa = new() if bottom_prod else None
    assert K.add_chain_ok(B * C, M, C, n)
    outs, bottom = K.add_chain_bwd([dict(ac=lv["ac"], amin=lv["amin"], amax=lv["amax"], bc=lv["bc"], bmin=lv["bmin"], bmax=lv["bmax"], qmin=lv["qmin"],
                                         qmax=lv["qmax"], gacc=acc[0], prod_b=(lv["z"], K.ACT_NONE, None, acc[1], acc[2])) for lv, acc in zip(levels, accs)],
                                   g, prod_a=(za, K.ACT_NONE, None, pa[0], pa[1]) if bottom_prod else None)
    assert torch.equal(bottom[..., :M], ref_bottom[..., :M])
    for l in range(n):
        assert torch.equal(outs[l][..., :M], ref_out[l][..., :M]), l
        for got, want in zip(accs[l], ref_acc[l]):
            close(got.double().cpu(), want.double().cpu(), rtol=1e-6, atol=1e-6 * float(want.abs().max()) + 1e-30)
    if bottom_prod:
        for got, want in zip(pa, pa_ref):
            close(got.double().cpu(), want.double().cpu(), rtol=1e-6, atol=1e-6 * float(want.abs().max()) + 1e-30)
    assert not K.add_chain_ok(9 * C, M, C, n) and not K.add_chain_ok(B * C, M, C, 25) and not K.add_chain_ok(B * C, M, C, 1)


def test_qgemm_exact_integer_maps():
    """A = I-like asymmetric integer codes: catches transposed fragments / wrong tr-read lane maps bit-exactly"""
    B, Ci, Co, M = 1, 64, 96, 160
    wi = ((torch.arange(Co * Ci).reshape(Co, Ci) * 7) % 23 - 11).float() + 20 * torch.eye(Co, Ci)
    c = ((torch.arange(Ci * M).reshape(1, Ci, M) * 13) % 251).float()
    # ranges chosen so that delta_w = 1 (2a/255 = 1) and delta_x = 1, min_x = 0  => z = sum Wi*c exactly
    wlo, whi = torch.full((Co, 1, 1), -127.5), torch.full((Co, 1, 1), 127.5)
    xlo, xhi = torch.tensor([0.0]), torch.tensor([255.0])
    wc = K.wq_codes(wi.reshape(Co, Ci, 1).cuda().contiguous(), wlo.cuda(), whi.cuda())
    assert torch.equal(wc.idx.cpu().float(), wi)
    _, xc = K.actq_fwd(padded(c), K.ACT_NONE, None, K.Q_QUANT, xlo.cuda(), xhi.cuda(), None, want_idx=True)
    assert torch.equal(xc.cpu().float(), c)
    z = K.qpw_fwd(xc, wc, None, xlo.cuda(), xhi.cuda())
    assert torch.equal(z.cpu(), torch.einsum("oc,bcm->bom", wi, c))
    g = ((torch.arange(Co * M).reshape(1, Co, M) * 5) % 17 - 8).float()
    gx = K.qpw_bwd_x(padded(g), wc)
    assert torch.equal(gx.cpu(), torch.einsum("oc,bom->bcm", wi, g))
    gw = torch.zeros(Co, Ci, device="cuda")
    K.qpw_bwd_w(padded(g), xc, xlo.cuda(), xhi.cuda(), gw)
    assert torch.equal(gw.cpu(), torch.einsum("bom,bcm->oc", g, c))


# ---------------------------------------------------------------------------------------------
# codes-only layers (csrc/fused_q.hip)
# ---------------------------------------------------------------------------------------------
def _coded_input(B, C, M, seed):
    gen = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, 256, (B, C, M), generator=gen, dtype=torch.uint8)
    xlo, xhi = torch.tensor([-0.913]), torch.tensor([1.377])
    x = ((xhi - xlo) / 255) * codes.float() + xlo          # exactly what a producer's epilogue stores
    xc = K.empty_codes((B, C, M), "cuda")
    xc.copy_(codes)
    return codes, x, xc, xlo, xhi


def _idx_mismatch(a, b):
    d = (a.int() - b.int()).abs()
    return float((d > 0).float().mean()), int(d.max())


@pytest.mark.parametrize("B,C,M", [(2, 24, 77), (2, 512, 999), (3, 5, 130)])
def test_gnq_coded(B, C, M):
    codes, x, xc, xlo, xhi = _coded_input(B, C, M, seed=C)
    assert torch.equal(K.decode(xc, xlo.cuda(), xhi.cuda()).cpu(), x)
    gm, bt, g = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), rnd(B, C, M, seed=4)
    ylo, yhi = torch.tensor([-2.1]), torch.tensor([2.4])
    xr, gr, br = x.clone().requires_grad_(True), gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    lo_r, hi_r = ylo.clone().requires_grad_(True), yhi.clone().requires_grad_(True)
    z = F.group_norm(xr, 1, gr, br, 1e-8)
    y = O.act_quantize(z, lo_r, hi_r)
    y.backward(g)
    out, yc, mr = K.gnq_fwd(xc, xlo.cuda(), xhi.cuda(), gm.cuda(), bt.cuda(), 1e-8, ylo.cuda(), yhi.cuda(), True)
    frac, dmax = _idx_mismatch(yc.cpu(), O.act_indices(z.detach(), ylo, yhi))
    assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
    delta = (yhi - ylo) / 255
    assert torch.equal(out.cpu(), delta * yc.cpu().float() + ylo)         # fp32 copy == decode(codes)
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
    gg, gb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gx = K.gnq_bwd(xc, xlo.cuda(), xhi.cuda(), padded(g), gm.cuda(), bt.cuda(), mr, ylo.cuda(), yhi.cuda(), gacc, gg, gb)
    tol = 2e-3 + 20 * frac
    bad = (gx.cpu() - xr.grad).abs() > 1e-4 + 1e-3 * xr.grad.abs()
    assert bad.float().mean() <= tol, bad.float().mean()
    close(gg, gr.grad, rtol=5e-3, atol=5e-3 * float(gr.grad.abs().max()) + 1e-3)
    close(gb, br.grad, rtol=5e-3, atol=5e-3 * float(br.grad.abs().max()) + 1e-3)
    ga = gacc.view(-1, 3).sum(0).cpu().numpy()
    sc = float(g.abs().sum()) * 2e-5 + 1e-4
    np.testing.assert_allclose(ga[0], lo_r.grad.item(), rtol=5e-3, atol=sc)
    np.testing.assert_allclose(ga[1], hi_r.grad.item(), rtol=5e-3, atol=sc)
    # fqss_gnq_bwd_p: the same backward that also runs the PRODUCER's epilogue backward (a conv + PReLU + fake-quant whose
    # output codes are xc) == fqss_gnq_bwd followed by fqss_actq_bwd on the producer's pre-quant z
    cu = lambda t: t.cuda()
    slope = torch.tensor([0.3], device="cuda")
    xv = x.cuda()
    pz = padded(torch.where(xv > 0, xv, xv / 0.3) + 0.3 * float(delta) * rnd(B, C, M, seed=9).cuda())   # a z that quantizes to ~xc
    pg_ref, pb_ref = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda"), torch.zeros(C, device="cuda")
    gz_ref = K.actq_bwd(pz, gx, K.ACT_PRELU, slope, K.Q_QUANT, cu(xlo), cu(xhi), pg_ref, gbias=pb_ref, C=C)
    gacc2, pg, pb = torch.zeros_like(gacc), torch.zeros_like(pg_ref), torch.zeros(C, device="cuda")
    gg2, gb2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    gz = K.gnq_bwd(xc, cu(xlo), cu(xhi), padded(g), cu(gm), cu(bt), mr, cu(ylo), cu(yhi), gacc2, gg2, gb2,
                   producer=(pz, K.ACT_PRELU, slope, pg, pb))
    assert torch.equal(gz.cpu(), gz_ref.cpu())                       # same arithmetic, element for element
    close(pb, pb_ref, rtol=1e-4, atol=1e-5 * float(pb_ref.abs().max()) + 1e-6)
    np.testing.assert_allclose(pg.view(-1, 3).sum(0).cpu().numpy(), pg_ref.view(-1, 3).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3 * sc)
    assert torch.equal(gg2.cpu(), gg.cpu()) and torch.equal(gacc2.view(-1, 3).sum(0).cpu(), gacc.view(-1, 3).sum(0).cpu())


@pytest.mark.parametrize("B,C,M,dil", [(2, 32, 77, 1), (2, 32, 77, 4), (1, 512, 999, 128), (3, 7, 130, 2)])
def test_dwq_coded(B, C, M, dil):
    codes, x, xc, xlo, xhi = _coded_input(B, C, M, seed=C + dil)
    w, bias, g = rnd(C, 1, 3, seed=2, scale=0.5), rnd(C, seed=3, scale=0.1), rnd(B, C, M, seed=4)
    slope = torch.tensor([0.25])
    ylo, yhi = torch.tensor([-1.3]), torch.tensor([2.2])
    xr, wr, br, sr = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True), slope.clone().requires_grad_(True)
    lo_r, hi_r = ylo.clone().requires_grad_(True), yhi.clone().requires_grad_(True)
    z = F.prelu(F.conv1d(xr, wr, br, padding=dil, dilation=dil, groups=C), sr)
    y = O.act_quantize(z, lo_r, hi_r)
    y.backward(g)
    cu = lambda t: t.cuda()
    out, yc = K.dwq_fwd(xc, cu(xlo), cu(xhi), cu(w), cu(bias), dil, dil, K.ACT_PRELU, cu(slope), cu(ylo), cu(yhi), True)
    frac, dmax = _idx_mismatch(yc.cpu(), O.act_indices(z.detach(), ylo, yhi))
    assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
    assert torch.equal(out.cpu(), ((yhi - ylo) / 255) * yc.cpu().float() + ylo)
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
    gb = torch.zeros(C, device="cuda")
    gz = K.dwq_bwd_z(xc, cu(xlo), cu(xhi), cu(w), cu(bias), padded(g), dil, dil, K.ACT_PRELU, cu(slope), cu(ylo), cu(yhi), gacc, gb)
    gx = K.dwconv_bwd_x(gz, cu(w), dil, dil)
    gw = torch.zeros(C, 1, 3, device="cuda")
    K.dwq_bwd_w(gz, xc, cu(xlo), cu(xhi), gw, dil, dil)
    bad = (gx.cpu() - xr.grad).abs() > 1e-4 + 1e-3 * xr.grad.abs()
    assert bad.float().mean() <= 2e-3 + 20 * frac
    close(gw, wr.grad, rtol=5e-3, atol=5e-3 * float(wr.grad.abs().max()) + 1e-3)
    close(gb, br.grad, rtol=5e-3, atol=5e-3 * float(br.grad.abs().max()) + 1e-3)
    ga = gacc.view(-1, 3).sum(0).cpu().numpy()
    sc = float(g.abs().sum()) * 2e-5 + 1e-4
    np.testing.assert_allclose(ga[0], lo_r.grad.item(), rtol=5e-3, atol=sc)
    np.testing.assert_allclose(ga[1], hi_r.grad.item(), rtol=5e-3, atol=sc)
    np.testing.assert_allclose(ga[2], sr.grad.item(), rtol=5e-3, atol=sc)
    # the single-launch backward (gz kept in LDS) agrees with the three-kernel chain: gx bit for bit, sums to fp32 noise
    gacc1 = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
    gb1, gw1 = torch.zeros(C, device="cuda"), torch.zeros(C, 1, 3, device="cuda")
    gx1 = K.dwq_bwd(xc, cu(xlo), cu(xhi), cu(w), cu(bias), padded(g), dil, dil, K.ACT_PRELU, cu(slope), cu(ylo), cu(yhi), gacc1, gb1, gw1)
    assert torch.equal(gx1.cpu(), gx.cpu())
    close(gw1, gw, rtol=1e-4, atol=1e-5 * float(gw.abs().max()) + 1e-6)
    close(gb1, gb, rtol=1e-4, atol=1e-5 * float(gb.abs().max()) + 1e-6)
    np.testing.assert_allclose(gacc1.view(-1, 3).sum(0).cpu().numpy(), ga, rtol=1e-4, atol=1e-3 * sc)   # fp32 per-thread partials, other order
    assert K.dwq_bwd(xc, cu(xlo), cu(xhi), cu(w), cu(bias), padded(g), dil, dil, K.ACT_PRELU, cu(slope), cu(ylo), cu(yhi), gacc1, gb1, None,
                     want_gx=False) is None


# ---------------------------------------------------------------------------------------------
# degenerate inputs and argument validation through the C ABI (the library never faults on them)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,M,mode", [(2, 16, 77, "add_codes"), (2, 128, 999, "add_codes"), (3, 5, 130, "sub_codes"), (2, 16, 77, "add_f32"),
                                        (2, 128, 501, "prelu"), (1, 7, 64, "prelu")])
def test_ewq_coded(B, C, M, mode):
    """fqss_ewq_fwd / fqss_ewq_bwd / fqss_ewq_bwd_p (AddQ / SubQ / NlQ on codes: qat_layers.py:62-84, 511-518 of the reference)
    against the oracle quantizer on z = act(dec(a) + sb * b): output codes, gx of both operands, range / slope gradients; the
    _p form (the producers' output-quantizer backward fused in) against ewq_bwd followed by actq_bwd on each producer."""
    codes, xa, ac, alo, ahi = _coded_input(B, C, M, seed=C + M)
    cu = lambda t: t.cuda()
    gen = torch.Generator().manual_seed(7)
    bcodes = torch.randint(0, 256, (B, C, M), generator=gen, dtype=torch.uint8)
    blo, bhi = torch.tensor([-1.21]), torch.tensor([0.77])
    xb = ((bhi - blo) / 255) * bcodes.float() + blo
    bc = K.empty_codes((B, C, M), "cuda")
    bc.copy_(bcodes)
    sb = {"add_codes": 1.0, "sub_codes": -1.0, "add_f32": 1.0, "prelu": 0.0}[mode]
    act = K.ACT_PRELU if mode == "prelu" else K.ACT_NONE
    slope = torch.tensor([0.25])
    ylo, yhi = torch.tensor([-1.9]), torch.tensor([2.3])
    g = rnd(B, C, M, seed=4)
    ar, br, sr = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True), slope.clone().requires_grad_(True)
    lo_r, hi_r = ylo.clone().requires_grad_(True), yhi.clone().requires_grad_(True)
    z = F.prelu(ar, sr) if mode == "prelu" else ar + sb * br
    y = O.act_quantize(z, lo_r, hi_r)
    y.backward(g)
    has_b = mode != "prelu"
    coded_b = mode in ("add_codes", "sub_codes")
    bf = padded(xb) if (has_b and not coded_b) else None
    args_b = (bc, cu(blo), cu(bhi)) if coded_b else (None, None, None)
    sl = cu(slope) if mode == "prelu" else None
    for write_out in (True, False):
        out, yc = K.ewq_fwd(ac, cu(alo), cu(ahi), *args_b, bf, sb, act, sl, cu(ylo), cu(yhi), write_out)
        frac, dmax = _idx_mismatch(yc.cpu(), O.act_indices(z.detach(), ylo, yhi))
        assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
        if write_out:
            assert torch.equal(out.cpu(), ((yhi - ylo) / 255) * yc.cpu().float() + ylo)      # fp32 copy == decode(codes)
    gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
    gz = K.ewq_bwd(ac, cu(alo), cu(ahi), *args_b, bf, sb, padded(g), act, sl, cu(ylo), cu(yhi), gacc)
    tol = 2e-3 + 20 * frac
    bad = (gz.cpu() - ar.grad).abs() > 1e-4 + 1e-3 * ar.grad.abs()
    assert bad.float().mean() <= tol, bad.float().mean()
    if has_b:
        bad = (sb * gz.cpu() - br.grad).abs() > 1e-4 + 1e-3 * br.grad.abs()
        assert bad.float().mean() <= tol
    ga = gacc.view(-1, 3).sum(0).cpu().numpy()
    sc = float(g.abs().sum()) * 2e-5 + 1e-4
    np.testing.assert_allclose(ga[0], lo_r.grad.item(), rtol=5e-3, atol=sc)
    np.testing.assert_allclose(ga[1], hi_r.grad.item(), rtol=5e-3, atol=sc)
    if mode == "prelu":
        np.testing.assert_allclose(ga[2], sr.grad.item(), rtol=5e-3, atol=sc)
    if mode != "add_codes":
        return
    # ---- fqss_ewq_bwd_p: a and b are fresh outputs of two pointwise convs (res | skip of a TCN block); their pre-quant z's --------
    da, db = float((ahi - alo) / 255), float((bhi - blo) / 255)
    pza = padded(xa + 0.3 * da * rnd(B, C, M, seed=9))           # a z that quantizes to ~ac (some elements land in the next bin)
    pzb = padded(xb + 0.3 * db * rnd(B, C, M, seed=10))
    ref = {}
    for which, pz, lo, hi in (("a", pza, alo, ahi), ("b", pzb, blo, bhi)):
        pg, pb = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda"), torch.zeros(C, device="cuda")
        ref[which] = (K.actq_bwd(pz, gz, K.ACT_NONE, None, K.Q_QUANT, cu(lo), cu(hi), pg, gbias=pb, C=C), pg, pb)
    for use_a, use_b in ((True, True), (True, False), (False, True)):
        gacc2 = torch.zeros_like(gacc)
        pga, pgb = torch.zeros_like(gacc), torch.zeros_like(gacc)
        pba, pbb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        gz2, gza, gzb = K.ewq_bwd_p(ac, cu(alo), cu(ahi), bc, cu(blo), cu(bhi), sb, padded(g), act, None, cu(ylo), cu(yhi), gacc2, C,
                                    prod_a=(pza, K.ACT_NONE, None, pga, pba) if use_a else None,
                                    prod_b=(pzb, K.ACT_NONE, None, pgb, pbb) if use_b else None)
        assert (gz2 is None) == (use_a and use_b)
        if gz2 is not None:
            assert torch.equal(gz2.cpu(), gz.cpu())
        for used, got, which, pg, pb in ((use_a, gza, "a", pga, pba), (use_b, gzb, "b", pgb, pbb)):
            assert (got is not None) == used
            if used:
                rz, rg, rb = ref[which]
                assert torch.equal(got.cpu(), rz.cpu())                       # same arithmetic, element for element
                close(pb, rb, rtol=1e-4, atol=1e-5 * float(rb.abs().max()) + 1e-6)
                np.testing.assert_allclose(pg.view(-1, 3).sum(0).cpu().numpy(), rg.view(-1, 3).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3 * sc)
        np.testing.assert_allclose(gacc2.view(-1, 3).sum(0).cpu().numpy(), ga, rtol=1e-6, atol=1e-6 * sc)


@pytest.mark.parametrize("B,S,C,M", [(2, 2, 16, 77), (2, 2, 64, 3999), (1, 3, 8, 130), (3, 1, 5, 64), (1, 4, 12, 4100)])
def test_mulq_coded(B, S, C, M):
    """fqss_mulq_fwd / fqss_mulq_bwd (MulQ of ConvTasNetQ.forward on codes: qat_layers.py:134-153, convtasnetq.py:277 of the reference):
    codes bit-identical to the chain decode -> fqss_mul_bcast_fwd -> fqss_actq_fwd and within the oracle quantizer's bins; the
    backward element for element what fqss_actq_bwd + fqss_mul_bcast_bwd return on the materialised product; the fused-producer form
    (the mask conv's ReLU + output quantizer backward in the same launch) against fqss_actq_bwd on its result."""
    cu = lambda t: t.cuda()
    gen = torch.Generator().manual_seed(B + 10 * S + C + M)
    mcodes = torch.randint(0, 256, (B, S, C, M), generator=gen, dtype=torch.uint8)
    fcodes = torch.randint(0, 256, (B, C, M), generator=gen, dtype=torch.uint8)
    mlo, mhi = torch.tensor([0.0]), torch.tensor([1.37])            # a ReLU mask
    flo, fhi = torch.tensor([-1.21]), torch.tensor([0.77])
    ylo, yhi = torch.tensor([-0.9]), torch.tensor([0.6])
    xm = ((mhi - mlo) / 255) * mcodes.float() + mlo
    xf = ((fhi - flo) / 255) * fcodes.float() + flo
    mc, fc = K.empty_codes((B, S, C, M), "cuda"), K.empty_codes((B, C, M), "cuda")
    mc.copy_(mcodes)
    fc.copy_(fcodes)
    g = rnd(B, S, C, M, seed=4)
    mr, fr = xm.clone().requires_grad_(True), xf.clone().requires_grad_(True)
    lo_r, hi_r = ylo.clone().requires_grad_(True), yhi.clone().requires_grad_(True)
    z = mr * fr.unsqueeze(1)
    O.act_quantize(z, lo_r, hi_r).backward(g)
    # the un-fused chain on the GPU
    xm_g, xf_g = K.decode(mc, cu(mlo), cu(mhi)), K.decode(fc, cu(flo), cu(fhi))
    assert torch.equal(xm_g.cpu(), xm) and torch.equal(xf_g.cpu(), xf)
    z_g = K.mul_bcast_fwd(xm_g, xf_g)
    _, yc_ref = K.actq_fwd(z_g, K.ACT_NONE, None, K.Q_QUANT, cu(ylo), cu(yhi), None, want_idx=True)
    for write_out in (True, False):
        out, yc = K.mulq_fwd(mc, cu(mlo), cu(mhi), fc, cu(flo), cu(fhi), cu(ylo), cu(yhi), write_out)
        assert torch.equal(yc.cpu(), yc_ref.cpu().reshape(B, S, C, M))
        frac, dmax = _idx_mismatch(yc.cpu(), O.act_indices(z.detach(), ylo, yhi))
        assert dmax <= 1 and frac <= 2e-3, (frac, dmax)
        if write_out:
            assert torch.equal(out.cpu(), ((yhi - ylo) / 255) * yc.cpu().float() + ylo)
    gacc_ref = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device="cuda")
    gz_ref = K.actq_bwd(z_g, padded(g), K.ACT_NONE, None, K.Q_QUANT, cu(ylo), cu(yhi), gacc_ref)
    gm_ref, gf_ref = K.mul_bcast_bwd(gz_ref, xm_g, xf_g)
    gacc = torch.zeros_like(gacc_ref)
    gm, gf = K.mulq_bwd(mc, cu(mlo), cu(mhi), fc, cu(flo), cu(fhi), padded(g), cu(ylo), cu(yhi), gacc)
    assert torch.equal(gm.cpu(), gm_ref.cpu()) and torch.equal(gf.cpu(), gf_ref.cpu())
    sc = float(g.abs().sum()) * 2e-5 + 1e-4
    ga = gacc.view(-1, 3).sum(0).cpu().numpy()
    np.testing.assert_allclose(ga, gacc_ref.view(-1, 3).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3 * sc)
    np.testing.assert_allclose(ga[0], lo_r.grad.item(), rtol=5e-3, atol=sc)
    np.testing.assert_allclose(ga[1], hi_r.grad.item(), rtol=5e-3, atol=sc)
    tol = 2e-3 + 20 * frac
    assert ((gm.cpu() - mr.grad).abs() > 1e-4 + 1e-3 * mr.grad.abs()).float().mean() <= tol
    assert ((gf.cpu() - fr.grad).abs() > 2e-4 + 2e-3 * fr.grad.abs()).float().mean() <= S * tol
    gm2, gf2 = K.mulq_bwd(mc, cu(mlo), cu(mhi), fc, cu(flo), cu(fhi), padded(g), cu(ylo), cu(yhi), torch.zeros_like(gacc), want_gfeat=False)
    assert gf2 is None and torch.equal(gm2.cpu(), gm.cpu())
    # ---- the mask as the fresh output of a pointwise conv + ReLU + quantizer: that layer's epilogue backward in the same launch
    dm = float((mhi - mlo) / 255)
    pz = padded((xm + 0.3 * dm * rnd(B, S, C, M, seed=9) - 0.2 * (mcodes == 0).float()).reshape(B, S * C, M))   # code 0 <- negative pre-activations
    pg_ref, pb_ref = torch.zeros_like(gacc), torch.zeros(S * C, device="cuda")
    gzp_ref = K.actq_bwd(pz, gm.reshape(B, S * C, M), K.ACT_RELU, None, K.Q_QUANT, cu(mlo), cu(mhi), pg_ref, gbias=pb_ref, C=S * C)
    pg, pb, gacc3 = torch.zeros_like(gacc), torch.zeros(S * C, device="cuda"), torch.zeros_like(gacc)
    gzp, gf3 = K.mulq_bwd(mc, cu(mlo), cu(mhi), fc, cu(flo), cu(fhi), padded(g), cu(ylo), cu(yhi), gacc3,
                          prod=(pz, K.ACT_RELU, None, pg, pb))
    assert torch.equal(gzp.cpu().reshape(B, S * C, M), gzp_ref.cpu()) and torch.equal(gf3.cpu(), gf.cpu())
    close(pb, pb_ref, rtol=1e-4, atol=1e-5 * float(pb_ref.abs().max()) + 1e-6)
    np.testing.assert_allclose(pg.view(-1, 3).sum(0).cpu().numpy(), pg_ref.view(-1, 3).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3 * sc)
    np.testing.assert_allclose(gacc3.view(-1, 3).sum(0).cpu().numpy(), ga, rtol=1e-6, atol=1e-6 * sc)


@pytest.mark.parametrize("B,C,M", [(2, 32, 77), (2, 512, 999), (3, 128, 4100), (1, 24, 130)])
def test_code_statistics_from_the_producing_kernels(B, C, M):
    """SURVEY K7 ("stats can be produced by the previous kernel's epilogue"): the q-GEMM / depthwise forward emit the exact integer
    (sum c, sum c^2) of their output codes as per-workgroup slots; fqss_gnq_fwd fed with them returns bit for bit what it
    returns after its own statistics pass (qat_layers.py:445-448 on the codes of qat_quant.py:136-147)."""
    cu = lambda t: t.cuda().contiguous()
    gm, bt = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ylo, yhi = torch.tensor([-2.1]), torch.tensor([2.4])
    # ---- depthwise producer
    codes, x, xc, xlo, xhi = _coded_input(B, C, M, seed=C + 1)
    w, bias, slope = rnd(C, 1, 3, seed=2, scale=0.5), rnd(C, seed=3, scale=0.1), torch.tensor([0.25])
    dlo, dhi = torch.tensor([-1.3]), torch.tensor([2.2])
    st = K.new_stats("dwq", B, C, M, "cuda")
    assert st is not None and st.nslots == C * ((M + 4095) // 4096)
    _, yc = K.dwq_fwd(xc, cu(xlo), cu(xhi), cu(w), cu(bias), 2, 2, K.ACT_PRELU, cu(slope), cu(dlo), cu(dhi), False, stats=st)
    _, yc0 = K.dwq_fwd(xc, cu(xlo), cu(xhi), cu(w), cu(bias), 2, 2, K.ACT_PRELU, cu(slope), cu(dlo), cu(dhi), False)
    assert torch.equal(yc.cpu(), yc0.cpu())
    cc = yc.cpu().long()
    sums = st.ws.view(B, st.nslots, 2).sum(1).cpu()
    assert torch.equal(sums[:, 0], cc.sum((1, 2))) and torch.equal(sums[:, 1], (cc * cc).sum((1, 2)))
    a = K.gnq_fwd(yc, cu(dlo), cu(dhi), cu(gm), cu(bt), 1e-8, cu(ylo), cu(yhi), True, stats=st)
    b_ = K.gnq_fwd(yc, cu(dlo), cu(dhi), cu(gm), cu(bt), 1e-8, cu(ylo), cu(yhi), True)
    for u, v in zip(a, b_):
        assert torch.equal(u.cpu()[..., :M] if u.dim() == 3 else u.cpu(), v.cpu()[..., :M] if v.dim() == 3 else v.cpu())
    # ---- q-GEMM producer (C input channels -> 2C outputs, PReLU + fake-quant fused)
    if C % 16 == 0:
        Co = 2 * C
        wq, wlo, whi, xlo2, xhi2, x2, b2, _, _ = _q_setup(B, C, Co, M, seed=C + 5)
        wc = K.wq_codes(cu(wq), cu(wlo), cu(whi))
        _, xc2 = K.actq_fwd(padded(x2), K.ACT_NONE, None, K.Q_QUANT, cu(xlo2), cu(xhi2), None, want_idx=True)
        st = K.new_stats("qpw", B, Co, M, "cuda")
        assert st is not None and st.nslots == ((Co + 127) // 128) * ((M + 63) // 64)
        sl = torch.tensor([0.2], device="cuda")
        r1 = (torch.tensor([-1.1], device="cuda"), torch.tensor([1.7], device="cuda"))
        z, yc = K.qpw_fwdq(xc2, wc, cu(b2), None, cu(xlo2), cu(xhi2), Co, K.ACT_PRELU, sl, r1, stats=st)
        z0, yc0 = K.qpw_fwdq(xc2, wc, cu(b2), None, cu(xlo2), cu(xhi2), Co, K.ACT_PRELU, sl, r1)
        assert torch.equal(yc.cpu()[..., :M], yc0.cpu()[..., :M]) and torch.equal(z.cpu(), z0.cpu())
        cc = yc.cpu()[..., :M].long()
        sums = st.ws.view(B, st.nslots, 2).sum(1).cpu()
        assert torch.equal(sums[:, 0], cc.sum((1, 2))) and torch.equal(sums[:, 1], (cc * cc).sum((1, 2)))
        gm2, bt2 = 1 + 0.1 * rnd(Co, seed=4), 0.1 * rnd(Co, seed=5)
        a = K.gnq_fwd(yc, r1[0], r1[1], cu(gm2), cu(bt2), 1e-8, cu(ylo), cu(yhi), True, stats=st)
        b_ = K.gnq_fwd(yc, r1[0], r1[1], cu(gm2), cu(bt2), 1e-8, cu(ylo), cu(yhi), True)
        for u, v in zip(a, b_):
            assert torch.equal(u.cpu()[..., :M] if u.dim() == 3 else u.cpu(), v.cpu()[..., :M] if v.dim() == 3 else v.cpu())
    assert K.new_stats("dwq", 1, 2048, 4000, "cuda") is None          # too many slots: the gLN computes its own statistics


def test_empty_batches_are_noops_and_bad_arguments_are_refused():
    from fqss_amd import _lib
    dev = "cuda"
    lo, hi = torch.tensor([-1.0], device=dev), torch.tensor([1.0], device=dev)
    C, M = 16, 40
    # B = 0: every launcher returns OK without touching memory
    x0 = K.empty_act((0, C, M), dev)
    assert K.actq_fwd(x0, K.ACT_NONE, None, K.Q_QUANT, lo, hi, None).shape[0] == 0
    xc0 = K.empty_codes((0, C, M), dev)
    w = torch.randn(32, C, 1, device=dev) * 0.1
    wc = K.wq_codes(w, -torch.ones(32, 1, 1, device=dev), torch.ones(32, 1, 1, device=dev))
    assert K.qpw_fwd(xc0, wc, None, lo, hi).shape == (0, 32, M)
    gw = torch.zeros(32, C, device=dev)
    K.qpw_bwd_w(K.empty_act((0, 32, M), dev), xc0, lo, hi, gw)
    assert float(gw.abs().max()) == 0.0
    assert K.qpw_bwd_x(K.empty_act((0, 32, M), dev), wc).shape == (0, C, M)
    # refused, with a message, instead of a fault: unaligned code rows, Ci not a multiple of 16, null ranges
    xc = K.empty_codes((1, C, M), dev)
    z = K.empty_act((1, 32, M), dev)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    with pytest.raises(_lib.FqssError, match="aligned"):      # ld_xc = 41: rows not 16-B aligned
        _lib.call("fqss_qpw_fwd", P(xc), P(wc.idx), P(wc.dw), P(wc.rw), None, P(lo), P(hi), P(z), 1, C, 32, M, M + 1, 48, st)
    with pytest.raises(_lib.FqssError):                        # Ci = 24
        _lib.call("fqss_qpw_fwd", P(xc), P(wc.idx), P(wc.dw), P(wc.rw), None, P(lo), P(hi), P(z), 1, 24, 32, M, 48, 48, st)
    with pytest.raises(_lib.FqssError, match="null"):
        _lib.call("fqss_qpw_fwd", P(xc), P(wc.idx), P(wc.dw), P(wc.rw), None, None, P(hi), P(z), 1, C, 32, M, 48, 48, st)
    torch.cuda.synchronize()


# ---------------------------------------------------------------- general convolution geometry (row a15): frame gather / overlap-add
FRAME_CASES = [  # B, C, H, W, k, s, p, d
    (2, 6, 1, 53, (1, 8), (1, 4), (0, 2), (1, 1)),      # time-branch encoder conv k8 s4 p2
    (2, 6, 1, 53, (1, 3), (1, 1), (0, 2), (1, 2)),      # DConv dilated k3
    (2, 4, 22, 7, (8, 1), (4, 1), (2, 0), (1, 1)),      # frequency-branch encoder
    (2, 4, 22, 7, (3, 3), (1, 1), (1, 1), (1, 1)),      # decoder rewrite
    (1, 3, 9, 11, (3, 2), (2, 3), (1, 2), (2, 1)),      # everything at once
    (3, 5, 1, 16, (1, 1), (1, 1), (0, 0), (1, 1)),
]


@pytest.mark.parametrize("B,C,H,W,k,s,p,d", FRAME_CASES)
def test_frames_gather_and_ola(B, C, H, W, k, s, p, d):
    geom = K.ConvGeom(k, s, p, d)
    x = rnd(B, C, H, W, seed=5)
    want = F.unfold(x, k, dilation=d, padding=p, stride=s)                     # [B, C*kh*kw, L]
    for conv in (lambda t: t.cuda(), lambda t: K.empty_act((B, C, H, W), "cuda").copy_(t), lambda t: K.empty_sig((B, C, H, W), "cuda").copy_(t)):
        f, Ho, Wo = K.frames_gather(conv(x), geom)
        assert (Ho, Wo) == geom.out_hw(H, W) and tuple(f.shape) == tuple(want.shape)
        assert torch.equal(f.cpu(), want)                                    # a gather: bit-exact
    fr = rnd(*want.shape, seed=6)
    bias = rnd(C, seed=7)
    y = K.frames_ola(padded(fr), bias.cuda(), (B, C, H, W), geom)
    close(y, F.fold(fr, (H, W), k, dilation=d, padding=p, stride=s) + bias[None, :, None, None], rtol=1e-6, atol=1e-6)
    y0 = K.frames_ola(fr.cuda(), None, (B, C, H, W), geom)
    close(y0, F.fold(fr, (H, W), k, dilation=d, padding=p, stride=s), rtol=1e-6, atol=1e-6)
    out = torch.zeros(C, device="cuda")
    K.chan_sum(padded(x.reshape(B, C, H * W)), out)
    close(out, x.sum((0, 2, 3)), rtol=1e-5, atol=1e-5)


def test_frames_geometry_is_checked():
    from fqss_amd._lib import FqssError
    with pytest.raises((ValueError, FqssError)):
        K.frames_gather(torch.zeros(1, 2, 1, 3, device="cuda"), K.ConvGeom((1, 8), (1, 4)))


# ---------------------------------------------------------------- streaming attention core (row a15): long sequences, cross attention
@pytest.mark.parametrize("Lq,Lk,B,nh,hd,bf", [(300, 517, 2, 2, 48, True), (257, 257, 1, 3, 64, False), (33, 70, 2, 4, 4, True), (700, 64, 1, 2, 16, False),
                                              (1, 1, 1, 1, 64, True), (5, 37, 2, 2, 32, True), (40, 3, 1, 2, 32, False), (129, 95, 2, 1, 64, True),
                                              (250, 250, 3, 4, 16, False), (70, 33, 2, 2, 16, True)])
def test_attn_long(Lq, Lk, B, nh, hd, bf):
    E = nh * hd
    shp = lambda L: (B, L, E) if bf else (L, B, E)
    q, k, v = rnd(*shp(Lq), seed=1, scale=0.4), rnd(*shp(Lk), seed=2, scale=0.9), rnd(*shp(Lk), seed=3)
    go = rnd(*shp(Lq), seed=4)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))

    def heads(t, L):
        t = t if bf else t.transpose(0, 1)                       # -> [B, L, E]
        return t.reshape(B, L, nh, hd).permute(0, 2, 1, 3)      # [B, nh, L, hd]
    s = heads(qr, Lq) @ heads(kr, Lk).transpose(-1, -2)
    p = torch.softmax(s, -1)
    o = (p @ heads(vr, Lk)).permute(0, 2, 1, 3).reshape(B, Lq, E)
    o = o if bf else o.transpose(0, 1)
    o.backward(go)
    # operands as column blocks of a wider projection (the layout MhaCoreX passes)
    wide = lambda t: torch.cat([t, t, t], -1).cuda()
    Xq, Xk = wide(q), wide(k)
    Xk[..., 2 * E:] = v.cuda()
    obs_a, obs_s = (torch.tensor([-1, 0], dtype=torch.int32, device="cuda") for _ in range(2))
    out, stats = K.attn_long_fwd(Xq[..., :E], Xk[..., E:2 * E], Xk[..., 2 * E:], nh, bf, obs_a, obs_s)
    close(out, o, rtol=2e-5, atol=2e-5)
    ws = torch.zeros(2, device="cuda")
    K.observer_ema(ws[:1], ws[1:], obs_a, 0.0)                  # alpha 0: (min, max) of this call
    close(ws, torch.stack([s.min(), s.max()]), rtol=1e-5, atol=1e-5)
    K.observer_ema(ws[:1], ws[1:], obs_s, 0.0)
    close(ws, torch.stack([p.min(), p.max()]), rtol=1e-4, atol=1e-7)
    gq, gk, gv = K.attn_long_bwd(Xq[..., :E], Xk[..., E:2 * E], Xk[..., 2 * E:], out, go.cuda(), stats, nh, bf)
    close(gq, qr.grad, rtol=1e-4, atol=2e-5)
    close(gk, kr.grad, rtol=1e-4, atol=2e-5)
    close(gv, vr.grad, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("Lq,Lk,B,nh,hd,bf", [(250, 250, 3, 4, 16, False), (300, 517, 2, 2, 64, True), (129, 95, 2, 3, 32, False), (5, 37, 2, 2, 32, True),
                                              (1, 1, 1, 1, 64, True), (33, 70, 1, 2, 16, True)])
def test_attn_long_coded(Lq, Lk, B, nh, hd, bf):
    """the attention core from the 8-bit CODES of q, k, v (fqss_attn_long_fwd_c / _bwd_c) against torch on the de-quantized values:
    the output, and the gradients with respect to those values"""
    E = nh * hd
    shp = lambda L: (B, L, E) if bf else (L, B, E)
    g = torch.Generator().manual_seed(7)
    qc, kc, vc = (torch.randint(0, 256, shp(L), generator=g, dtype=torch.uint8) for L in (Lq, Lk, Lk))
    rng = [(-0.61, 0.73), (-1.3, 0.9), (-0.8, 1.7)]
    deq = lambda c, lo, hi: (torch.tensor((hi - lo), dtype=torch.float32) / 255.0) * c.float() + lo
    q, k, v = (deq(c, *r) for c, r in zip((qc, kc, vc), rng))
    go = rnd(*shp(Lq), seed=4)
    qr, kr, vr = (t.clone().double().requires_grad_(True) for t in (q, k, v))

    def heads(t, L):
        t = t if bf else t.transpose(0, 1)
        return t.reshape(B, L, nh, hd).permute(0, 2, 1, 3)
    p = torch.softmax(heads(qr, Lq) @ heads(kr, Lk).transpose(-1, -2), -1)
    o = (p @ heads(vr, Lk)).permute(0, 2, 1, 3).reshape(B, Lq, E)
    o = o if bf else o.transpose(0, 1)
    o.backward(go.double())
    ranges = [(torch.tensor([lo], device="cuda"), torch.tensor([hi], device="cuda")) for lo, hi in rng]
    out, stats = K.attn_long_fwd_c(qc.cuda(), kc.cuda(), vc.cuda(), ranges, nh, bf)
    close(out, o.float(), rtol=2e-5, atol=2e-5)
    gq, gk, gv = K.attn_long_bwd_c(qc.cuda(), kc.cuda(), vc.cuda(), ranges, out, go.cuda(), stats, nh, bf)
    close(gq, qr.grad.float(), rtol=1e-4, atol=2e-5)
    close(gk, kr.grad.float(), rtol=1e-4, atol=2e-5)
    close(gv, vr.grad.float(), rtol=1e-4, atol=2e-5)


# ---------------------------------------------------------------- HTDemucs small ops and the spectrogram pair (row a15)
def test_chan_and_col_scale():
    x, s, g = rnd(3, 5, 77, seed=1), rnd(5, seed=2), rnd(3, 5, 77, seed=3)
    close(K.chan_op(padded(x), s.cuda(), 0), x * s[None, :, None], rtol=0, atol=0)
    close(K.chan_op(padded(x), s.cuda(), 1), x + s[None, :, None], rtol=0, atol=0)
    gs = torch.zeros(5, device="cuda")
    gx = K.chan_scale_bwd(padded(g), padded(x), s.cuda(), gs)
    close(gx, g * s[None, :, None], rtol=0, atol=0)
    close(gs, (g * x).sum((0, 2)), rtol=1e-5, atol=1e-5)
    xr, sr, gr = rnd(7, 9, 24, seed=4), rnd(24, seed=5), rnd(7, 9, 24, seed=6)
    close(K.col_scale_fwd(xr.cuda(), sr.cuda()), xr * sr, rtol=0, atol=0)
    gs = torch.zeros(24, device="cuda")
    gx = K.col_scale_bwd(gr.cuda(), xr.cuda(), sr.cuda(), gs)
    close(gx, gr * sr, rtol=0, atol=0)
    close(gs, (gr * xr).sum((0, 1)), rtol=1e-5, atol=1e-5)


def test_sample_norm():
    x = rnd(3, 4, 50, 7, seed=8, scale=2.0) + 0.3
    ms = K.sample_meanstd(x.cuda())
    mean, std = x.mean((1, 2, 3)), x.std((1, 2, 3))
    close(ms, torch.stack([mean, std], 1), rtol=1e-6, atol=1e-6)
    y = K.sample_norm(x.cuda(), ms, False)
    close(y, (x - mean.view(3, 1, 1, 1)) / (1e-5 + std.view(3, 1, 1, 1)), rtol=1e-5, atol=1e-6)
    close(K.sample_norm(y, ms, True), x, rtol=1e-5, atol=1e-5)


def _ref_spec(x, nfft):
    """HTDemucsQ._spec (htdemucsq.py:931-950) with demucs.spec.spectro = torch.stft(normalized, centred, reflect, Hann)"""
    hl = nfft // 4
    le = -(-x.shape[-1] // hl)
    pad = hl // 2 * 3
    xp = F.pad(x, (pad, pad + le * hl - x.shape[-1]), mode="reflect")
    z = torch.stft(xp, nfft, hl, window=torch.hann_window(nfft), win_length=nfft, normalized=True, center=True, return_complex=True,
                   pad_mode="reflect")[..., :-1, :]
    assert z.shape[-1] == le + 4
    return z[..., 2:2 + le]


def _ref_ispec(z, nfft, length):
    """HTDemucsQ._ispec (htdemucsq.py:952-960) with demucs.spec.ispectro = torch.istft"""
    hl = nfft // 4
    z = F.pad(F.pad(z, (0, 0, 0, 1)), (2, 2))
    pad = hl // 2 * 3
    le = hl * -(-length // hl) + 2 * pad
    x = torch.istft(z, nfft, hl, window=torch.hann_window(nfft), win_length=nfft, normalized=True, length=le, center=True)
    return x[..., pad:pad + length]


@pytest.mark.parametrize("nfft,L,rows", [(64, 333, 3), (4096, 20000, 2), (256, 1024, 1)])
def test_stft_istft(nfft, L, rows):
    hl = nfft // 4
    le, pad = -(-L // hl), hl // 2 * 3
    x = rnd(rows, L, seed=9)
    want = _ref_spec(x, nfft)                                        # [rows, nfft/2, le] complex
    z = K.stft(padded(x), nfft, hl, le, pad)                         # [rows, 2, le, nfft/2]
    zt = K.transpose2d(z)                                            # [rows, 2, nfft/2, le]
    close(zt[:, 0], want.real, rtol=1e-4, atol=2e-5)
    close(zt[:, 1], want.imag, rtol=1e-4, atol=2e-5)
    # inverse on a random spectrum, and its adjoint
    zr = torch.complex(rnd(rows, nfft // 2, le, seed=10), rnd(rows, nfft // 2, le, seed=11)).requires_grad_(True)
    y_ref = _ref_ispec(zr, nfft, L)
    g = rnd(rows, L, seed=12)
    y_ref.backward(g)
    planes = torch.stack([zr.detach().real, zr.detach().imag], 1).cuda()                # [rows, 2, nfft/2, le]
    y = K.istft(K.transpose2d(planes), nfft, hl, pad, L)
    close(y, y_ref, rtol=1e-4, atol=2e-5)
    gz = K.transpose2d(K.istft_bwd(padded(g), nfft, hl, pad, le))
    close(gz[:, 0], zr.grad.real, rtol=1e-4, atol=2e-5)
    # torch's complex gradient convention: grad = dL/dRe + i dL/dIm
    close(gz[:, 1], zr.grad.imag, rtol=1e-4, atol=2e-5)


def test_attn_long_short_key_tile_ignores_stale_lds():
    """Lk below one LDS tile: the padding keys carry probability 0, whatever the tile held before (a NaN-filled launch first)"""
    B, nh, hd, Lq, Lk = 2, 2, 8, 9, 20
    E = nh * hd
    bad = torch.full((B, 64, E), float("nan"), device="cuda")
    K.attn_long_fwd(bad, bad, bad, nh, True)
    q, k, v = rnd(B, Lq, E, seed=1).cuda(), rnd(B, Lk, E, seed=2).cuda(), rnd(B, Lk, E, seed=3).cuda()
    o, stats = K.attn_long_fwd(q, k, v, nh, True)
    assert torch.isfinite(o).all()
    h = lambda t, L: t.reshape(B, L, nh, hd).permute(0, 2, 1, 3)
    want = (torch.softmax(h(q, Lq) @ h(k, Lk).transpose(-1, -2), -1) @ h(v, Lk)).permute(0, 2, 1, 3).reshape(B, Lq, E)
    close(o, want, rtol=2e-5, atol=2e-5)
    gq, gk, gv = K.attn_long_bwd(q, k, v, o, rnd(B, Lq, E, seed=4).cuda(), stats, nh, True)
    assert torch.isfinite(gq).all() and torch.isfinite(gk).all() and torch.isfinite(gv).all()


@pytest.mark.gpu
@pytest.mark.parametrize("L,B,E,nh", [(250, 7, 64, 4), (63, 33, 256, 8), (5, 3, 16, 1)])
def test_mha_prep_matches_the_unfused_quantizer_chain(L, B, E, nh):
    """fqss_mha_prep_fwd / _bwd (the q / k / v / div quantizers and q / sqrt(head_dim) of MultiheadAttentionQ, qat_layers.py:890-905, as
    one pass each way) against the chain it replaces -- three fqss_actq_fwd on the thirds, fqss_unary_fwd(DIVS), fqss_actq_fwd and
    their backward launches: q / k / v and the input gradient bit for bit, the four quantizers' range partials to summation order"""
    import math
    from fqss_amd import kernels as K
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(L * 1000 + E)
    X = torch.randn(L, B, 3 * E, device=dev, generator=g) * 1.3
    scale = math.sqrt(E // nh)
    rng = [(torch.tensor([lo], device=dev), torch.tensor([hi], device=dev)) for lo, hi in ((-2.1, 2.4), (-1.0, 3.0), (-2.9, 1.1), (-0.4, 0.6))]
    q, k, v = K.mha_prep_fwd(X, E, scale, rng)
    parts = [K.actq_fwd(X[..., i * E:(i + 1) * E], K.ACT_NONE, None, K.Q_QUANT, rng[i][0], rng[i][1], None) for i in range(3)]
    qd = K.unary_fwd(parts[0], K.UNARY_DIVS, scale)
    q_ref = K.actq_fwd(qd, K.ACT_NONE, None, K.Q_QUANT, rng[3][0], rng[3][1], None)
    assert torch.equal(q, q_ref) and torch.equal(k, parts[1]) and torch.equal(v, parts[2])
    gq, gk, gv = (torch.randn(L, B, E, device=dev, generator=g) for _ in range(3))
    gaccs = [torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev) for _ in range(4)]
    gX = K.mha_prep_bwd(X, gq, gk, gv, E, scale, rng, gaccs)
    refs = [torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev) for _ in range(4)]
    g2 = K.unary_bwd(K.actq_bwd(qd, gq, K.ACT_NONE, None, K.Q_QUANT, rng[3][0], rng[3][1], refs[3]), None, K.UNARY_DIVS, scale)
    want = [K.actq_bwd(X[..., i * E:(i + 1) * E].contiguous(), gi, K.ACT_NONE, None, K.Q_QUANT, rng[i][0], rng[i][1], refs[i])
            for i, gi in enumerate((g2, gk, gv))]
    for i in range(3):
        assert torch.equal(gX[..., i * E:(i + 1) * E], want[i]), i
    for a, b in zip(gaccs, refs):
        np.testing.assert_allclose(a.view(-1, 3).sum(0).cpu().numpy(), b.view(-1, 3).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("Co,Ci,pro,act,split,B,M", [
    (256, 512, 1, 0, 128, 3, 300),      # T3 of the teacher chain: GroupNorm prologue, res | skip split, both residuals, ragged last tile
    (512, 128, 0, 1, 512, 2, 391),      # T1: PReLU + GroupNorm statistics (k_tgemm_k128 since round 5; k_tgemm2<0> behind FQSS_T1_K128=0)
    (1024, 128, 2, 2, 1024, 1, 130),    # mask conv: PReLU prologue, ReLU, four row tiles (k_tgemm2<2>: eight k-tiles of 16)
    (256, 64, 1, 0, 256, 2, 128),       # Ci = 64 < 128: not a shape of the tiled form -> k_tgemm, two 32-deep k-tiles (its peeled head and tail meet)
    (256, 256, 0, 1, 256, 2, 200),      # no prologue at Ci = 256: k_tgemm2<0> (Ci = 128 goes to k_tgemm_k128)
    (256, 384, 1, 0, 256, 2, 150),      # 24 k-tiles of 16: the longest straight-line loader instantiation of k_tgemm2 but one (Ci = 512)
    (128, 512, 1, 0, 128, 2, 200),      # bottleneck conv: fewer than 256 rows -> the round-3 kernel (k_tgemm) keeps serving it
    (256, 128, 0, 0, 256, 2, 260),      # Ci = 128 without a prologue: k_tgemm_k128 (two row blocks of 128), no statistics, no activation
    (512, 128, 0, 1, 512, 3, 1000),     # T1 again (k_tgemm_k128: weights in registers): several column tiles per workgroup, samples change under way
    (128, 128, 0, 1, 128, 2, 70),       # ... one row block, a ragged second column tile, fewer tiles than column groups
    (512, 512, 1, 2, 512, 1, 140),      # two row tiles behind a GroupNorm prologue (bias from global memory), 16 k-tiles
    (512, 256, 2, 0, 512, 2, 129),      # sixteen k-tiles of 16 behind a PReLU prologue
])
def test_teacher_gemm_against_fp64(Co, Ci, pro, act, split, B, M):
    """fqss_tgemm (csrc/teacher.hip: k_tgemm2, the 256-row form with the weight planes moved by LDS-DMA, and k_tgemm for the other
    shapes) against fp64 math on the fp32 operands: the six-product bf16 split is fp32-grade, so the result must sit within a few fp32
    roundings of the exact sum; the GroupNorm statistics of the epilogue against fp64 sums of what was written."""
    dev = "cuda"
    g = torch.Generator().manual_seed(Co + Ci + M)
    R = lambda *s: torch.randn(*s, generator=g)
    w, bias = R(Co, Ci) / Ci ** 0.5, R(Co) * 0.1
    x = K.empty_act((B, Ci, M), dev)
    x.copy_(R(B, Ci, M).to(dev) * 1.5 + 0.3)
    gamma, beta = (1.0 + 0.2 * R(Ci)).to(dev), (0.1 * R(Ci)).to(dev)
    slope_in, slope_out = torch.tensor([0.2], device=dev), torch.tensor([0.3], device=dev)
    planes = K.split3_planes(w.to(dev))
    st_in = K.tstat_buffer(1, B, dev)[0]
    K.tstats(x, st_in)
    so = K.tstat_buffer(1, B, dev)[0]
    r1 = K.empty_act((B, split, M), dev)
    r1.copy_(R(B, split, M).to(dev))
    r2 = None
    if split < Co:
        r2 = K.empty_act((B, Co - split, M), dev)
        r2.copy_(R(B, Co - split, M).to(dev))
    use_res = pro == 1 and Co == 256 and Ci == 512
    out = K.tgemm(planes, x, bias.to(dev), act=act, slope=slope_out if act == 1 else None, pro=pro, pro_stats=st_in if pro == 1 else None,
                  pro_gamma=gamma if pro == 1 else None, pro_beta=beta if pro == 1 else None, pro_eps=1e-8,
                  pro_slope=slope_in if pro == 2 else None, stats_out=so if act == 1 else None, M1=split,
                  r1=r1 if use_res else None, r2=r2 if use_res else None)
    c = torch.cat(list(out), 1) if isinstance(out, tuple) else out
    xd = x.double().cpu()
    if pro == 1:
        mu = xd.mean(dim=(1, 2), keepdim=True)
        var = (xd * xd).mean(dim=(1, 2), keepdim=True) - mu * mu
        xd = (xd - mu) / torch.sqrt(var + 1e-8) * gamma.double().cpu().view(1, -1, 1) + beta.double().cpu().view(1, -1, 1)
    elif pro == 2:
        xd = torch.where(xd > 0, xd, 0.2 * xd)
    ref = torch.einsum("oc,bcm->bom", w.double(), xd) + bias.double().view(1, -1, 1)
    if act == 1:
        ref = torch.where(ref > 0, ref, float(slope_out.item()) * ref)
    elif act == 2:
        ref = ref.clamp(min=0)
    if use_res:
        ref = ref + torch.cat([r1, r2], 1).double().cpu()
    err = float((c.double().cpu() - ref).abs().max())
    assert err <= 4e-6 * float(ref.abs().max()), (err, float(ref.abs().max()))
    if act == 1:
        s = so.double().cpu()[:, :, :2].sum(1)
        cd = c.double().cpu()
        np.testing.assert_allclose(s[:, 0].numpy(), cd.sum(dim=(1, 2)).numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(s[:, 1].numpy(), (cd * cd).sum(dim=(1, 2)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("B,C,M", [(2, 24, 3999), (1, 16, 3999), (3, 8, 777), (2, 600, 130), (1, 4, 4096), (8, 512, 3999), (2, 7, 9001), (1, 6, 5),
                                   (3, 250, 64)])
@pytest.mark.parametrize("write_out", [False, True])
def test_gnq_apply_code_table_bit_identical(B, C, M, write_out, monkeypatch):
    """k_gnq_apply_t (round 5, VERDICT r04 next #1a): per (b, c) row the output code of GroupNormQ is a 256-entry function of the input
    code -- one evaluation of the quantizer arithmetic per code and row, an LDS table, out = T[in] -- against the per-element kernel of
    rounds 1-4 (FQSS_GNQ_APPLY_V1=1): codes, mean / rstd and the optional fp32 copy bit for bit.  Shapes: the gate shapes of
    test_gn_dw_fused_bit_identical, the cfg-2 shape (4 rows per workgroup), C not divisible by 2 / 4 (1 / 2 rows per workgroup), rows
    longer than a workgroup pass, rows shorter than one lane's 16 codes.  Reference: qat_layers.py:438-452, qat_quant.py:136-147."""
    dev = "cuda"
    g = torch.Generator().manual_seed(B * 1000 + C + M)
    xc = K.empty_codes((B, C, M), dev)
    xc.copy_(torch.randint(0, 256, (B, C, M), generator=g, dtype=torch.uint8).to(dev))
    T1 = lambda v: torch.tensor([v], device=dev)
    lo, hi, lo1, hi1 = T1(-1.7), T1(2.9), T1(-2.2), T1(2.4)
    gamma, beta = (1.0 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    xi = xc.to(torch.int64)
    st = K.CodeStats(torch.stack([xi.sum(dim=(1, 2)), (xi * xi).sum(dim=(1, 2))], 1).reshape(-1).contiguous(), 1)
    res = {}
    for v1 in ("1", "0"):
        monkeypatch.setenv("FQSS_GNQ_APPLY_V1", v1)
        for stats in (st, None):        # statistics handed over by the producer / taken by the layer's own pass
            out, yc, mr = K.gnq_fwd(xc, lo, hi, gamma, beta, 1e-8, lo1, hi1, write_out=write_out, stats=stats)
            torch.cuda.synchronize()
            res[v1, stats is None] = (yc.clone(), mr.clone(), out[..., :M].clone() if write_out else None)
    for own in (False, True):
        a, b = res["1", own], res["0", own]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        if write_out:
            assert torch.equal(a[2], b[2])
    assert torch.equal(res["0", False][0], res["0", True][0])
    assert len(torch.unique(res["0", False][0])) > min(40, M // 2)          # (a constant output would pass trivially)


@pytest.mark.parametrize("B,C,M,dil", [(2, 24, 3999, 1), (1, 16, 3999, 128), (3, 8, 777, 4), (2, 300, 130, 2), (1, 4, 4096, 64), (2, 5, 1023, 8)])
@pytest.mark.parametrize("which", ["after", "before", "both"])
def test_depthwise_backward_takes_the_groupnorm_passes(B, C, M, dil, which):
    """k_dwq_bwd<3, GA, GB> (round 5, VERDICT r04 next #1b): gLN -> depthwise Conv1dNlQ (PReLU) -> gLN on codes, backward.  The depthwise
    layer's backward takes the APPLY pass of the GroupNormQ behind it (on the incoming gradient, by a per-row table over its own output
    code) and / or the ROWS pass of the GroupNormQ in front of it (on the gx it produces, by a per-row table over that GroupNorm's
    input code) -- against the chain of separate launches (fqss_gnq_bwd, fqss_dwq_bwd, fqss_gnq_bwd): every gradient tensor BIT for
    bit (same per-element arithmetic, same thread -> element order), the gamma / beta gradients bit for bit, the fp64 range partials
    and the fp32-atomic weight / bias sums to their summation-order noise.  Reference: convtasnetq.py:28-30, qat_layers.py:438-452."""
    dev = "cuda"
    g = torch.Generator().manual_seed(B * 1000 + C + M + dil)
    T1 = lambda v: torch.tensor([v], device=dev)
    lo0, hi0, lo1, hi1, lo2, hi2, lo3, hi3 = T1(-1.7), T1(2.9), T1(-2.2), T1(2.4), T1(-0.6), T1(1.9), T1(-1.5), T1(1.2)
    rnd_ = lambda *sh, s=1.0: (torch.randn(*sh, generator=g) * s).to(dev)
    gamma1, beta1, gamma2, beta2 = 1.0 + rnd_(C, s=0.3), rnd_(C, s=0.2), 1.0 + rnd_(C, s=0.3), rnd_(C, s=0.2)
    w, bias, slope = rnd_(C, 1, 3, s=0.6), rnd_(C, s=0.1), T1(0.2)
    x0c = K.empty_codes((B, C, M), dev)
    x0c.copy_(torch.randint(0, 256, (B, C, M), generator=g, dtype=torch.uint8).to(dev))
    _, y1c, mr1 = K.gnq_fwd(x0c, lo0, hi0, gamma1, beta1, 1e-8, lo1, hi1, write_out=False)
    std = K.new_stats("dwq", B, C, M, dev)
    _, y2c = K.dwq_fwd(y1c, lo1, hi1, w, bias, dil, dil, K.ACT_PRELU, slope, lo2, hi2, write_out=False, stats=std)
    _, y3c, mr2 = K.gnq_fwd(y2c, lo2, hi2, gamma2, beta2, 1e-8, lo3, hi3, write_out=False, stats=std)
    g3 = K.empty_act((B, C, M), dev).copy_(rnd_(B, C, M, s=1e-2))
    mk = lambda: dict(gacc1=torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev), gacc2=torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev),
                      gacc3=torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev), gg1=rnd_(C, s=0.1), gb1=rnd_(C, s=0.1), gg2=rnd_(C, s=0.1),
                      gb2=rnd_(C, s=0.1), gbias=torch.zeros(C, device=dev), gw=torch.zeros(C, 1, 3, device=dev))
    R = mk()
    F = {k: v.clone() for k, v in R.items()}
    # the separate launches
    gx2 = K.gnq_bwd(y2c, lo2, hi2, g3, gamma2, beta2, mr2, lo3, hi3, R["gacc3"], R["gg2"], R["gb2"])
    gx1 = K.dwq_bwd(y1c, lo1, hi1, w, bias, gx2, dil, dil, K.ACT_PRELU, slope, lo2, hi2, R["gacc2"], R["gbias"], R["gw"])
    gx0 = K.gnq_bwd(x0c, lo0, hi0, gx1, gamma1, beta1, mr1, lo1, hi1, R["gacc1"], R["gg1"], R["gb1"])
    # the hand-over forms
    after = before = None
    if which in ("after", "both"):
        gin, ws2 = K.gnq_bwd_rows(y2c, lo2, hi2, g3, gamma2, beta2, mr2, lo3, hi3, F["gacc3"])
        after = dict(gamma=gamma2, beta=beta2, mean_rstd=mr2, ws=ws2, qmin=lo3, qmax=hi3, ggamma=F["gg2"], gbeta=F["gb2"])
    else:
        gin = K.gnq_bwd(y2c, lo2, hi2, g3, gamma2, beta2, mr2, lo3, hi3, F["gacc3"], F["gg2"], F["gb2"])
    if which in ("before", "both"):
        before = dict(xc0=x0c, qmin0=lo0, qmax0=hi0, gamma=gamma1, beta=beta1, mean_rstd=mr1, gacc=F["gacc1"])
    fx1 = K.dwq_bwd(y1c, lo1, hi1, w, bias, gin, dil, dil, K.ACT_PRELU, slope, lo2, hi2, F["gacc2"], F["gbias"], F["gw"], after=after, before=before)
    if before is not None:
        fx0 = K.gnq_bwd_apply(x0c, lo0, hi0, fx1, gamma1, beta1, mr1, lo1, hi1, before["ws"], F["gg1"], F["gb1"])
    else:
        fx0 = K.gnq_bwd(x0c, lo0, hi0, fx1, gamma1, beta1, mr1, lo1, hi1, F["gacc1"], F["gg1"], F["gb1"])
    torch.cuda.synchronize()
    assert torch.equal(fx1, gx1), float((fx1 - gx1).abs().max())
    assert torch.equal(fx0, gx0), float((fx0 - gx0).abs().max())
    for k in ("gg1", "gb1", "gg2", "gb2"):
        assert torch.equal(F[k], R[k]), k
    for k in ("gacc1", "gacc2", "gacc3"):
        a, b = F[k].view(-1, 3).sum(0), R[k].view(-1, 3).sum(0)
        assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(b.abs().max())), (k, a, b)
        assert float(b.abs().max()) > 0
    for k in ("gbias", "gw"):
        assert float((F[k] - R[k]).abs().max()) <= 1e-5 * max(1e-6, float(R[k].abs().max())), k


@pytest.mark.parametrize("B,C,M,dil", [(2, 24, 3999, 1), (1, 16, 3999, 128), (3, 8, 777, 4), (2, 600, 130, 2), (1, 4, 4096, 64)])
def test_gn_dw_fused_bit_identical(B, C, M, dil):
    """(EXPERIMENT, include/fqss_experiments.h: not in the product library -- runs when FQSS_LIB points at `make -C fqss_amd/csrc
    experiments`' variants/libfqss_experiments.so, skipped otherwise.)  fqss_gndwq_fwd (round 5: on per-row code tables, k_gndwq_fwd_t;
    the per-element form of round 4 is fqss_gndwq_fwd_v1): GroupNormQ + 3-tap depthwise Conv1dNlQ (PReLU), both quantizing, as ONE launch -- against the two
    launches it replaces (fqss_gnq_fwd, fqss_dwq_fwd): the GroupNorm's output codes, its mean / rstd, the depthwise layer's output
    codes and the integer statistics of those codes, bit for bit (dilations 1 .. 128: unaligned taps out of the LDS row, the
    zero-padded row ends, rows shorter than a workgroup's 4096 positions, several rows per workgroup)."""
    from fqss_amd import _lib
    if not hasattr(_lib.load(), "fqss_gndwq_fwd"):
        pytest.skip("fqss_gndwq_fwd is an experiment: build `make -C fqss_amd/csrc experiments` and set FQSS_LIB to run this gate")
    dev = "cuda"
    g = torch.Generator().manual_seed(B * 1000 + C + M + dil)
    xc = K.empty_codes((B, C, M), dev)
    xc.copy_(torch.randint(0, 256, (B, C, M), generator=g, dtype=torch.uint8).to(dev))
    T1 = lambda v: torch.tensor([v], device=dev)
    lo, hi, lo1, hi1, lo2, hi2 = T1(-1.7), T1(2.9), T1(-2.2), T1(2.4), T1(-0.6), T1(1.9)
    gamma, beta = (1.0 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    w, bias = (torch.randn(C, 1, 3, generator=g) * 0.6).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    slope = T1(0.2)
    xi = xc.to(torch.int64)
    st = K.CodeStats(torch.stack([xi.sum(dim=(1, 2)), (xi * xi).sum(dim=(1, 2))], 1).reshape(-1).contiguous(), 1)
    _, yc1, mr = K.gnq_fwd(xc, lo, hi, gamma, beta, 1e-8, lo1, hi1, write_out=False, stats=st)
    std = K.new_stats("dwq", B, C, M, dev)
    _, yc2 = K.dwq_fwd(yc1, lo1, hi1, w, bias, dil, dil, K.ACT_PRELU, slope, lo2, hi2, write_out=False, stats=std)
    _, y1, mr_f = K.gnq_fwd_deferred(xc)
    d = dict(xc=xc, qmin_x=lo, qmax_x=hi, gamma=gamma, beta=beta, eps=1e-8, qmin=lo1, qmax=hi1, stats=st, yc=y1, mean_rstd=mr_f, done=False)
    _, y2, st2 = K.gndwq_fwd(d, w, bias, dil, dil, K.ACT_PRELU, slope, lo2, hi2, True)
    assert torch.equal(y1, yc1) and torch.equal(mr_f, mr)
    assert torch.equal(y2, yc2)
    if std is not None and st2 is not None:
        a = std.ws.view(B, -1, 2).sum(1)
        b = st2.ws.view(B, -1, 2).sum(1)
        assert torch.equal(a, b)
        yi = yc2.to(torch.int64)
        assert torch.equal(b[:, 0], yi.sum(dim=(1, 2))) and torch.equal(b[:, 1], (yi * yi).sum(dim=(1, 2)))


def test_unary_maps_read_a_column_block_in_place():
    """fqss_unary_rows_fwd (round 6): the float MultiheadAttention's `q / sqrt(head_dim)` (qat_layers.py:889-901 with the quantizers off: the
    teacher of cfg 3 / 4 / 5) on the q THIRD of the in-projection [L, B, 3E] read in place -- equal, bit for bit, to the map of a
    `.contiguous()` copy of the slice, for every map kind and for views that do not qualify (odd width: the copy path)"""
    g = torch.Generator().manual_seed(5)
    X = torch.randn(37, 5, 3 * 64, generator=g).cuda()
    for kind, p in ((K.UNARY_TANH, 1.0), (K.UNARY_SIGMOID, 1.0), (K.UNARY_DIVS, 4.0), (K.UNARY_GELU, 1.0)):
        for lo in (0, 64, 128):
            v = X[..., lo:lo + 64]
            assert not v.is_contiguous() and K.as_rowmat_view(v) == (37 * 5, 64, 192)
            assert torch.equal(K.unary_fwd(v, kind, p), K.unary_fwd(v.contiguous(), kind, p))
    odd = X[..., 1:63]                                   # not 16-B aligned, not a multiple of 4: the copy path
    assert K.as_rowmat_view(odd) is None
    assert torch.equal(K.unary_fwd(odd, K.UNARY_DIVS, 3.0), K.unary_fwd(odd.contiguous(), K.UNARY_DIVS, 3.0))
    assert torch.equal(K.unary_fwd(X[..., :64], K.UNARY_DIVS, 8.0), X[..., :64] / 8.0)       # (a power of two: exact either way)


@pytest.mark.parametrize("cols,ld,trail", [(431, 432, 0), (110250 % 1000 + 250, 512, 0), (510, 512, 2), (62, 128, 66), (7, 16, 0), (433, 448, 15)])
def test_activation_quantizer_stores_nothing_past_the_last_column(cols, ld, trail):
    """fqss_actq_fwd / fqss_actq_bwd (round 6): every 16-B aligned row takes the 16-B form -- rows padded to 64 B by more than three
    floats ([B, C, 110250] -> 110256) used to fall to the one-element kernel -- and the last, partial group of a row is stored element by
    element (store_group4): the output may be a COLUMN BLOCK of a wider matrix whose next columns hold data.  Against the dense
    call: same values; the columns behind the block keep their contents (`trail` of them are real data in the cases that have any)."""
    dev = "cuda"
    g = torch.Generator().manual_seed(cols * 7 + ld)
    rows = 37
    z = torch.randn(rows, cols, generator=g).to(dev)
    gy = torch.randn(rows, cols, generator=g).to(dev)
    lo, hi = torch.tensor([-0.8], device=dev), torch.tensor([1.1], device=dev)
    slope = torch.tensor([0.2], device=dev)
    # reference: dense operands
    gacc0 = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
    out0, idx0 = K.actq_fwd(z.clone(), K.ACT_PRELU, slope, K.Q_QUANT, lo, hi, None, want_idx=True, dense_idx=True)
    gz0 = K.actq_bwd(z.clone(), gy.clone(), K.ACT_PRELU, slope, K.Q_QUANT, lo, hi, gacc0)
    # the same rows as the leading columns of wider buffers, sentinel values behind them
    zb, gb = torch.full((rows, ld), 7.5, device=dev), torch.full((rows, ld), -3.25, device=dev)
    zb[:, :cols], gb[:, :cols] = z, gy
    ob = torch.full((rows, ld), 123.0, device=dev)
    gacc1 = torch.zeros_like(gacc0)
    gz1 = K.actq_bwd(zb[:, :cols], gb[:, :cols], K.ACT_PRELU, slope, K.Q_QUANT, lo, hi, gacc1, out=ob[:, :cols])
    torch.cuda.synchronize()
    assert torch.equal(gz1, gz0[:, :cols] if gz0.shape[1] != cols else gz0) or torch.equal(ob[:, :cols], gz0[..., :cols])
    assert float((ob[:, cols:] - 123.0).abs().max()) == 0, "columns behind the block were written"
    a, b = gacc1.view(-1, 3).sum(0), gacc0.view(-1, 3).sum(0)          # (per-thread fp32 partials: the dense rows take another thread -> element map)
    assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
    out1 = K.actq_fwd(zb[:, :cols], K.ACT_PRELU, slope, K.Q_QUANT, lo, hi, None)
    assert torch.equal(out1[..., :cols], out0[..., :cols])


@pytest.mark.parametrize("shape", [(2, 5, 7, 431), (1, 3, 4, 64), (3, 2, 9, 33)])
def test_permute4_into_row_padded_output_and_unary_maps_on_padded_activations(shape):
    """fqss_permute4_ld (round 6): the [B, P, Q, T] -> [B, Q, P, T] move into a row-padded activation (rows 16-B aligned, padding
    zero-filled) equals the dense move; kernels.unary_fwd / unary_bwd map a row-padded activation where it lies (kernels.padded_dense)
    and equal the maps of its dense copy."""
    dev = "cuda"
    B, P, Q, T = shape
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(T)).to(dev)
    sB, sP, sQ, _ = x.stride()
    dense = K.permute4(x, (B, Q, P), (sB, sQ, sP), T, dense=False)
    padded = K.permute4(x, (B, Q, P), (sB, sQ, sP), T, dense=False, pad_out=True)
    assert padded.shape == dense.shape == (B, Q, P, T) and torch.equal(padded, dense) and torch.equal(dense, x.transpose(1, 2))
    if T % 4:
        assert padded.stride(-2) % 4 == 0 and padded.stride(-2) > T
        buf = padded.as_strided((B * Q * P, padded.stride(-2)), (padded.stride(-2), 1))
        assert float(buf[:, T:].abs().max()) == 0                     # the padding columns hold zeros
    a = K.empty_act((B, P * Q, T), dev).copy_(x.reshape(B, P * Q, T))
    if K.padded_dense(a) is not None:
        for kind in (K.UNARY_GELU, K.UNARY_TANH):
            y0 = K.unary_fwd(a.contiguous(), kind)
            y1 = K.unary_fwd(a, kind)
            assert not y1.is_contiguous() and torch.equal(y1, y0)
            g = torch.randn(a.shape, generator=torch.Generator().manual_seed(1)).to(dev)
            ref_in = a.contiguous() if kind == K.UNARY_GELU else y0
            arg_in = a if kind == K.UNARY_GELU else y1
            assert torch.equal(K.unary_bwd(g, arg_in, kind), K.unary_bwd(g, ref_in, kind))
