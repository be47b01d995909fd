"""Descriptor-struct forms of the long C entry points (include/fqss.h "Descriptor-struct forms"): same
kernels and results as the flat forms the Python host calls; descriptors are validated before anything is launched."""
import ctypes as C

import pytest
import torch

from fqss_amd import _lib
from fqss_amd import kernels as K

_DT = {torch.float32: _lib.DT_F32, torch.uint8: _lib.DT_U8, torch.int8: _lib.DT_I8, torch.float64: _lib.DT_F64,
       torch.int16: _lib.DT_U16, torch.int64: _lib.DT_I64}


def desc(t, dtype=None):
    d = _lib.FqssTensor()
    d.data, d.dtype, d.ndim = t.data_ptr(), _DT[t.dtype] if dtype is None else dtype, t.dim()
    for i in range(t.dim()):
        d.shape[i], d.stride[i] = t.shape[i], t.stride(i)
    return d


def qp(lo, hi, act=0, slope=None, gacc=None):
    q = _lib.FqssQParams()
    q.qmin, q.qmax, q.act = lo.data_ptr(), hi.data_ptr(), act
    q.slope = slope.data_ptr() if slope is not None else None
    q.gacc = gacc.data_ptr() if gacc is not None else None
    return q


def ref(x):
    return C.byref(x) if x is not None else None


def test_workspace_bytes_and_descriptor_validation_need_no_gpu():
    shp = (C.c_int64 * 3)(8, 512, 3999)
    assert _lib.query("fqss_workspace_bytes", b"gln_fq_fwd", shp, 3) == 2 * 64 * 8 * 8
    assert _lib.query("fqss_workspace_bytes", b"gln_fq_bwd", shp, 3) == (2 * 8 * 512 + 2 * 8) * 8
    assert _lib.query("fqss_workspace_bytes", b"pwconv_fq_fwd", shp, 3) == 8 * _lib.query("fqss_qpw_stat_slots", 512, 3999) * 16
    assert _lib.query("fqss_workspace_bytes", b"dwconv_fq_fwd", shp, 3) == 8 * _lib.query("fqss_dwq_stat_slots", 512, 3999) * 16
    assert _lib.query("fqss_workspace_bytes", b"add_fq_bwd", shp, 3) == 0
    assert _lib.query("fqss_workspace_bytes", b"no_such_op", shp, 3) == -1
    assert b"no_such_op" in _lib.load().fqss_last_error()
    # a descriptor with a null data pointer / a strided innermost dimension is refused before any launch
    t = _lib.FqssTensor()
    t.dtype, t.ndim = _lib.DT_U8, 3
    q = _lib.FqssQParams()
    q.qmin = q.qmax = 16      # never dereferenced on the host
    with pytest.raises(_lib.FqssError, match="null tensor"):
        _lib.call("fqss_add_fq_fwd", C.byref(t), C.byref(q), None, None, 1.0, C.byref(t), None, C.byref(q), None, 0, None)
    t.data = 4096
    t.shape[:] = [1, 4, 16, 0]
    t.stride[:] = [64, 16, 2, 0]
    with pytest.raises(_lib.FqssError, match="innermost stride"):
        _lib.call("fqss_add_fq_fwd", C.byref(t), C.byref(q), None, None, 1.0, C.byref(t), None, C.byref(q), None, 0, None)


@pytest.mark.gpu
def test_descriptor_forms_match_the_flat_forms():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    B, Cb, Ch, M = 2, 128, 256, 1000
    lo, hi = torch.tensor([-1.5], device=dev), torch.tensor([2.0], device=dev)
    lo2, hi2 = torch.tensor([-2.5], device=dev), torch.tensor([1.0], device=dev)
    slope = torch.tensor([0.25], device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def codes(C_):
        return K.empty_codes((B, C_, M), dev).random_(0, 256, generator=g)

    def act(C_):
        return K.empty_act((B, C_, M), dev).normal_(generator=g)

    # AddQ forward
    a, b = codes(Cb), codes(Cb)
    _, yc = K.ewq_fwd(a, lo, hi, b, lo2, hi2, None, 1.0, 0, None, lo, hi2, False)
    y2 = K.empty_codes((B, Cb, M), dev)
    da, db, dy = desc(a), desc(b), desc(y2)
    qa, qb, qo = qp(lo, hi), qp(lo2, hi2), qp(lo, hi2)
    _lib.call("fqss_add_fq_fwd", ref(da), ref(qa), ref(db), ref(qb), 1.0, ref(dy), None, ref(qo), None, 0, st)
    assert torch.equal(y2[..., :M], yc[..., :M])

    # AddQ backward with both producers fused (the residual add of a TCN block)
    gin, za, zb = act(Cb), act(Cb), act(Cb)
    gacc, pga, pgb = (torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev) for _ in range(3))
    gba, gbb = torch.zeros(Cb, device=dev), torch.zeros(Cb, device=dev)
    _, gza, gzb = K.ewq_bwd_p(a, lo, hi, b, lo2, hi2, 1.0, gin, 0, None, lo, hi2, gacc, Cb, prod_a=(za, 0, None, pga, gba),
                              prod_b=(zb, 0, None, pgb, gbb))
    gacc2, pga2, pgb2 = (torch.zeros_like(gacc) for _ in range(3))
    gba2, gbb2 = torch.zeros_like(gba), torch.zeros_like(gbb)
    oa, ob = K.empty_act((B, Cb, M), dev), K.empty_act((B, Cb, M), dev)
    dza, dzb, doa, dob, dg = desc(za), desc(zb), desc(oa), desc(ob), desc(gin)
    pa, pb = _lib.FqssProducer(), _lib.FqssProducer()
    for p, dz, do, ga, gb_ in ((pa, dza, doa, pga2, gba2), (pb, dzb, dob, pgb2, gbb2)):
        p.z, p.out, p.act, p.slope, p.gacc, p.gbias = C.pointer(dz), C.pointer(do), 0, None, ga.data_ptr(), gb_.data_ptr()
    qo2 = qp(lo, hi2, gacc=gacc2)
    _lib.call("fqss_add_fq_bwd", ref(da), ref(qa), ref(db), ref(qb), 1.0, ref(dg), None, ref(qo2), ref(pa), ref(pb), None, 0, st)
    assert torch.equal(oa[..., :M], gza[..., :M]) and torch.equal(ob[..., :M], gzb[..., :M])
    for u, v in ((gacc, gacc2), (pga, pga2), (pgb, pgb2)):
        torch.testing.assert_close(u, v, rtol=1e-12, atol=1e-12)          # fp64 atomics: order only
    torch.testing.assert_close(gba, gba2, rtol=1e-5, atol=1e-5)

    # Conv1dNlQ 128 -> 256 with PReLU + output quantizer + code statistics, then GroupNormQ from those statistics
    x = codes(Cb)
    ones = torch.ones(Ch, 1, 1, device=dev) * 0.2
    wc = K.wq_codes(torch.randn(Ch, Cb, 1, device=dev, generator=g) * 0.05, -ones, ones)
    bias = torch.randn(Ch, device=dev, generator=g) * 0.1
    stats = K.new_stats("qpw", B, Ch, M, dev)
    z_f, yc_f = K.qpw_fwdq(x, wc, bias, None, lo, hi, Ch, 1, slope, (lo2, hi), stats=stats)
    shp = (C.c_int64 * 3)(B, Ch, M)
    need = _lib.query("fqss_workspace_bytes", b"pwconv_fq_fwd", shp, 3)
    assert need == stats.ws.numel() * 8 > 0
    ws = torch.zeros(need // 8, dtype=torch.int64, device=dev)
    z_d, yc_d = K.empty_act((B, Ch, M), dev), K.empty_codes((B, Ch, M), dev)
    w = _lib.FqssWCodes()
    w.idx, w.idxT, w.dw, w.rw, w.Co, w.Ci = wc.idx.data_ptr(), wc.idxT.data_ptr(), wc.dw.data_ptr(), wc.rw.data_ptr(), Ch, Cb
    dx, dz, dyc = desc(x), desc(z_d), desc(yc_d)
    qx, q1 = qp(lo, hi), qp(lo2, hi, act=1, slope=slope)
    _lib.call("fqss_pwconv_fq_fwd", ref(dx), ref(qx), ref(w), bias.data_ptr(), None, ref(dz), None, ref(dyc), None, ref(q1), None,
              ws.data_ptr(), need, st)
    assert torch.equal(z_d[..., :M], z_f[..., :M]) and torch.equal(yc_d[..., :M], yc_f[..., :M])
    assert torch.equal(ws, stats.ws)
    with pytest.raises(_lib.FqssError, match="workspace smaller"):
        _lib.call("fqss_pwconv_fq_fwd", ref(dx), ref(qx), ref(w), bias.data_ptr(), None, ref(dz), None, ref(dyc), None, ref(q1), None,
                  ws.data_ptr(), need - 8, st)

    gm, bt = torch.rand(Ch, device=dev, generator=g) + 0.5, torch.randn(Ch, device=dev, generator=g) * 0.1
    _, gy_f, mr_f = K.gnq_fwd(yc_f, lo2, hi, gm, bt, 1e-8, lo, hi2, False, stats=stats)
    gy_d, mr_d = K.empty_codes((B, Ch, M), dev), torch.empty(B, 2, device=dev)
    dgy = desc(gy_d)
    qin, qout = qp(lo2, hi), qp(lo, hi2)
    _lib.call("fqss_gln_fq_fwd", ref(dyc), ref(qin), gm.data_ptr(), bt.data_ptr(), 1e-8, ref(dgy), None, mr_d.data_ptr(), ref(qout), None, 0,
              ws.data_ptr(), stats.nslots, st)
    assert torch.equal(gy_d[..., :M], gy_f[..., :M]) and torch.equal(mr_d, mr_f)
    # ... and with its own statistics pass into a caller-provided workspace
    need_gn = _lib.query("fqss_workspace_bytes", b"gln_fq_fwd", shp, 3)
    ws_gn = torch.empty(need_gn // 8, dtype=torch.int64, device=dev)
    gy_d.zero_()
    _lib.call("fqss_gln_fq_fwd", ref(dyc), ref(qin), gm.data_ptr(), bt.data_ptr(), 1e-8, ref(dgy), None, mr_d.data_ptr(), ref(qout),
              ws_gn.data_ptr(), need_gn, None, 0, st)
    assert torch.equal(gy_d[..., :M], gy_f[..., :M]) and torch.equal(mr_d, mr_f)
    with pytest.raises(_lib.FqssError, match="workspace smaller"):
        _lib.call("fqss_gln_fq_fwd", ref(dyc), ref(qin), gm.data_ptr(), bt.data_ptr(), 1e-8, ref(dgy), None, mr_d.data_ptr(), ref(qout),
                  ws_gn.data_ptr(), need_gn - 8, None, 0, st)

    # teacher GEMM T1: 1x1 conv 128 -> 256 + PReLU + statistics
    xf = act(Cb)
    wt = torch.randn(Ch, Cb, device=dev, generator=g) * 0.05
    planes = K.split3_planes(wt)
    stats_f = torch.zeros(B * K.TSTAT_SLOTS * K.TSTAT_STRIDE, dtype=torch.float64, device=dev)
    stats_d = torch.zeros_like(stats_f)
    c_f, c_d = K.empty_act((B, Ch, M), dev), K.empty_act((B, Ch, M), dev)
    ld = K.rowmat(xf)[2]
    _lib.call("fqss_tgemm", planes.data_ptr(), xf.data_ptr(), B, Cb, Ch, M, ld, 0, None, None, None, 0.0, None, bias.data_ptr(), 1,
              slope.data_ptr(), stats_f.data_ptr(), Ch, c_f.data_ptr(), None, K.rowmat(c_f)[2], None, None, 0, st)
    td = _lib.FqssTGemmDesc()
    dpl, dxf, dc = desc(planes.view(3, Ch, Cb), _lib.DT_U16), desc(xf), desc(c_d)
    td.planes, td.x, td.c1 = C.pointer(dpl), C.pointer(dxf), C.pointer(dc)
    td.pro, td.bias, td.act, td.slope, td.stats_out, td.M1 = 0, bias.data_ptr(), 1, slope.data_ptr(), stats_d.data_ptr(), Ch
    _lib.call("fqss_tgemm_desc", ref(td), st)
    assert torch.equal(c_d[..., :M], c_f[..., :M])
    torch.testing.assert_close(stats_d, stats_f, rtol=1e-12, atol=1e-9)
    torch.cuda.synchronize()
