"""Thin torch-tensor front end of the C ABI: shape/stride checks on the host, then one ctypes
call per kernel on torch's current HIP stream.  No arithmetic happens in this file."""
import os

import torch

from . import _lib

Q_BYPASS, Q_OBSERVE, Q_QUANT = 0, 1, 2
ACT_NONE, ACT_PRELU, ACT_RELU = 0, 1, 2
ACT_GELU = 3          # nn.GELU (erf form) in front of a quantizer: fqss_actq_fwd / fqss_actq_bwd only (qat_layers.fq_node)
ACT_POST_RELU = 4     # a ReLU BEHIND the quantizer, relu(fq(x)): fqss_actq_fwd / fqss_actq_bwd only (qat_layers.fq_node(post_relu=True))

GACC_DOUBLES = 2048 * 3   # FQSS_GACC_SLOTS x (dmin, dmax, dslope)
USE_X3 = True            # route plain fp32 pointwise convs through the bf16 3x3-split GEMM
LD_ALIGN = 16  # row stride of activation buffers is padded to 16 floats (64 B) -> 16-B/lane path


def _p(t):
    return None if t is None else t.data_ptr()


import ctypes as _C

DESC_ABI = os.environ.get("FQSS_FLAT_ABI", "0") == "0"   # the long entry points go through their descriptor-struct forms (include/fqss.h)
_DT = {torch.float32: _lib.DT_F32, torch.uint8: _lib.DT_U8, torch.int8: _lib.DT_I8, torch.float64: _lib.DT_F64, torch.int16: _lib.DT_U16,
       torch.int64: _lib.DT_I64}


def _desc(t, dtype=None):
    """FqssTensor descriptor of a tensor (None -> None); the caller keeps it alive across the call"""
    if t is None:
        return None
    d = _lib.FqssTensor()
    d.data, d.dtype, d.ndim = t.data_ptr(), _DT[t.dtype] if dtype is None else dtype, t.dim()
    for i in range(t.dim()):
        d.shape[i], d.stride[i] = t.shape[i], t.stride(i)
    return d


def _ref(x):
    return _C.byref(x) if x is not None else None


def _qparams(qmin, qmax, act=0, slope=None, gacc=None):
    q = _lib.FqssQParams()
    q.qmin, q.qmax, q.act, q.slope, q.gacc = _p(qmin), _p(qmax), act, _p(slope), _p(gacc)
    return q


def _stream():
    return None if _lib.BACKEND == "cpu" else torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    """tensors must live where the SELECTED backend computes: ROCm device memory for the HIP library (the product; a CPU tensor raises --
    there is no fallback), host memory for the CPU backend of `--use_cpu` (_lib.set_backend)"""
    cpu = _lib.BACKEND == "cpu"
    for t in ts:
        if t is not None and t.is_cuda == cpu:
            raise _lib.FqssError("the CPU backend (--use_cpu) takes host tensors" if cpu else
                                 "fqss_amd ops need ROCm device tensors (no CPU fallback; see oracle/ for the CPU checker)")
        if t is not None and t.dtype not in (torch.float32,):
            raise _lib.FqssError(f"fp32 tensor expected, got {t.dtype}")


def empty_act(shape, device, ld_align=LD_ALIGN):
    """[..., M] fp32 activation whose rows are padded to a multiple of `ld_align` floats."""
    M = shape[-1]
    ld = (M + ld_align - 1) // ld_align * ld_align
    buf = torch.empty(*shape[:-1], ld, device=device, dtype=torch.float32)
    return buf[..., :M] if ld != M else buf


def rowmat(t):
    """(rows, cols, ld) of a tensor that is a row matrix with unit column stride, else None.
    Size-1 dims are ignored; the first non-unit dim above the last one defines the row stride."""
    if t.dim() == 0:
        return None
    cols = t.shape[-1]
    if cols > 1 and t.stride(-1) != 1:
        return None
    rows, ld = 1, None
    for d in range(t.dim() - 2, -1, -1):
        n = t.shape[d]
        if n == 1:
            continue
        if ld is None:
            ld = t.stride(d)
            if ld < cols:
                return None
        elif t.stride(d) != ld * rows:
            return None
        rows *= n
    if ld is None:
        # a single row: any row stride >= cols describes it; keep the buffer's own (padded, 16-B aligned) one when the tensor is a view
        # of a padded buffer -- the C side checks row alignment even for one row (a [1, 1, M] output of the tiny HTDemucs's DConv)
        ld = t.stride(-2) if t.dim() >= 2 and t.stride(-2) >= cols else max(cols, 1)
    return rows, cols, ld


def _rowmat_collapsed(t):
    """(rows, cols, ld) with several trailing dims merged into the columns: a view whose last k dims are dense and whose leading dims
    share one stride -- a crop along an inner dim of a channel-first tensor, [B, C, F', T] out of [B, C, F, T] -- is a row matrix of
    B * C rows of F' * T columns with row stride F * T; else None"""
    nd = t.dim()
    for k in range(nd - 2, 0, -1):            # columns = dims k .. nd-1
        cols, dense = 1, True
        for d in range(nd - 1, k - 1, -1):
            if t.shape[d] != 1 and t.stride(d) != cols:
                dense = False
                break
            cols *= t.shape[d]
        if not dense:
            return None
        rows, ld, ok = 1, None, True
        for d in range(k - 1, -1, -1):
            n = t.shape[d]
            if n == 1:
                continue
            if ld is None:
                ld = t.stride(d)
                if ld < cols:
                    ok = False
                    break
            elif t.stride(d) != ld * rows:
                ok = False
                break
            rows *= n
        if ok and ld is not None:
            return rows, cols, ld
    return None


def rowmat_collapsed_ok(t):
    """a view the element-wise kernels read in place as a row matrix (as_rowmat): 16-B aligned rows of dense trailing dims"""
    rm = _rowmat_collapsed(t) if ROWMAT_COLLAPSE else None
    return rm is not None and rm[2] % 4 == 0 and t.data_ptr() % 16 == 0 and t.dtype == torch.float32


ROWMAT_COLLAPSE = os.environ.get("FQSS_ROWMAT_COLLAPSE", "1") != "0"    # element-wise kernels take cropped channel-first views without a dense copy


def as_rowmat(t):
    """return (tensor, rows, cols, ld); copies into a padded buffer only if the layout is foreign"""
    rm = rowmat(t)
    if rm is None:
        c = empty_act(tuple(t.shape), t.device)
        c.copy_(t)
        t, rm = c, rowmat(c)
    return (t,) + rm


# ------------------------------------------------------------------ K1 / K3
def empty_codes(shape, device):
    """u8 activation codes [..., M] with rows padded to 16 B (the q-GEMM staging loads 16 codes/lane)"""
    M = shape[-1]
    ld = (M + 15) // 16 * 16
    buf = torch.empty(*shape[:-1], ld, device=device, dtype=torch.uint8)
    return buf[..., :M] if ld != M else buf


def actq_fwd(z, act, slope, qmode, qmin, qmax, obs_ws, want_idx=False, dense_idx=False, write_out=True):
    """write_out=False (QUANT + want_idx only): the fp32 result is an UNINITIALISED carrier, only codes are written"""
    _need_gpu(z, slope, qmin, qmax)
    rc = _rowmat_collapsed(z) if (ROWMAT_COLLAPSE and z.dim() >= 3 and rowmat(z) is None) else None
    skip_out = (not write_out) and want_idx and qmode == Q_QUANT
    if rc is not None:
        # a cropped channel-first view ([B, C, F', T] out of [B, C, F, T]): rows of F' T columns at row stride F T, no dense copy; the
        # outputs are dense, un-padded tensors of z's shape (the same rows / columns with ld = columns)
        rows, cols, ld_z = rc
        out = torch.empty(z.shape, device=z.device, dtype=torch.float32)
        ld_o = ld_i = cols
        idx = torch.empty(z.shape, device=z.device, dtype=torch.uint8) if want_idx else None
    else:
        z, rows, cols, ld_z = as_rowmat(z)
        out = empty_act(tuple(z.shape), z.device)
        _, _, _, ld_o = (out,) + rowmat(out)
        idx, ld_i = None, cols
        if want_idx:
            idx = torch.empty(z.shape, device=z.device, dtype=torch.uint8) if dense_idx else empty_codes(tuple(z.shape), z.device)
            ld_i = rowmat(idx)[2]
    _lib.call("fqss_actq_fwd", _p(z), None if skip_out else _p(out), _p(idx), rows, cols, ld_z, ld_o, ld_i, act,
              _p(slope), qmode, _p(qmin), _p(qmax), _p(obs_ws), _stream())
    return (out, idx) if want_idx else out


def obs_reset(obs_ws):
    _lib.call("fqss_obs_reset", _p(obs_ws), obs_ws.numel() // 2, _stream())


def observer_ema(qmin, qmax, obs_ws, alpha=0.9):
    _lib.call("fqss_observer_ema", _p(qmin), _p(qmax), _p(obs_ws), float(alpha), _stream())


def minmax(x, obs_ws):
    _need_gpu(x)
    x, rows, cols, ld = as_rowmat(x)
    _lib.call("fqss_minmax", _p(x), rows, cols, ld, _p(obs_ws), _stream())


def actq_bwd(z, g, act, slope, qmode, qmin, qmax, gacc, gbias=None, C=0, out=None):
    """out: optional row-matrix view (e.g. a column block of a wider buffer) that receives gz"""
    _need_gpu(z, g)
    rcz = rcg = None
    if ROWMAT_COLLAPSE and gbias is None and out is None and z.dim() >= 3 and z.shape == g.shape and (rowmat(z) is None or rowmat(g) is None):
        rcz, rcg = _rowmat_collapsed(z), _rowmat_collapsed(g)       # (see actq_fwd: cropped channel-first views, no dense copy)
    if rcz is not None and rcg is not None and rcz[:2] == rcg[:2]:
        (rows, cols, ld_z), ld_g = rcz, rcg[2]
        gz = torch.empty(z.shape, device=z.device, dtype=torch.float32)
        ld_gz = cols
    else:
        z, rows, cols, ld_z = as_rowmat(z)
        g, r2, c2, ld_g = as_rowmat(g)
        assert (rows, cols) == (r2, c2), "actq_bwd: z/g shape mismatch"
        gz = empty_act(tuple(z.shape), z.device) if out is None else out
        rm = rowmat(gz)
        assert rm is not None and rm[:2] == (rows, cols), "actq_bwd: `out` must be a row-matrix view of z's shape"
        ld_gz = rm[2]
    _lib.call("fqss_actq_bwd", _p(z), _p(g), _p(gz), rows, cols, ld_z, ld_g, ld_gz, act, _p(slope), qmode,
              _p(qmin), _p(qmax), _p(gacc), _p(gbias), C, _stream())
    return gz


def actq_bwd_colbias(z, g, act, slope, qmode, qmin, qmax, gacc, gbias):
    """actq_bwd over a row linear's output [..., F] that also adds the bias gradient (column sums of gz) into gbias [F]"""
    _need_gpu(z, g, gbias)
    F = z.shape[-1]
    z, R, ld_z = _rows(z, F)
    g, R2, ld_g = _rows(g, F)
    assert R == R2, "actq_bwd_colbias: z/g shape mismatch"
    gz = torch.empty(*z.shape, device=z.device, dtype=torch.float32)
    _lib.call("fqss_actq_bwd_colbias", _p(z), _p(g), _p(gz), R, F, ld_z, ld_g, F, act, _p(slope), qmode, _p(qmin), _p(qmax), _p(gacc),
              _p(gbias), _stream())
    return gz


def actq2_bwd_colbias(z, g, qmin1, qmax1, qmin2, qmax2, gacc1, gacc2, gbias):
    """backward of qrow_fwdq2's epilogue (NlQ(ReLU) behind a LinearQ's quantizer) in one pass: gz + bias column sums + both range partials"""
    _need_gpu(z, g, gbias)
    F = z.shape[-1]
    z, R, ld_z = _rows(z, F)
    g, R2, ld_g = _rows(g, F)
    assert R == R2, "actq2_bwd_colbias: z/g shape mismatch"
    gz = torch.empty(*z.shape, device=z.device, dtype=torch.float32)
    _lib.call("fqss_actq2_bwd_colbias", _p(z), _p(g), _p(gz), R, F, ld_z, ld_g, F, _p(qmin1), _p(qmax1), _p(qmin2), _p(qmax2), _p(gacc1), _p(gacc2),
              _p(gbias), _stream())
    return gz


def colbias_ok(F):
    return F % 4 == 0 and F <= 64 * 2048


# ------------------------------------------------------------------ K2
def _w_layout(shape, axis):
    outer = 1
    for d in shape[:axis]:
        outer *= d
    inner = 1
    for d in shape[axis + 1:]:
        inner *= d
    return outer, shape[axis], inner


def wq_observe(w, axis, qmin, qmax):
    _need_gpu(w, qmin, qmax)
    o, c, i = _w_layout(w.shape, axis)
    _lib.call("fqss_wq_observe", _p(w.contiguous()), o, c, i, _p(qmin), _p(qmax), _stream())


def wq_fwd(w, axis, qmin, qmax, want_idx=False):
    _need_gpu(w, qmin, qmax)
    w = w.contiguous()
    o, c, i = _w_layout(w.shape, axis)
    wq = torch.empty_like(w)
    idx = torch.empty(w.shape, device=w.device, dtype=torch.int8) if want_idx else None
    _lib.call("fqss_wq_fwd", _p(w), _p(wq), _p(idx), o, c, i, _p(qmin), _p(qmax), _stream())
    return (wq, idx) if want_idx else wq


def wq_bwd(w, g, axis, qmin, qmax, out=None):
    """out=None: fresh (gw, gmin, gmax); out=(gw, gmin, gmax): accumulate (+=) into the given buffers"""
    _need_gpu(w, g, qmin, qmax)
    w, g = w.contiguous(), g.contiguous()
    o, c, i = _w_layout(w.shape, axis)
    if out is None:
        gw, gmin, gmax, acc = torch.empty_like(w), torch.empty_like(qmin), torch.empty_like(qmax), 0
    else:
        (gw, gmin, gmax), acc = out, 1
        assert gw.is_contiguous() and gmin.is_contiguous() and gmax.is_contiguous()
    _lib.call("fqss_wq_bwd", _p(w), _p(g), _p(gw), _p(gmin), _p(gmax), o, c, i, _p(qmin), _p(qmax), acc, _stream())
    return gw, gmin, gmax


def gacc_flush_multi(table):
    _lib.call("fqss_gacc_flush_multi", _p(table), table.shape[0], _stream())


def wq_multi_fwd(table, total_channels):
    _lib.call("fqss_wq_multi_fwd", _p(table), table.shape[0], total_channels, _stream())


def wq_multi_bwd(table, total_channels):
    _lib.call("fqss_wq_multi_bwd", _p(table), table.shape[0], total_channels, _stream())


def gacc_flush(gacc, gmin, gmax, gslope):
    _lib.call("fqss_gacc_flush", _p(gacc), _p(gmin), _p(gmax), _p(gslope), _stream())


# ------------------------------------------------------------------ K4 / K5  pointwise conv
def _bcm(t):
    """[B][C][M] tensor -> (tensor, B, C, M, ld) with batch stride C*ld"""
    assert t.dim() == 3, "expected [B, C, M]"
    t, rows, cols, ld = as_rowmat(t)
    return t, t.shape[0], t.shape[1], t.shape[2], ld


def pwconv_fwd(x, w, bias, six=False):
    """six: the six-product split GEMM (fqss_pwconv_fwd_x3s) instead of the nine exact products"""
    _need_gpu(x, w, bias)
    x, B, Ci, M, ld_x = _bcm(x)
    Co = w.shape[0]
    assert w.is_contiguous() and w.numel() == Co * Ci
    z = empty_act((B, Co, M), x.device)
    # bf16-MFMA 3x3 exact split when rows are 16-B aligned, fp32-MFMA kernel otherwise
    fn = "fqss_pwconv_fwd" if not (USE_X3 and Ci % 4 == 0 and ld_x % 4 == 0 and x.data_ptr() % 16 == 0) else \
        "fqss_pwconv_fwd_x3s" if six else "fqss_pwconv_fwd_x3"
    _lib.call(fn, _p(x), _p(w), _p(bias), _p(z), B, Ci, Co, M, ld_x, rowmat(z)[2], _stream())
    return z


def pwconv_fwd_wq(x, wc, bias):
    """pointwise conv of a FLOAT input with a fake-quantized weight given as its int8 codes (WCodes: idx [Co][Ci], dw [Co]); None when the
    operands do not meet the kernel's alignment rules (the caller then runs the float x float form)"""
    _need_gpu(x, bias)
    x, B, Ci, M, ld_x = _bcm(x)
    if Ci != wc.Ci or Ci % 16 or Ci < 16 or ld_x % 4 or x.data_ptr() % 16 or not wc.idx.is_contiguous():
        return None
    z = empty_act((B, wc.Co, M), x.device)
    _lib.call("fqss_pwconv_fwd_wq", _p(x), _p(wc.idx), _p(wc.dw), _p(bias), _p(z), B, Ci, wc.Co, M, ld_x, rowmat(z)[2], _stream())
    return z


CONV_IMPLICIT = os.environ.get("FQSS_CONV_IMPLICIT", "1") != "0"
CONV_IMPLICIT_MAX_CO = int(os.environ.get("FQSS_CONV_IMPLICIT_MAX_CO", "64"))


def conv1d_s1_ok(Ci, taps):
    """stride-1 1-D convolutions run without a frame image (fqss_conv1d_s1_*) when the weight rows are 16-B aligned"""
    return CONV_IMPLICIT and taps > 1 and (Ci * taps) % 4 == 0


def conv1d_s1_fwd(x, w2, bias, taps, dil, pad):
    """x [B, Ci, M], w2 [Co, Ci * taps] (flattened conv weight; row stride a multiple of 4: a padded view is fine) -> z [B, Co, Mo],
    Mo = M + 2 pad - dil (taps - 1)"""
    _need_gpu(x, w2, bias)
    x, B, Ci, M, ld_x = _bcm(x)
    Co = w2.shape[0]
    assert w2.dim() == 2 and w2.shape[1] == Ci * taps and w2.stride(1) == 1
    if w2.stride(0) % 4 != 0 or w2.data_ptr() % 16 != 0:       # rows to 16 B (a tiny weight: Co x Ci*taps)
        wp = torch.zeros(Co, (Ci * taps + 3) // 4 * 4, device=w2.device, dtype=torch.float32)
        wp[:, :Ci * taps] = w2
        w2 = wp[:, :Ci * taps]
    Mo = M + 2 * pad - dil * (taps - 1)
    z = empty_act((B, Co, Mo), x.device)
    _lib.call("fqss_conv1d_s1_fwd", _p(x), _p(w2), _p(bias), _p(z), B, Ci, Co, M, Mo, taps, dil, pad, ld_x, w2.stride(0), rowmat(z)[2], _stream())
    return z


def conv1d_s1_bwd_w(gz, x, gw, taps, dil, pad):
    """gw [Co, Ci * taps] (accumulator) += the weight gradient of conv1d_s1_fwd"""
    _need_gpu(gz, x, gw)
    gz, B, Co, Mo, ld_gz = _bcm(gz)
    x, _, Ci, M, ld_x = _bcm(x)
    assert gw.is_contiguous() and gw.numel() == Co * Ci * taps and Mo == M + 2 * pad - dil * (taps - 1)
    _lib.call("fqss_conv1d_s1_bwd_w", _p(gz), _p(x), _p(gw), B, Ci, Co, M, Mo, taps, dil, pad, ld_gz, ld_x, _stream())


def pwconv_bwd_x(gz, w, Ci):
    _need_gpu(gz, w)
    gz, B, Co, M, ld_gz = _bcm(gz)
    gx = empty_act((B, Ci, M), gz.device)
    _lib.call("fqss_pwconv_bwd_x", _p(gz), _p(w), _p(gx), B, Ci, Co, M, ld_gz, rowmat(gx)[2], _stream())
    return gx


def pwconv_bwd_w(gz, x, gw):
    """gw[Co][Ci] += sum_b gz[b] x[b]^T (gw: caller-zeroed accumulator)"""
    _need_gpu(gz, x, gw)
    gz, B, Co, M, ld_gz = _bcm(gz)
    x, _, Ci, _, ld_x = _bcm(x)
    assert gw.is_contiguous() and gw.numel() == Co * Ci
    _lib.call("fqss_pwconv_bwd_w", _p(gz), _p(x), _p(gw), B, Ci, Co, M, ld_gz, ld_x, _stream())


# ------------------------------------------------------------------ K4q / K5q  grid-valued pointwise conv (bf16 MFMA)
class WCodes:
    """int8 codes of a fake-quantized pointwise weight: idx [Co][Ci], idxT [Ci][Co], dw[Co], rw[Co]"""
    __slots__ = ("idx", "idxT", "dw", "rw", "Co", "Ci")


def wq_codes(w, qmin, qmax):
    _need_gpu(w, qmin, qmax)
    Co, Ci = w.shape[0], w.shape[1]
    assert w.numel() == Co * Ci and w.is_contiguous()
    c = WCodes()
    c.Co, c.Ci = Co, Ci
    c.idx = torch.empty(Co, Ci, device=w.device, dtype=torch.int8)
    c.idxT = torch.empty(Ci, Co, device=w.device, dtype=torch.int8)
    c.dw = torch.empty(Co, device=w.device, dtype=torch.float32)
    c.rw = torch.empty(Co, device=w.device, dtype=torch.float32)
    _lib.call("fqss_wq_codes", _p(w), _p(c.idx), _p(c.idxT), _p(c.dw), _p(c.rw), Co, Ci, _p(qmin), _p(qmax), _stream())
    return c


def q_eligible(Ci, Co):
    return _lib.BACKEND != "cpu" and Ci % 16 == 0 and Co % 16 == 0 and Ci <= 512


def qpw_fwd(xc, wc, bias, qmin_x, qmax_x):
    """xc: u8 codes [B,Ci,M] (padded rows) -> z fp32 [B,Co,M]"""
    B, Ci, M = xc.shape
    rm = rowmat(xc)
    assert rm is not None and rm[2] % 16 == 0
    z = empty_act((B, wc.Co, M), xc.device)
    _lib.call("fqss_qpw_fwd", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias), _p(qmin_x), _p(qmax_x), _p(z),
              B, Ci, wc.Co, M, rm[2], rowmat(z)[2], _stream())
    return z


def qpw_bwd_x(gz, wc, add=None):
    """gx = W_q^T gz (+ add: a tensor of gx's shape summed in the epilogue -- the gradient of the other branch of a residual fork)"""
    assert wc.idxT is not None, "this layer's codes live in a concatenated pair image: use qpw_bwd_x2"
    gz, B, Co, M, ld_gz = _bcm(gz)
    if ld_gz % 4 != 0 or gz.data_ptr() % 16 != 0:
        c = empty_act(tuple(gz.shape), gz.device)
        c.copy_(gz)
        gz, ld_gz = c, rowmat(c)[2]
    gx = empty_act((B, wc.Ci, M), gz.device)
    if add is not None:
        assert tuple(add.shape) == (B, wc.Ci, M)
        add, ld_add = _aligned_grad(add)
        _lib.call("fqss_qpw_bwd_x_add", _p(gz), _p(wc.idxT), _p(wc.dw), _p(add), _p(gx), B, wc.Ci, Co, M, ld_gz, ld_add, rowmat(gx)[2], _stream())
    else:
        _lib.call("fqss_qpw_bwd_x", _p(gz), _p(wc.idxT), _p(wc.dw), _p(gx), B, wc.Ci, Co, M, ld_gz, rowmat(gx)[2], _stream())
    return gx


def qpw_bwd_w(gz, xc, qmin_x, qmax_x, gw):
    gz, B, Co, M, ld_gz = _bcm(gz)
    if ld_gz % 4 != 0 or gz.data_ptr() % 16 != 0:
        c = empty_act(tuple(gz.shape), gz.device)
        c.copy_(gz)
        gz, ld_gz = c, rowmat(c)[2]
    Ci = xc.shape[1]
    _lib.call("fqss_qpw_bwd_w", _p(gz), _p(xc), _p(qmin_x), _p(qmax_x), _p(gw), B, Ci, Co, M, ld_gz, rowmat(xc)[2], _stream())


# ------------------------------------------------------------------ K6q / K7q  codes-only layers (csrc/fused_q.hip)
def qpw_fwd2(xc, wc, bias1, bias2, qmin_x, qmax_x, Co1):
    """two layers on the same coded input as one GEMM over concatenated channels -> (z1 [B,Co1,M], z2 [B,Co-Co1,M])"""
    B, Ci, M = xc.shape
    rm = rowmat(xc)
    assert rm is not None and Ci == wc.Ci and 0 < Co1 < wc.Co
    z1, z2 = empty_act((B, Co1, M), xc.device), empty_act((B, wc.Co - Co1, M), xc.device)
    _lib.call("fqss_qpw_fwd2", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias1), _p(bias2), _p(qmin_x), _p(qmax_x), _p(z1), _p(z2),
              B, Ci, Co1, wc.Co - Co1, M, rm[2], rowmat(z1)[2], rowmat(z2)[2], _stream())
    return z1, z2


def qpw_fwdq(xc, wc, bias1, bias2, qmin_x, qmax_x, Co1, act, slope, r1, r2=None, stats=None, adds=None):
    """q-GEMM forward with the output quantizer(s) fused: -> (z1, yc1) or (z1, z2, yc1, yc2) for a pair (r = (qmin, qmax));
    stats (single output only): a CodeStats from new_stats("qpw", ...) that receives the integer statistics of yc1.
    adds (no activation): (add1 | None, add2 | None), add = (a_codes, amin, amax, qmin, qmax) -- the AddQ that alone consumes that
    output runs in the epilogue (fqss_qpw_fwdq_add); its sum codes are appended to the result as a pair (s1 | None, s2 | None)"""
    B, Ci, M = xc.shape
    rm = rowmat(xc)
    Co2 = wc.Co - Co1
    assert rm is not None and Ci == wc.Ci and Co2 >= 0 and (Co2 == 0) == (r2 is None)
    z1, yc1 = empty_act((B, Co1, M), xc.device), empty_codes((B, Co1, M), xc.device)
    z2 = yc2 = None
    if Co2:
        z2, yc2 = empty_act((B, Co2, M), xc.device), empty_codes((B, Co2, M), xc.device)
    if adds is not None and any(a is not None for a in adds):
        assert act == ACT_NONE and stats is None and (adds[1] is None or Co2)
        sums, recs = [], []
        for a, Co in zip(adds, (Co1, Co2)):
            if a is None:
                sums.append(None); recs.append(None)
                continue
            ac, amin, amax, qmin, qmax = a
            assert tuple(ac.shape) == (B, Co, M) and rowmat(ac) is not None
            sc = empty_codes((B, Co, M), xc.device)
            r = _lib.FqssAddAfter()
            r.a, r.ld_a, r.amin, r.amax, r.qmin, r.qmax, r.y, r.ld_y = _p(ac), rowmat(ac)[2], _p(amin), _p(amax), _p(qmin), _p(qmax), _p(sc), rowmat(sc)[2]
            sums.append(sc); recs.append(r)
        _lib.call("fqss_qpw_fwdq_add", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias1), _p(bias2), _p(qmin_x), _p(qmax_x), _p(z1), _p(z2),
                  _p(r1[0]), _p(r1[1]), _p(r2[0]) if r2 else None, _p(r2[1]) if r2 else None, _p(yc1), _p(yc2),
                  B, Ci, Co1, Co2, M, rm[2], rowmat(z1)[2], rowmat(z2)[2] if Co2 else 0, rowmat(yc1)[2], rowmat(yc2)[2] if Co2 else 0,
                  _ref(recs[0]) if recs[0] is not None else None, _ref(recs[1]) if recs[1] is not None else None, _stream())
        return ((z1, z2, yc1, yc2) if Co2 else (z1, yc1)) + (tuple(sums),)
    if DESC_ABI:
        w = _lib.FqssWCodes()
        w.idx, w.idxT, w.dw, w.rw, w.Co, w.Ci = _p(wc.idx), _p(wc.idxT), _p(wc.dw), _p(wc.rw), wc.Co, wc.Ci
        dx, dz1, dz2, dy1, dy2 = _desc(xc), _desc(z1), _desc(z2), _desc(yc1), _desc(yc2)
        qx, q1 = _qparams(qmin_x, qmax_x), _qparams(r1[0], r1[1], act, slope)
        q2 = _qparams(r2[0], r2[1], act, slope) if r2 else None
        ws, nws = (_p(stats.ws), stats.ws.numel() * stats.ws.element_size()) if stats is not None else (None, 0)
        _lib.call("fqss_pwconv_fq_fwd", _ref(dx), _ref(qx), _ref(w), _p(bias1), _p(bias2), _ref(dz1), _ref(dz2), _ref(dy1), _ref(dy2),
                  _ref(q1), _ref(q2), ws, nws, _stream())
        return (z1, z2, yc1, yc2) if Co2 else (z1, yc1)
    _lib.call("fqss_qpw_fwdq", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias1), _p(bias2), _p(qmin_x), _p(qmax_x), _p(z1), _p(z2),
              act, _p(slope), _p(r1[0]), _p(r1[1]), _p(r2[0]) if r2 else None, _p(r2[1]) if r2 else None, _p(yc1), _p(yc2),
              B, Ci, Co1, Co2, M, rm[2], rowmat(z1)[2], rowmat(z2)[2] if Co2 else 0, rowmat(yc1)[2], rowmat(yc2)[2] if Co2 else 0,
              _p(stats.ws) if stats is not None else None, _stream())
    return (z1, z2, yc1, yc2) if Co2 else (z1, yc1)


def qpw_bwd_x2(gz1, gz2, wc):
    """gx = W1q^T gz1 + W2q^T gz2 in one GEMM (K = Co1 + Co2)"""
    gz1, ld1 = _aligned_grad(gz1)
    gz2, ld2 = _aligned_grad(gz2)
    B, Co1, M = gz1.shape
    Co2 = gz2.shape[1]
    assert Co1 + Co2 == wc.Co and gz2.shape[0] == B and gz2.shape[2] == M
    gx = empty_act((B, wc.Ci, M), gz1.device)
    _lib.call("fqss_qpw_bwd_x2", _p(gz1), _p(gz2), _p(wc.idxT), _p(wc.dw), _p(gx), B, wc.Ci, Co1, Co2, M, ld1, ld2, rowmat(gx)[2], _stream())
    return gx


def qpw_bwd_w2(gz1, gz2, xc, qmin_x, qmax_x, gw):
    """gw [Co1+Co2, Ci] += [gz1; gz2] x^T"""
    gz1, ld1 = _aligned_grad(gz1)
    gz2, ld2 = _aligned_grad(gz2)
    B, Co1, M = gz1.shape
    Co2 = gz2.shape[1]
    Ci = xc.shape[1]
    assert gw.is_contiguous() and gw.numel() == (Co1 + Co2) * Ci
    _lib.call("fqss_qpw_bwd_w2", _p(gz1), _p(gz2), _p(xc), _p(qmin_x), _p(qmax_x), _p(gw), B, Ci, Co1, Co2, M, ld1, ld2, rowmat(xc)[2],
              _stream())


class WgradQueue:
    """Weight-gradient launches of quantized 1x1 convolutions, queued during a backward segment and run together by ONE grouped
    launch per <= 25 layers (fqss_qpw_bwd_w_group: no float atomics, bit-reproducible): their results feed nothing but the
    optimizer.  The queue keeps gz / codes alive until flush(); the workspace (arrival tickets + slab slots) belongs to the queue and
    is allocated once -- flush() of a later step (and of a hipGraph capture) finds it in place.
    Memory: holding every gz of a backward segment until its flush costs the eager cfg-2 step ~2 GB of peak memory with one segment
    (50 x 8 x 512 | 128 x 4000 fp32 gradients; 0.5 GB per segment with the four data-parallel segments); inside a captured graph the
    pool reuses it across replays.  On a 288-GB part that buys the two-launch form; a caller short of memory flushes more often."""

    def __init__(self):
        self.jobs, self.ws = [], None

    def push(self, gz1, gz2, xc, qmin_x, qmax_x, gw):
        gz1, ld1 = _aligned_grad(gz1)
        ld2 = 0
        if gz2 is not None:
            gz2, ld2 = _aligned_grad(gz2)
        assert gw.is_contiguous()
        if any(j[5].data_ptr() == gw.data_ptr() for j in self.jobs):
            # a layer applied twice in one backward: the grouped launch owns each gw exclusively (no atomics), so the second
            # contribution takes the per-layer kernel right away (its adds are atomic and commute with the queued launch)
            if gz2 is not None:
                qpw_bwd_w2(gz1, gz2, xc, qmin_x, qmax_x, gw)
            else:
                qpw_bwd_w(gz1, xc, qmin_x, qmax_x, gw)
            return
        self.jobs.append((gz1, gz2, xc, qmin_x, qmax_x, gw, ld1, ld2))

    def flush(self):
        if not self.jobs:
            return
        n = len(self.jobs)
        arr = (_lib.FqssWgradJob * n)()
        for j, (gz1, gz2, xc, lo, hi, gw, ld1, ld2) in zip(arr, self.jobs):
            B, Co1, M = gz1.shape
            j.gz1, j.gz2, j.xc, j.qmin_x, j.qmax_x, j.gw = _p(gz1), _p(gz2), _p(xc), _p(lo), _p(hi), _p(gw)
            j.B, j.Ci, j.Co1, j.Co2, j.M = B, xc.shape[1], Co1, (gz2.shape[1] if gz2 is not None else 0), M
            j.ld_gz1, j.ld_gz2, j.ld_xc = ld1, ld2, rowmat(xc)[2]
            assert gw.numel() == (j.Co1 + j.Co2) * j.Ci
        need = _lib.query("fqss_qpw_bwd_w_group_ws", arr, n)
        if need < 0:
            raise _lib.FqssError("fqss_qpw_bwd_w_group_ws: " + _lib.load().fqss_last_error().decode())
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.zeros(need, dtype=torch.uint8, device=self.jobs[0][0].device)      # tickets start at zero; every launch leaves them zero
        _lib.call("fqss_qpw_bwd_w_group", arr, n, _p(self.ws), self.ws.numel(), _stream())
        self.jobs = []


def _codes3(xc):
    """u8 codes [B,C,M] with 16-B aligned rows -> (B, C, M, ld)"""
    rm = rowmat(xc)
    assert xc.dim() == 3 and rm is not None and rm[2] % 16 == 0 and xc.data_ptr() % 16 == 0, "bad code layout"
    return xc.shape[0], xc.shape[1], xc.shape[2], rm[2]


def decode(xc, qmin, qmax):
    rm = rowmat(xc)
    out = empty_act(tuple(xc.shape), xc.device)
    _lib.call("fqss_decode", _p(xc), _p(out), rm[0], rm[1], rm[2], rowmat(out)[2], _p(qmin), _p(qmax), _stream())
    return out


class CodeStats:
    """exact integer statistics (sum c, sum c^2) of a coded activation as [B][nslots][2] int64 partial sums, emitted by the
    epilogue of the kernel that produced the codes (fqss_qpw_fwdq, fqss_dwq_fwd) for a consuming GroupNormQ"""
    __slots__ = ("ws", "nslots")

    def __init__(self, ws, nslots):
        self.ws, self.nslots = ws, nslots


_STAT_SLOTS = {}


def stat_slots(kind, C, M):
    key = (kind, C, M)
    v = _STAT_SLOTS.get(key)
    if v is None:
        v = _STAT_SLOTS[key] = _lib.query("fqss_qpw_stat_slots" if kind == "qpw" else "fqss_dwq_stat_slots", C, M)
    return v


def new_stats(kind, B, C, M, device):
    """CodeStats buffer for the codes [B, C, M] a q-GEMM ("qpw") / depthwise layer ("dwq") is about to produce, or None"""
    n = stat_slots(kind, C, M)
    return CodeStats(torch.empty(B * n * 2, device=device, dtype=torch.int64), n) if n else None


def gnq_fwd(xc, qmin_x, qmax_x, gamma, beta, eps, qmin, qmax, write_out, stats=None):
    B, C, M, ld_xc = _codes3(xc)
    yc = empty_codes((B, C, M), xc.device)
    out = empty_act((B, C, M), xc.device)      # carrier; written only when write_out
    mean_rstd = torch.empty(B, 2, device=xc.device, dtype=torch.float32)
    ws = torch.empty(2 * 64 * B, device=xc.device, dtype=torch.int64) if stats is None else None
    _lib.call("fqss_gnq_fwd", _p(xc), _p(qmin_x), _p(qmax_x), _p(gamma), _p(beta), _p(yc), _p(out) if write_out else None,
              _p(mean_rstd), B, C, M, ld_xc, rowmat(yc)[2], rowmat(out)[2], float(eps), _p(qmin), _p(qmax), _p(ws),
              _p(stats.ws) if stats is not None else None, stats.nslots if stats is not None else 0, _stream())
    return out, yc, mean_rstd


def gnq_fwd_deferred(xc):
    """the buffers fqss_gnq_fwd would fill -- (carrier, codes, mean_rstd) -- WITHOUT the launch: the depthwise layer behind the GroupNormQ
    runs both layers as one kernel (gndwq_fwd) and fills them; anything else that reads them first must launch gnq_fwd_into"""
    B, C, M, _ = _codes3(xc)
    return (empty_act((B, C, M), xc.device), empty_codes((B, C, M), xc.device),
            torch.empty(B, 2, device=xc.device, dtype=torch.float32))


def gnq_fwd_into(xc, qmin_x, qmax_x, gamma, beta, eps, qmin, qmax, stats, yc, mean_rstd):
    """fqss_gnq_fwd into buffers allocated by gnq_fwd_deferred (codes only: the carrier stays a carrier)"""
    B, C, M, ld_xc = _codes3(xc)
    _lib.call("fqss_gnq_fwd", _p(xc), _p(qmin_x), _p(qmax_x), _p(gamma), _p(beta), _p(yc), None, _p(mean_rstd), B, C, M, ld_xc,
              rowmat(yc)[2], 0, float(eps), _p(qmin), _p(qmax), None, _p(stats.ws), stats.nslots, _stream())


def gndwq_fwd(d, w, bias, dil, pad, act, slope, qmin2, qmax2, want_stats2):
    """GroupNormQ (deferred record d of ops.GroupNormActQ) + depthwise Conv1dNlQ in ONE launch -> (carrier, codes, CodeStats or None) of
    the depthwise layer; fills the GroupNorm's codes and mean / rstd on the way"""
    xc = d["xc"]
    B, C, M, ld_xc = _codes3(xc)
    y2 = empty_codes((B, C, M), xc.device)
    out2 = empty_act((B, C, M), xc.device)
    st2 = CodeStats(torch.empty(B * C * 2, device=xc.device, dtype=torch.int64), C) if want_stats2 and C <= 1024 else None
    _lib.call("fqss_gndwq_fwd", _p(xc), _p(d["qmin_x"]), _p(d["qmax_x"]), _p(d["gamma"]), _p(d["beta"]), float(d["eps"]), _p(d["stats"].ws),
              d["stats"].nslots, _p(d["mean_rstd"]), _p(d["yc"]), _p(d["qmin"]), _p(d["qmax"]), _p(w), _p(bias), dil, pad, act, _p(slope),
              _p(y2), _p(qmin2), _p(qmax2), _p(st2.ws) if st2 is not None else None, B, C, M, ld_xc, rowmat(d["yc"])[2], rowmat(y2)[2],
              _stream())
    return out2, y2, st2


def _aligned_grad(g):
    g, _, _, ld = as_rowmat(g)
    if ld % 4 != 0 or g.data_ptr() % 16 != 0:
        c = empty_act(tuple(g.shape), g.device)
        c.copy_(g)
        g, ld = c, rowmat(c)[2]
    return g, ld


def gnq_bwd(xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, qmin, qmax, gacc, ggamma, gbeta, producer=None):
    """producer = (z, act, slope, gacc, gbias) of the layer that made xc: the result is then THAT layer's gz (its
    output-quantizer / non-linearity backward applied on the fly, its partials accumulated) instead of gx"""
    B, C, M, ld_xc = _codes3(xc)
    g, ld_g = _aligned_grad(g)
    gx = empty_act((B, C, M), xc.device)
    ws = torch.empty(2 * B * C + 2 * B, device=xc.device, dtype=torch.float64)
    if producer is None:
        _lib.call("fqss_gnq_bwd", _p(xc), _p(qmin_x), _p(qmax_x), _p(g), _p(gamma), _p(beta), _p(mean_rstd), _p(gx),
                  _p(ggamma), _p(gbeta), B, C, M, ld_xc, ld_g, rowmat(gx)[2], _p(qmin), _p(qmax), _p(gacc), _p(ws), _stream())
    else:
        pz, pact, pslope, pgacc, pgbias = producer
        rz = rowmat(pz)
        assert tuple(pz.shape) == (B, C, M) and rz is not None
        _lib.call("fqss_gnq_bwd_p", _p(xc), _p(qmin_x), _p(qmax_x), _p(g), _p(gamma), _p(beta), _p(mean_rstd), _p(gx),
                  _p(ggamma), _p(gbeta), B, C, M, ld_xc, ld_g, rowmat(gx)[2], _p(qmin), _p(qmax), _p(gacc), _p(ws),
                  _p(pz), rz[2], pact, _p(pslope), _p(pgacc), _p(pgbias), _stream())
    return gx


def gnq_bwd_rows(xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, qmin, qmax, gacc):
    """first pass of gnq_bwd alone -> (g as the kernels read it, ws [B*C][2] doubles): the depthwise layer in front of this GroupNormQ
    runs the second pass on load (dwq_bwd(after=...))"""
    B, C, M, ld_xc = _codes3(xc)
    g, ld_g = _aligned_grad(g)
    ws = torch.empty(2 * B * C, device=xc.device, dtype=torch.float64)
    _lib.call("fqss_gnq_bwd_rows", _p(xc), _p(qmin_x), _p(qmax_x), _p(g), _p(gamma), _p(beta), _p(mean_rstd), B, C, M, ld_xc, ld_g,
              _p(qmin), _p(qmax), _p(gacc), _p(ws), _stream())
    return g, ws


def gnq_bwd_apply(xc, qmin_x, qmax_x, g, gamma, beta, mean_rstd, qmin, qmax, ws, ggamma, gbeta, producer=None):
    """second pass of gnq_bwd alone, the row sums `ws` coming from the depthwise layer behind this GroupNormQ (dwq_bwd(before=...))"""
    B, C, M, ld_xc = _codes3(xc)
    g, ld_g = _aligned_grad(g)
    gx = empty_act((B, C, M), xc.device)
    pz, pact, pslope, pgacc, pgbias = producer if producer is not None else (None, 0, None, None, None)
    ld_pz = 0
    if pz is not None:
        rz = rowmat(pz)
        assert tuple(pz.shape) == (B, C, M) and rz is not None
        ld_pz = rz[2]
    _lib.call("fqss_gnq_bwd_apply", _p(xc), _p(qmin_x), _p(qmax_x), _p(g), _p(gamma), _p(beta), _p(mean_rstd), _p(gx), _p(ggamma),
              _p(gbeta), B, C, M, ld_xc, ld_g, rowmat(gx)[2], _p(qmin), _p(qmax), _p(ws), _p(pz), ld_pz, pact, _p(pslope), _p(pgacc),
              _p(pgbias), _stream())
    return gx


def dwq_fwd(xc, qmin_x, qmax_x, w, bias, dil, pad, act, slope, qmin, qmax, write_out, stats=None):
    """stats: a CodeStats from new_stats("dwq", ...) that receives the integer statistics of the output codes"""
    B, C, M, ld_xc = _codes3(xc)
    yc = empty_codes((B, C, M), xc.device)
    out = empty_act((B, C, M), xc.device)
    _lib.call("fqss_dwq_fwd", _p(xc), _p(qmin_x), _p(qmax_x), _p(w), _p(bias), _p(yc), _p(out) if write_out else None,
              B, C, M, w.shape[-1], dil, pad, ld_xc, rowmat(yc)[2], rowmat(out)[2], act, _p(slope), _p(qmin), _p(qmax),
              _p(stats.ws) if stats is not None else None, _stream())
    return out, yc


def dwq_bwd_z(xc, qmin_x, qmax_x, w, bias, g, dil, pad, act, slope, qmin, qmax, gacc, gbias):
    B, C, M, ld_xc = _codes3(xc)
    g, ld_g = _aligned_grad(g)
    gz = empty_act((B, C, M), xc.device)
    _lib.call("fqss_dwq_bwd_z", _p(xc), _p(qmin_x), _p(qmax_x), _p(w), _p(bias), _p(g), _p(gz), B, C, M, w.shape[-1], dil,
              pad, ld_xc, ld_g, rowmat(gz)[2], act, _p(slope), _p(qmin), _p(qmax), _p(gacc), _p(gbias), _stream())
    return gz


def _codes2(xc):
    rm = rowmat(xc)
    assert rm is not None and rm[2] % 16 == 0 and xc.data_ptr() % 16 == 0, "bad code layout"
    return rm


def ewq_fwd(ac, amin, amax, bc, bmin, bmax, bf, sb, act, slope, qmin, qmax, write_out):
    """y = fq(act(dec(a) + sb*B)); B = codes bc | fp32 bf | None.  returns (carrier/out, codes)"""
    rows, cols, ld_a = _codes2(ac)
    ld_b = _codes2(bc)[2] if bc is not None else 0
    ld_bf = 0
    if bf is not None:
        bf, ld_bf = _aligned_grad(bf)
    yc = empty_codes(tuple(ac.shape), ac.device)
    out = empty_act(tuple(ac.shape), ac.device)
    _lib.call("fqss_ewq_fwd", _p(ac), _p(amin), _p(amax), _p(bc), _p(bmin), _p(bmax), _p(bf), float(sb), _p(yc),
              _p(out) if write_out else None, rows, cols, ld_a, ld_b, ld_bf, rowmat(yc)[2], rowmat(out)[2], act, _p(slope),
              _p(qmin), _p(qmax), _stream())
    return out, yc


ADD_CHAIN_MAX = 24      # levels per fqss_add_chain_bwd launch (csrc/fused_q.hip kChainMax)


def add_chain_ok(rows, cols, C, nlev):
    return bool(_lib.query("fqss_add_chain_ok", rows, cols, C, nlev))


def add_chain_bwd(levels, g, prod_a=None):
    """the backward of a chain of AddQ layers in one launch (fqss_add_chain_bwd).  levels[0] = the first add of the forward; a level =
    dict(ac, amin, amax, bc, bmin, bmax, qmin, qmax, gacc, prod_b=(z, act, slope, gacc, gbias)); g: gradient of the top level's output;
    prod_a: the same 5-tuple for the bottom level's first operand, or None.  -> ([gz of b's producer per level], gz of a's producer or
    dL/d(a of the bottom level))"""
    ac0 = levels[0]["ac"]
    B, C, M = ac0.shape
    rows, cols, ld_a = _codes2(ac0)
    ld_b = _codes2(levels[0]["bc"])[2]
    g, ld_g = _aligned_grad(g)
    arr = (_lib.FqssAddChainLevel * len(levels))()
    outs, keep = [], []
    ld_bz = ld_bout = None
    for r, lv in zip(arr, levels):
        z, pact, pslope, pgacc, pgbias = lv["prod_b"]
        assert pact == ACT_NONE and tuple(lv["ac"].shape) == (B, C, M) and tuple(lv["bc"].shape) == (B, C, M) and tuple(z.shape) == (B, C, M)
        o = empty_act((B, C, M), ac0.device)
        outs.append(o)
        assert _codes2(lv["ac"])[2] == ld_a and _codes2(lv["bc"])[2] == ld_b
        lz, lo = rowmat(z)[2], rowmat(o)[2]
        assert (ld_bz is None or (ld_bz, ld_bout) == (lz, lo))
        ld_bz, ld_bout = lz, lo
        r.ac, r.bc, r.bz, r.bout = _p(lv["ac"]), _p(lv["bc"]), _p(z), _p(o)
        r.amin, r.amax, r.bmin, r.bmax, r.qmin, r.qmax = (_p(lv[k]) for k in ("amin", "amax", "bmin", "bmax", "qmin", "qmax"))
        r.gacc, r.bgacc, r.bgbias = _p(lv["gacc"]), _p(pgacc), _p(pgbias)
    ga = aout = az = agacc = agbias = None
    ld_ga = ld_az = ld_aout = 0
    if prod_a is not None:
        az, pact, pslope, agacc, agbias = prod_a
        assert pact == ACT_NONE and tuple(az.shape) == (B, C, M)
        aout = empty_act((B, C, M), ac0.device)
        ld_az, ld_aout = rowmat(az)[2], rowmat(aout)[2]
    else:
        ga = empty_act((B, C, M), ac0.device)
        ld_ga = rowmat(ga)[2]
    _lib.call("fqss_add_chain_bwd", arr, len(levels), _p(g), ld_g, _p(ga), ld_ga, _p(az), ld_az, _p(aout), ld_aout, _p(agacc), _p(agbias),
              rows, cols, C, ld_a, ld_b, ld_bz, ld_bout, _stream())
    return outs, (aout if prod_a is not None else ga)


def ewq_bwd_p(ac, amin, amax, bc, bmin, bmax, sb, g, act, slope, qmin, qmax, gacc, C, prod_a=None, prod_b=None):
    """ewq_bwd with the epilogue backward of the layers that produced operand a and/or b fused in.
    prod = (z, act, slope, gacc, gbias) -> returns (gz or None, gz_of_producer_a or None, gz_of_producer_b or None)"""
    rows, cols, ld_a = _codes2(ac)
    ld_b = _codes2(bc)[2] if bc is not None else 0
    g, ld_g = _aligned_grad(g)
    need_plain = prod_a is None or (bc is not None and prod_b is None)
    gz = empty_act(tuple(ac.shape), ac.device) if need_plain else None
    outs, args = [], []
    for pr in (prod_a, prod_b):
        if pr is None:
            outs.append(None)
            args += [None, 0, 0, None, None, None, None, 0]
        else:
            z, pact, pslope, pgacc, pgbias = pr
            o = empty_act(tuple(ac.shape), ac.device)
            outs.append(o)
            args += [_p(z), rowmat(z)[2], pact, _p(pslope), _p(pgacc), _p(pgbias), _p(o), rowmat(o)[2]]
    if DESC_ABI and ac.dim() == 3 and ac.shape[1] == C:
        # descriptor form (fqss_add_fq_bwd): tensors / quantizers / producers as structs instead of 38 positional arguments
        keep = [_desc(ac), _desc(bc), _desc(g), _desc(gz)]
        qa, qb, qo = _qparams(amin, amax), (_qparams(bmin, bmax) if bc is not None else None), _qparams(qmin, qmax, act, slope, gacc)
        prods = []
        for pr, o in zip((prod_a, prod_b), outs):
            if pr is None:
                prods.append(None)
                continue
            z, pact, pslope, pgacc, pgbias = pr
            dz, do = _desc(z), _desc(o)
            keep += [dz, do]
            P = _lib.FqssProducer()
            P.z, P.out, P.act, P.slope, P.gacc, P.gbias = _C.pointer(dz), _C.pointer(do), pact, _p(pslope), _p(pgacc), _p(pgbias)
            prods.append(P)
        _lib.call("fqss_add_fq_bwd", _ref(keep[0]), _ref(qa), _ref(keep[1]), _ref(qb), float(sb), _ref(keep[2]), _ref(keep[3]), _ref(qo),
                  _ref(prods[0]), _ref(prods[1]), None, 0, _stream())
        return gz, outs[0], outs[1]
    _lib.call("fqss_ewq_bwd_p", _p(ac), _p(amin), _p(amax), _p(bc), _p(bmin), _p(bmax), float(sb), _p(g), _p(gz), rows, cols, ld_a,
              ld_b, ld_g, rowmat(gz)[2] if gz is not None else 0, act, _p(slope), _p(qmin), _p(qmax), _p(gacc), C, *args, _stream())
    return gz, outs[0], outs[1]


def ewq_bwd(ac, amin, amax, bc, bmin, bmax, bf, sb, g, act, slope, qmin, qmax, gacc):
    rows, cols, ld_a = _codes2(ac)
    ld_b = _codes2(bc)[2] if bc is not None else 0
    ld_bf = 0
    if bf is not None:
        bf, ld_bf = _aligned_grad(bf)
    g, ld_g = _aligned_grad(g)
    gz = empty_act(tuple(ac.shape), ac.device)
    _lib.call("fqss_ewq_bwd", _p(ac), _p(amin), _p(amax), _p(bc), _p(bmin), _p(bmax), _p(bf), float(sb), _p(g), _p(gz),
              rows, cols, ld_a, ld_b, ld_bf, ld_g, rowmat(gz)[2], act, _p(slope), _p(qmin), _p(qmax), _p(gacc), _stream())
    return gz


DWQ_ROW_MAX = 12 * 1024   # kDwRowMax (csrc/fused_q.hip)


def dwq_bwd(xc, qmin_x, qmax_x, w, bias, g, dil, pad, act, slope, qmin, qmax, gacc, gbias, gw, want_gx=True, after=None, before=None):
    """whole backward of the coded depthwise layer in one launch: returns gx (or None); gw / gbias / gacc are "+=".
    after = dict(gamma, beta, mean_rstd, ws, qmin, qmax, ggamma, gbeta) of the GroupNormQ that consumes this layer's output: g is the
    gradient w.r.t. THAT layer's output and its apply pass runs on load; before = dict(xc0, qmin0, qmax0, gamma, beta, mean_rstd, gacc)
    of the GroupNormQ that produced xc: its rows pass runs on gx, the row sums land in before["ws"] (csrc/fused_q.hip k_dwq_bwd<3, GA, GB>)"""
    B, C, M, ld_xc = _codes3(xc)
    assert M <= DWQ_ROW_MAX
    g, ld_g = _aligned_grad(g)
    gx = empty_act((B, C, M), xc.device) if want_gx else None
    if after is not None or before is not None:
        a = b = None
        if after is not None:
            a = _lib.FqssGnAfter()
            a.gamma, a.beta, a.mean_rstd, a.ws = _p(after["gamma"]), _p(after["beta"]), _p(after["mean_rstd"]), _p(after["ws"])
            a.qmin, a.qmax, a.ggamma, a.gbeta = _p(after["qmin"]), _p(after["qmax"]), _p(after["ggamma"]), _p(after["gbeta"])
            assert after["ws"].numel() == 2 * B * C and after["ws"].dtype == torch.float64
        if before is not None:
            b = _lib.FqssGnBefore()
            x0 = before["xc0"]
            assert tuple(x0.shape) == (B, C, M) and want_gx
            before["ws"] = torch.empty(2 * B * C, device=xc.device, dtype=torch.float64)
            b.xc0, b.ld_xc0, b.qmin0, b.qmax0 = _p(x0), _codes3(x0)[3], _p(before["qmin0"]), _p(before["qmax0"])
            b.gamma, b.beta, b.mean_rstd, b.ws, b.gacc = _p(before["gamma"]), _p(before["beta"]), _p(before["mean_rstd"]), _p(before["ws"]), _p(before["gacc"])
        _lib.call("fqss_dwq_bwd_gn", _p(xc), _p(qmin_x), _p(qmax_x), _p(w), _p(bias), _p(g), _p(gx), _p(gw), B, C, M, w.shape[-1], dil, pad,
                  ld_xc, ld_g, rowmat(gx)[2] if gx is not None else 0, act, _p(slope), _p(qmin), _p(qmax), _p(gacc), _p(gbias), _ref(a), _ref(b),
                  _stream())
        return gx
    _lib.call("fqss_dwq_bwd", _p(xc), _p(qmin_x), _p(qmax_x), _p(w), _p(bias), _p(g), _p(gx), _p(gw), B, C, M, w.shape[-1], dil, pad,
              ld_xc, ld_g, rowmat(gx)[2] if gx is not None else 0, act, _p(slope), _p(qmin), _p(qmax), _p(gacc), _p(gbias), _stream())
    return gx


def dwq_bwd_w(gz, xc, qmin_x, qmax_x, gw, dil, pad):
    B, C, M, ld_xc = _codes3(xc)
    _lib.call("fqss_dwq_bwd_w", _p(gz), _p(xc), _p(qmin_x), _p(qmax_x), _p(gw), B, C, M, gw.shape[-1], dil, pad,
              rowmat(gz)[2], ld_xc, _stream())


# ------------------------------------------------------------------ K6  depthwise conv
def dwconv_fwd(x, w, bias, dil, pad):
    _need_gpu(x, w, bias)
    x, B, C, M, ld_x = _bcm(x)
    K = w.shape[-1]
    z = empty_act((B, C, M), x.device)
    _lib.call("fqss_dwconv_fwd", _p(x), _p(w), _p(bias), _p(z), B, C, M, K, dil, pad, ld_x, rowmat(z)[2], _stream())
    return z


def dwconv_bwd_x(gz, w, dil, pad):
    _need_gpu(gz, w)
    gz, B, C, M, ld_gz = _bcm(gz)
    K = w.shape[-1]
    gx = empty_act((B, C, M), gz.device)
    _lib.call("fqss_dwconv_bwd_x", _p(gz), _p(w), _p(gx), B, C, M, K, dil, pad, ld_gz, rowmat(gx)[2], _stream())
    return gx


def dwconv_bwd_w(gz, x, gw, dil, pad):
    _need_gpu(gz, x, gw)
    gz, B, C, M, ld_gz = _bcm(gz)
    x, _, _, _, ld_x = _bcm(x)
    K = gw.shape[-1]
    _lib.call("fqss_dwconv_bwd_w", _p(gz), _p(x), _p(gw), B, C, M, K, dil, pad, ld_gz, ld_x, _stream())


# ------------------------------------------------------------------ K7  GroupNorm(1, C)
def gn_fwd(x, gamma, beta, eps):
    _need_gpu(x, gamma, beta)
    x, B, C, M, ld_x = _bcm(x)
    z = empty_act((B, C, M), x.device)
    mean_rstd = torch.empty(B, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gn_fwd", _p(x), _p(gamma), _p(beta), _p(z), _p(mean_rstd), B, C, M, ld_x, rowmat(z)[2],
              float(eps), _p(ws), _stream())
    return z, mean_rstd


def gn_fwd_tail(x, gamma, beta, eps, tail, ls=None, res=None):
    """forward-only GroupNorm(1, C) + what follows it in ONE apply pass (no autograd: the frozen teacher).  tail 1: gelu(gn(x));
    tail 2: glu(gn(x)) * ls[:, None] + res -> [B, C/2, M]; None when the operands do not meet the kernel's 16-B row rule"""
    _need_gpu(x, gamma, beta, ls, res)
    x, B, C, M, ld_x = _bcm(x)
    if x.data_ptr() % 16 or ld_x % 4:
        return None
    ld_r = 0
    if tail == 2:
        res, Br, Cr, Mr, ld_r = _bcm(res)
        if (Br, Cr, Mr) != (B, C // 2, M) or ls.numel() != C // 2 or not ls.is_contiguous():
            return None          # (a residual whose rows are not 16-B aligned -- a module input of odd length -- is read with 4-B loads)
    y = empty_act((B, C // 2 if tail == 2 else C, M), x.device)
    ws = torch.empty(2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gn_fwd_tail", _p(x), _p(gamma), _p(beta), _p(y), B, C, M, ld_x, rowmat(y)[2], float(eps), _p(ws), tail, _p(ls), _p(res), ld_r,
              _stream())
    return y


def gn_bwd(gz, x, gamma, mean_rstd, ggamma, gbeta):
    _need_gpu(gz, x, gamma, mean_rstd, ggamma, gbeta)
    gz, B, C, M, ld_gz = _bcm(gz)
    x, _, _, _, ld_x = _bcm(x)
    gx = empty_act((B, C, M), x.device)
    ws = torch.empty(2 * B * C + 2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gn_bwd", _p(gz), _p(x), _p(gamma), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), B, C, M, ld_gz,
              ld_x, rowmat(gx)[2], _p(ws), _stream())
    return gx


def gnq_fwd_f(x, gamma, beta, eps, qmin, qmax, want_idx=True):
    """fq(GroupNorm(1, C)(x)) on a float input, the quantizer inside the apply pass -> (y, codes or None, mean_rstd)"""
    _need_gpu(x, gamma, beta, qmin, qmax)
    x, B, C, M, ld_x = _bcm(x)
    y = empty_act((B, C, M), x.device)
    idx = empty_codes((B, C, M), x.device) if want_idx else None
    mean_rstd = torch.empty(B, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gnq_fwd_f", _p(x), _p(gamma), _p(beta), _p(y), _p(idx), _p(mean_rstd), B, C, M, ld_x, rowmat(y)[2],
              rowmat(idx)[2] if idx is not None else 0, float(eps), _p(ws), _p(qmin), _p(qmax), _stream())
    return y, idx, mean_rstd


def gnq_bwd_f(g, x, gamma, beta, mean_rstd, qmin, qmax, gacc, ggamma, gbeta):
    """backward of gnq_fwd_f: the STE and its range partials inside the GroupNorm backward's own passes"""
    _need_gpu(g, x, gamma, beta, mean_rstd, ggamma, gbeta)
    g, B, C, M, ld_g = _bcm(g)
    x, _, _, _, ld_x = _bcm(x)
    gx = empty_act((B, C, M), x.device)
    ws = torch.empty(2 * B * C + 2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gnq_bwd_f", _p(g), _p(x), _p(gamma), _p(beta), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), B, C, M, ld_g, ld_x, rowmat(gx)[2],
              _p(ws), _p(qmin), _p(qmax), _p(gacc), _stream())
    return gx


# ------------------------------------------------------------------ K8 / K9 / K14
def axpby(a, b, sb, sa=1.0):
    """z = sa*a + sb*b"""
    _need_gpu(a, b)
    assert a.shape == b.shape
    a, rows, cols, ld_a = as_rowmat(a)
    b, _, _, ld_b = as_rowmat(b)
    z = empty_act(tuple(a.shape), a.device)
    _lib.call("fqss_axpby", _p(a), _p(b), float(sa), float(sb), _p(z), rows, cols, ld_a, ld_b, rowmat(z)[2], _stream())
    return z


def axpby_(a, b, sb, sa=1.0):
    """a = sa*a + sb*b in place (element-wise: the output may alias an operand)"""
    _need_gpu(a, b)
    assert a.shape == b.shape and a.is_contiguous()
    rm = rowmat(a)
    b, rows, cols, ld_b = as_rowmat(b)
    assert rm is not None and (rm[0], rm[1]) == (rows, cols)
    _lib.call("fqss_axpby", _p(a), _p(b), float(sa), float(sb), _p(a), rows, cols, rm[2], ld_b, rm[2], _stream())
    return a


def mul_bcast_fwd(mask, feat):
    """mask [B,S,C,M] * feat [B,C,M] -> [B,S,C,M]"""
    _need_gpu(mask, feat)
    B, S, C, M = mask.shape
    mask, _, _, ld_m = as_rowmat(mask)
    feat, _, _, ld_f = as_rowmat(feat)
    z = empty_act((B, S, C, M), mask.device)
    _lib.call("fqss_mul_bcast_fwd", _p(mask), _p(feat), _p(z), B, S, C, M, ld_m, ld_f, rowmat(z)[2], _stream())
    return z


def mul_bcast_bwd(gz, mask, feat):
    _need_gpu(gz, mask, feat)
    B, S, C, M = mask.shape
    gz, _, _, ld_gz = as_rowmat(gz)
    mask, _, _, ld_m = as_rowmat(mask)
    feat, _, _, ld_f = as_rowmat(feat)
    gmask = empty_act((B, S, C, M), mask.device)
    gfeat = empty_act((B, C, M), mask.device)
    _lib.call("fqss_mul_bcast_bwd", _p(gz), _p(mask), _p(feat), _p(gmask), _p(gfeat), B, S, C, M, ld_gz, ld_m, ld_f,
              rowmat(gmask)[2], rowmat(gfeat)[2], _stream())
    return gmask, gfeat


def mulq_fwd(mc, mmin, mmax, fc, fmin, fmax, qmin, qmax, write_out):
    """codes of fq(mask [B,S,C,M] * feat [B,C,M]) from the operands' codes; returns (carrier / fp32 out, codes [B,S,C,M])"""
    B, S, C, M = mc.shape
    assert tuple(fc.shape) == (B, C, M) and 1 <= S <= 4
    ld_m, ld_f = _codes2(mc)[2], _codes2(fc)[2]
    yc = empty_codes((B, S, C, M), mc.device)
    out = empty_act((B, S, C, M), mc.device)
    _lib.call("fqss_mulq_fwd", _p(mc), _p(mmin), _p(mmax), _p(fc), _p(fmin), _p(fmax), _p(yc), _p(out) if write_out else None,
              B, S, C, M, ld_m, ld_f, rowmat(yc)[2], rowmat(out)[2], _p(qmin), _p(qmax), _stream())
    return out, yc


def mulq_bwd(mc, mmin, mmax, fc, fmin, fmax, g, qmin, qmax, gacc, want_gfeat=True, prod=None):
    """backward of mulq_fwd: (gmask [B,S,C,M], gfeat [B,C,M] | None); prod = (z, act, slope, gacc, gbias) of the layer that produced
    the mask: its epilogue backward runs in the same launch and gmask is THAT layer's gz"""
    B, S, C, M = mc.shape
    ld_m, ld_f = _codes2(mc)[2], _codes2(fc)[2]
    g, ld_g = _aligned_grad(g)
    gmask = empty_act((B, S, C, M), mc.device)
    gfeat = empty_act((B, C, M), mc.device) if want_gfeat else None
    pargs = [None, 0, 0, None, None, None]
    if prod is not None:
        z, pact, pslope, pgacc, pgbias = prod
        pargs = [_p(z), rowmat(z)[2], pact, _p(pslope), _p(pgacc), _p(pgbias)]
    _lib.call("fqss_mulq_bwd", _p(mc), _p(mmin), _p(mmax), _p(fc), _p(fmin), _p(fmax), _p(g), _p(gmask), _p(gfeat), B, S, C, M,
              ld_m, ld_f, ld_g, rowmat(gmask)[2], rowmat(gfeat)[2] if gfeat is not None else 0, _p(qmin), _p(qmax), _p(gacc),
              *pargs, _stream())
    return gmask, gfeat


# ------------------------------------------------------------------ K10-K13  codec
def splitter2(x, normalize=True):
    """x [B,1,T] or [B,T] -> [B,2,T] (process.preprocess, n_splitter=2); multi-channel inputs [B, A, ...] flatten to [B, A*...]:
    the [B, 2, A*...] result IS torch.cat([msb, lsb], dim=1)"""
    _need_gpu(x)
    x2 = x.reshape(x.shape[0], -1).contiguous()
    B, T = x2.shape
    ws = torch.empty(2, device=x.device, dtype=torch.int32)
    obs_reset(ws)
    minmax(x2, ws)
    out = torch.empty(B, 2, T, device=x.device, dtype=torch.float32)
    _lib.call("fqss_splitter2" if normalize else "fqss_splitter2_raw", _p(x2), _p(out), B, T, _p(ws), _stream())
    return out


def frames_conv_fwd(x, w, stride, add=None):
    """x [N,Ci,T] dense, w [Co,Ci,K] -> z [N,Co,M] (+ add [N,Co,M])"""
    _need_gpu(x, w)
    x = x.contiguous()
    N, Ci, T = x.shape
    Co, _, K = w.shape
    M = (T - K) // stride + 1
    z = empty_act((N, Co, M), x.device)
    if add is not None:
        assert tuple(add.shape) == (N, Co, M)
        add, _, _, ld_add = as_rowmat(add)
        _lib.call("fqss_frames_conv_add_fwd", _p(x), _p(w.contiguous()), _p(add), ld_add, _p(z), N, Ci, Co, T, K, stride, M,
                  rowmat(z)[2], _stream())
        return z
    _lib.call("fqss_frames_conv_fwd", _p(x), _p(w.contiguous()), _p(z), N, Ci, Co, T, K, stride, M, rowmat(z)[2], _stream())
    return z


def ola_convtr_fwd(x, w, stride):
    """x [N,C,M], w [C,1,K] (or [C,K]) -> out [N,1,T], T=(M-1)*stride+K"""
    _need_gpu(x, w)
    x, N, C, M, ld_x = _bcm(x)
    K = w.shape[-1]
    T = (M - 1) * stride + K
    out = torch.empty(N, 1, T, device=x.device, dtype=torch.float32)
    _lib.call("fqss_ola_convtr_fwd", _p(x), _p(w.contiguous()), _p(out), N, C, M, ld_x, K, stride, T, _stream())
    return out


def ola_convtr_fwd_q(xc, qmin, qmax, w, stride):
    """ola_convtr_fwd on the u8 codes of x [N,C,M] (de-quantised on load)"""
    N, C, M, ld_x = _codes3(xc)
    K = w.shape[-1]
    T = (M - 1) * stride + K
    out = torch.empty(N, 1, T, device=xc.device, dtype=torch.float32)
    _lib.call("fqss_ola_convtr_fwd_q", _p(xc), _p(qmin), _p(qmax), _p(w.contiguous()), _p(out), N, C, M, ld_x, K, stride, T, _stream())
    return out


def ola_convtr_ok(w, stride):
    """window shapes the coded / masking forms of the decoder are built for"""
    return (w.shape[-1], stride) in ((16, 8), (32, 16), (2, 1)) and (w.dim() == 2 or w.shape[1] == 1)


def ola_convtr_mul_fwd(mask, feat, w, stride):
    """decoder of the float model on mask [B,S,C,M] * feat [B,C,M] formed on load -> [B*S,1,T]"""
    _need_gpu(mask, feat, w)
    B, S, C, M = mask.shape
    mask, _, _, ld_m = as_rowmat(mask)
    feat, _, _, ld_f = as_rowmat(feat)
    K = w.shape[-1]
    T = (M - 1) * stride + K
    out = torch.empty(B * S, 1, T, device=mask.device, dtype=torch.float32)
    _lib.call("fqss_ola_convtr_mul_fwd", _p(mask), _p(feat), _p(w.contiguous()), _p(out), B * S, S, C, M, ld_m, ld_f, K, stride, T, _stream())
    return out


def frames_wgrad(a, x, gw, stride):
    """gw[C][Ci][K] += sum_{n,m} a[n][c][m] * x[n][ci][m*stride+k]"""
    _need_gpu(a, x, gw)
    a, N, C, M, ld_a = _bcm(a)
    x = x.contiguous()
    _, Ci, T = x.shape
    K = gw.shape[-1]
    assert gw.is_contiguous() and gw.numel() == C * Ci * K
    if _frames_wgrad1_ok(x, T, K, stride) and ld_a % 4 == 0 and a.data_ptr() % 16 == 0:
        if Ci == 1:
            _lib.call("fqss_frames_wgrad1", _p(a), _p(x), _p(gw), N, C, M, ld_a, T, K, stride, _stream())
        else:       # one matrix-core launch per input channel (the encoder's splitter channels)
            for ci in range(Ci):
                _lib.call("fqss_frames_wgrad1s", _p(a), x.data_ptr() + 4 * ci * T, Ci * T, gw.data_ptr() + 4 * ci * K, Ci * K, N, C, M,
                          ld_a, T, K, stride, _stream())
        return
    _lib.call("fqss_frames_wgrad", _p(a), _p(x), _p(gw), N, C, Ci, M, ld_a, T, K, stride, _stream())


def _frames_wgrad1_ok(sig, T, K, stride):
    return (K, stride) in ((16, 8), (32, 16)) and os.environ.get("FQSS_FRAMES_WGRAD1", "1") != "0"


def frames_wgrad1_q(ac, qmin, qmax, sig, gw, stride):
    """gw[C][1][K] += sum_{n,m} dec(ac)[n][c][m] * sig[n][0][m*stride+k]; False when the shape is not served (caller decodes)"""
    N, C, M, ld_a = _codes3(ac)
    sig = sig.contiguous()
    T = sig.shape[-1]
    K = gw.shape[-1]
    if not (_frames_wgrad1_ok(sig, T, K, stride) and sig.numel() == N * T and gw.is_contiguous() and gw.numel() == C * K):
        return False
    _lib.call("fqss_frames_wgrad1_q", _p(ac), _p(qmin), _p(qmax), _p(sig), _p(gw), N, C, M, ld_a, T, K, stride, _stream())
    return True


# ------------------------------------------------------------------ K15 / K16
def kd_loss(est, fest, tgt, kd_lambda, want_grad=True, per_sample=False, threshold=None):
    """per_sample: the speechbrain env's objective (log per sample, mean over the samples whose loss exceeds `threshold`), B <= 2"""
    _need_gpu(est, fest, tgt)
    est, fest, tgt = est.contiguous(), fest.contiguous(), tgt.contiguous()
    B, S, T = est.shape
    assert S == 2, "the PIT kernel is built for n_src = 2"
    dev = est.device
    stats = torch.empty(B, 32, device=dev, dtype=torch.float64)
    out = torch.empty(4, device=dev, dtype=torch.float32)
    w = torch.empty(B, device=dev, dtype=torch.float32)
    sisdr = torch.empty(B, device=dev, dtype=torch.float32)
    gest = torch.empty_like(est) if want_grad else None
    if per_sample:
        _lib.call("fqss_kd_loss_per_sample", _p(est), _p(fest), _p(tgt), B, T, float(kd_lambda), int(threshold is not None),
                  float(threshold or 0.0), _p(stats), _p(out), _p(w), _p(sisdr), _p(gest), _stream())
    else:
        _lib.call("fqss_kd_loss", _p(est), _p(fest), _p(tgt), B, T, float(kd_lambda), _p(stats), _p(out), _p(w),
                  _p(sisdr), _p(gest), _stream())
    return out, w, sisdr, gest


def pit_sisdr_loss(est, tgt, want_grad=True):
    """the teacher-free PIT SI-SDR loss (kd_lambda = 0): (out, sisdr [B], gest); out[0] = loss in dB (mean over the batch)"""
    _need_gpu(est, tgt)
    est, tgt = est.contiguous(), tgt.contiguous()
    B, S, T = est.shape
    assert S == 2, "the PIT kernel is built for n_src = 2"
    dev = est.device
    stats = torch.empty(B, 32, device=dev, dtype=torch.float64)
    out, w, sisdr = torch.empty(4, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    gest = torch.empty_like(est) if want_grad else None
    _lib.call("fqss_pit_sisdr_loss", _p(est), _p(tgt), B, T, _p(stats), _p(out), _p(w), _p(sisdr), _p(gest), _stream())
    return out, sisdr, gest


def kd_moments(est, fest, tgt):
    """the 24 fp64 second-order moments per sample that the streaming pass of fqss_kd_loss accumulates ([B, 32], slots as in
    csrc/train_ops.hip: sums 0..5 of e0 e1 f0 f1 t0 t1, self products 6..11, e_i.t_j 12..15, e_i.f_j 16..19, f_i.t_j 20..23)"""
    _need_gpu(est, fest, tgt)
    est, fest, tgt = est.contiguous(), fest.contiguous(), tgt.contiguous()
    B, S, T = est.shape
    assert S == 2, "the moment kernel is built for n_src = 2"
    dev = est.device
    stats = torch.empty(B, 32, device=dev, dtype=torch.float64)
    _lib.call("fqss_kd_moments", _p(est), _p(fest), _p(tgt), B, T, _p(stats), _stream())
    return stats


class DetMode:
    """FQSS_DETERMINISTIC=1: the fp32 gradient atomics of the step become integer atomics on fixed-point shadows of the arenas they
    target (csrc/fqss_dev.h grad_add; include/fqss.h fqss_set_deterministic) -- slot 0 the parameter gradients, slot 1 the dL/dW_q
    arena, slot 2 a pool for the temporary weight-gradient accumulators of the un-fused (observer-phase) layers -- and `finish(slot)`
    rounds the sums once into the fp32 arena.  One DetMode owns the device-wide control block at a time: `activate()` (device
    synchronisation: outside graph capture) before the first backward that should run under it."""
    owner = None
    SLOTS = 3

    def __init__(self):
        self.arenas = [None] * self.SLOTS
        self.shadows = [None] * self.SLOTS
        self._top = 0           # bump pointer into the slot-2 pool (floats)

    def attach(self, slot, arena):
        assert arena.dtype == torch.float32 and arena.is_contiguous()
        self.arenas[slot] = arena
        self.shadows[slot] = torch.zeros(2 * arena.numel(), dtype=torch.int64, device=arena.device)
        if slot == 0 and self.arenas[2] is None:
            self.attach(2, torch.zeros(arena.numel(), device=arena.device))
            return
        if DetMode.owner is self:
            DetMode.owner = None
        self.activate()

    def activate(self):
        if DetMode.owner is self:
            return
        for slot in range(self.SLOTS):
            a, sh = self.arenas[slot], self.shadows[slot]
            _lib.call("fqss_set_deterministic", slot, _p(a), a.numel() if a is not None else 0, _p(sh))
        DetMode.owner = self

    def finish(self, slot):
        if self.arenas[slot] is not None:
            assert DetMode.owner is self, "another DetMode owns the control block: activate() outside graph capture first"
            _lib.call("fqss_det_finish", _p(self.arenas[slot]), self.arenas[slot].numel(), _p(self.shadows[slot]), _stream())

    # ---- temporaries (a weight gradient that autograd carries on instead of an arena slot)
    def begin_backward(self):
        self._top = 0

    def temp_like(self, w):
        """zeroed fp32 accumulator of w's shape inside the slot-2 pool (None when the pool is exhausted: the caller's own buffer then)"""
        n = (w.numel() + 63) // 64 * 64
        pool = self.arenas[2]
        if self._top + n > pool.numel():
            return None
        t = pool[self._top:self._top + w.numel()].view(w.shape)
        self._top += n
        t.zero_()
        return t

    def finish_temp(self, t):
        pool = self.arenas[2]
        off = (t.data_ptr() - pool.data_ptr()) // 4
        assert 0 <= off < pool.numel() and DetMode.owner is self
        _lib.call("fqss_det_finish", _p(t), t.numel(), self.shadows[2].data_ptr() + 16 * off, _stream())

    @staticmethod
    def off():
        for slot in range(DetMode.SLOTS):
            _lib.call("fqss_set_deterministic", slot, None, 0, None)
        DetMode.owner = None


def sumsq(g, acc):
    _lib.call("fqss_sumsq", _p(g), g.numel(), _p(acc), _stream())


def adam_clip(p, g, m, v, sumsq_acc, step_t, gnorm_out, max_norm, grad_scale, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              t0=None):
    _lib.call("fqss_adam_clip", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(sumsq_acc), float(max_norm),
              float(grad_scale), float(lr), float(beta1), float(beta2), float(eps), _p(step_t), _p(t0), _p(gnorm_out),
              _stream())


# ------------------------------------------------------------------ BatchNorm under BatchNormQ (csrc/batchnorm.hip)
def bn_moments(x):
    """[B, C, M] -> fp64 [C, 2]: (sum, sum of squares) per channel"""
    _need_gpu(x)
    x, B, C, M, ld = _bcm(x)
    out = torch.zeros(C, 2, device=x.device, dtype=torch.float64)
    _lib.call("fqss_bn_moments", _p(x), _p(out), B, C, M, ld, _stream())
    return out


def bn_apply(x, a, b):
    """y = x * a[c] + b[c]"""
    _need_gpu(x, a, b)
    x, B, C, M, ld = _bcm(x)
    y = empty_act((B, C, M), x.device)
    _lib.call("fqss_bn_apply", _p(x), _p(a.contiguous()), _p(b.contiguous()), _p(y), B, C, M, ld, rowmat(y)[2], _stream())
    return y


def bn_bwd_reduce(g, x):
    """-> fp64 [C, 2]: (sum g, sum g x) per channel"""
    _need_gpu(g, x)
    x, B, C, M, ld_x = _bcm(x)
    g, _, _, _, ld_g = _bcm(g)
    out = torch.zeros(C, 2, device=x.device, dtype=torch.float64)
    _lib.call("fqss_bn_bwd_reduce", _p(g), _p(x), _p(out), B, C, M, ld_g, ld_x, _stream())
    return out


def bn_bwd_apply(g, x, c1, c2, c3):
    """gx = g * c1[c] + x * c2[c] + c3[c]"""
    _need_gpu(g, x, c1, c2, c3)
    x, B, C, M, ld_x = _bcm(x)
    g, _, _, _, ld_g = _bcm(g)
    gx = empty_act((B, C, M), x.device)
    _lib.call("fqss_bn_bwd_apply", _p(g), _p(x), _p(c1.contiguous()), _p(c2.contiguous()), _p(c3.contiguous()), _p(gx), B, C, M, ld_g, ld_x,
              rowmat(gx)[2], _stream())
    return gx


# ------------------------------------------------------------------ fused float-teacher chain (csrc/teacher.hip)
def split3_planes(w2d):
    """fp32 [Co, Ci] -> three exact bf16 planes [3, Co, Ci] (uint16 storage)"""
    w2d = w2d.contiguous()
    planes = torch.empty(3, *w2d.shape, device=w2d.device, dtype=torch.int16)
    _lib.call("fqss_split3_planes", _p(w2d), _p(planes), w2d.numel(), _stream())
    Co, Ci = w2d.shape
    if Co % 256 == 0 and Ci % 128 == 0 and Ci <= 512 and TGEMM_TILED:
        # the image k_tgemm2 streams by LDS-DMA ([Co/256][Ci/32][3][256][32], swizzled); rides on the planes tensor
        tiles = torch.empty(3 * Co * Ci, device=w2d.device, dtype=torch.int16)
        _lib.call("fqss_split3_tiles", _p(w2d), _p(tiles), Co, Ci, _stream())
        planes._fqss_tiles = tiles
    return planes


TGEMM_TILED = os.environ.get("FQSS_TGEMM_V1", "0") == "0"     # FQSS_TGEMM_V1=1: the round-3 kernel everywhere (A/B)


TSTAT_SLOTS, TSTAT_STRIDE = 32, 16   # FQSS_TSTAT_SLOTS / FQSS_TSTAT_STRIDE (include/fqss.h)


def tstat_buffer(n, B, device):
    """n zeroed GroupNorm-statistics buffers for B samples (slot partials of (sum, sum^2), fp64)"""
    return torch.zeros(n, B, TSTAT_SLOTS, TSTAT_STRIDE, device=device, dtype=torch.float64)


def tgemm(planes, x, bias, act=ACT_NONE, slope=None, pro=0, pro_stats=None, pro_gamma=None, pro_beta=None, pro_eps=1e-8,
          pro_slope=None, stats_out=None, M1=None, r1=None, r2=None):
    """fused teacher GEMM; returns c1 (and c2 when M1 < Co)"""
    x, B, Ci, M, ld_x = _bcm(x)
    Co = planes.shape[1]
    M1 = Co if M1 is None else M1
    c1 = empty_act((B, M1, M), x.device)
    c2 = empty_act((B, Co - M1, M), x.device) if M1 < Co else None
    tiles = getattr(planes, "_fqss_tiles", None)
    if tiles is not None and M1 % 32 == 0:
        _lib.call("fqss_tgemm_tiled", _p(tiles), _p(x), B, Ci, Co, M, ld_x, pro, _p(pro_stats), _p(pro_gamma), _p(pro_beta),
                  float(pro_eps), _p(pro_slope), _p(bias), act, _p(slope), _p(stats_out), M1, _p(c1), _p(r1), rowmat(c1)[2],
                  _p(c2), _p(r2), rowmat(c2)[2] if c2 is not None else 0, _stream())
        return (c1, c2) if c2 is not None else c1
    if DESC_ABI and (r1 is None or rowmat(r1)[2] == rowmat(c1)[2]) and (r2 is None or rowmat(r2)[2] == rowmat(c2)[2]):
        td = _lib.FqssTGemmDesc()
        dpl, dx, dc1, dr1, dc2, dr2 = _desc(planes.view(3, Co, Ci), _lib.DT_U16), _desc(x), _desc(c1), _desc(r1), _desc(c2), _desc(r2)
        td.planes, td.x, td.c1 = _C.pointer(dpl), _C.pointer(dx), _C.pointer(dc1)
        if dr1 is not None:
            td.r1 = _C.pointer(dr1)
        if dc2 is not None:
            td.c2 = _C.pointer(dc2)
        if dr2 is not None:
            td.r2 = _C.pointer(dr2)
        td.pro, td.pro_stats, td.pro_gamma, td.pro_beta, td.pro_eps, td.pro_slope = pro, _p(pro_stats), _p(pro_gamma), _p(pro_beta), float(pro_eps), _p(pro_slope)
        td.bias, td.act, td.slope, td.stats_out, td.M1 = _p(bias), act, _p(slope), _p(stats_out), M1
        _lib.call("fqss_tgemm_desc", _ref(td), _stream())
        return (c1, c2) if c2 is not None else c1
    _lib.call("fqss_tgemm", _p(planes), _p(x), B, Ci, Co, M, ld_x, pro, _p(pro_stats), _p(pro_gamma), _p(pro_beta),
              float(pro_eps), _p(pro_slope), _p(bias), act, _p(slope), _p(stats_out), M1, _p(c1), _p(r1), rowmat(c1)[2],
              _p(c2), _p(r2), rowmat(c2)[2] if c2 is not None else 0, _stream())
    return (c1, c2) if c2 is not None else c1


def tdw(x, stats_in, gamma, beta, eps, w, bias, slope, stats_out, dil, pad):
    x, B, C, M, ld_x = _bcm(x)
    y = empty_act((B, C, M), x.device)
    _lib.call("fqss_tdw", _p(x), _p(stats_in), _p(gamma), _p(beta), float(eps), _p(w), _p(bias), _p(slope), _p(y),
              _p(stats_out), B, C, M, w.shape[-1], dil, pad, ld_x, rowmat(y)[2], _stream())
    return y


def tstats(x, ws):
    x, B, C, M, ld = _bcm(x)
    _lib.call("fqss_tstats", _p(x), B, C, M, ld, _p(ws), _stream())


# ================================================================== dual-path models (cfg 3)
UNARY_TANH, UNARY_SIGMOID, UNARY_DIVS = 0, 1, 2


def _rows(t, C):
    """a tensor whose last dim is C as a row matrix (tensor, R, ld); dense copy if its layout is foreign"""
    assert t.shape[-1] == C
    rm = rowmat(t)
    if rm is None or rm[1] != C:
        t = t.contiguous()
        rm = rowmat(t)
        if rm is None or rm[1] != C:
            return t, t.numel() // C, C
    return t, rm[0], rm[2]


def rowlin_fwd(x, w, bias, out=None):
    """z[..., o] = sum_i x[..., i] w[o][i] + bias[o]; `out`: a [..., Co] view (e.g. a column block) to write into"""
    _need_gpu(x, w, bias)
    Co, Ci = w.shape
    x, R, ld_x = _rows(x, Ci)
    assert w.stride(1) == 1
    z = torch.empty(*x.shape[:-1], Co, device=x.device, dtype=torch.float32) if out is None else out
    zz, Rz, ld_z = _rows(z, Co)
    assert zz is z and Rz == R, "rowlin_fwd: `out` must be a row-matrix view"
    w3 = _frozen_weight_planes(w, Ci, x, ld_x)
    if w3 is not None:
        _lib.call("fqss_rowlin_fwd_w3", _p(x), _p(w3), _p(bias), _p(z), R, Ci, Co, ld_x, ld_z, _stream())
    else:
        _lib.call("fqss_rowlin_fwd", _p(x), _p(w), _p(bias), _p(z), R, Ci, Co, ld_x, w.stride(0), ld_z, _stream())
    return z


W3_CACHE = os.environ.get("FQSS_W3_CACHE", "0") != "0"    # opt-in: measured neutral (docs/history/DESIGN_rounds_1-5.md 7e (4)), the default stays the on-the-fly split


def _frozen_weight_planes(w, Ci, x, ld_x):
    """the three exact bf16 planes of a FROZEN weight (a parameter used under torch.no_grad(): the float teacher's linears), split once
    and kept on the tensor until it is written to (`_version`): fqss_rowlin_fwd_w3 copies the weight tile instead of splitting it in
    every workgroup; None for every other weight (the student's fake-quantized weights are new tensors every step)"""
    if not W3_CACHE or _lib.BACKEND == "cpu" or torch.is_grad_enabled() or not isinstance(w, torch.nn.Parameter) or not w.is_contiguous():
        return None
    if Ci % 32 != 0 or ld_x % 4 != 0 or x.data_ptr() % 16 != 0:
        return None
    c = getattr(w, "_fqss_w3", None)
    if c is None or c[0] != w._version:
        if torch.cuda.is_current_stream_capturing():
            return None
        c = (w._version, split3_planes(w.detach()))
        w._fqss_w3 = c
    return c[1]


def rowlin_bwd_x(gz, w):
    _need_gpu(gz, w)
    Co, Ci = w.shape
    gz, R, ld_gz = _rows(gz, Co)
    gx = torch.empty(*gz.shape[:-1], Ci, device=gz.device, dtype=torch.float32)
    _lib.call("fqss_rowlin_bwd_x", _p(gz), _p(w), _p(gx), R, Ci, Co, ld_gz, w.stride(0), Ci, _stream())
    return gx


def rowlin_bwd_w(gz, x, gw):
    """gw[Co][Ci] += gz^T x over all rows (gw: caller-zeroed accumulator, may be a row-block view)"""
    _need_gpu(gz, x, gw)
    Co, Ci = gw.shape
    gz, R, ld_gz = _rows(gz, Co)
    x, R2, ld_x = _rows(x, Ci)
    assert R == R2 and gw.stride(1) == 1
    _lib.call("fqss_rowlin_bwd_w", _p(gz), _p(x), _p(gw), R, Ci, Co, ld_gz, ld_x, gw.stride(0), _stream())


PAIR_WGRAD = os.environ.get("FQSS_PAIR_WGRAD", "1") != "0"    # A/B switch: the two directions' W_hh gradients in one launch


def rowlin_bwd_w_pair(gz0, x0, gw0, gz1, x1, gw1):
    """two weight gradients of the same shape in one launch when the operands are views of the same tensors (constant element offsets
    between problem 0 and 1) and both accumulators live in one allocation; else two launches"""
    _need_gpu(gz0, x0, gw0, gz1, x1, gw1)
    Co, Ci = gw0.shape
    a0, R, ld_gz = _rows(gz0, Co)
    a1, R1, ld_gz1 = _rows(gz1, Co)
    b0, Rb, ld_x = _rows(x0, Ci)
    b1, Rb1, ld_x1 = _rows(x1, Ci)
    same = (R == R1 == Rb == Rb1 and ld_gz == ld_gz1 and ld_x == ld_x1 and gw0.shape == gw1.shape and gw0.stride() == gw1.stride() and gw0.stride(1) == 1
            and a0 is gz0 and a1 is gz1 and b0 is x0 and b1 is x1)
    if same and PAIR_WGRAD:
        d = [(t1.data_ptr() - t0.data_ptr()) for t0, t1 in ((a0, a1), (b0, b1), (gw0, gw1))]
        if all(v % 16 == 0 for v in d) and abs(d[2]) < (1 << 40):
            _lib.call("fqss_rowlin_bwd_w_batched", _p(a0), _p(b0), _p(gw0), R, Ci, Co, ld_gz, ld_x, gw0.stride(0), 2, d[0] // 4, d[1] // 4, d[2] // 4,
                      _stream())
            return
    rowlin_bwd_w(gz0, x0, gw0)
    rowlin_bwd_w(gz1, x1, gw1)


def colsum(g, out):
    """out[C] += column sums of g[..., C]"""
    _need_gpu(g, out)
    C = out.numel()
    g, R, ld = _rows(g, C)
    _lib.call("fqss_colsum", _p(g), _p(out), R, C, ld, _stream())


def layernorm_fwd(x, gamma, beta, eps):
    _need_gpu(x, gamma, beta)
    C = gamma.numel()
    x, R, ld_x = _rows(x, C)
    y = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    mean_rstd = torch.empty(R, 2, device=x.device, dtype=torch.float32)
    _lib.call("fqss_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean_rstd), R, C, ld_x, C, float(eps), _stream())
    return y, mean_rstd


def layernorm_bwd(gy, x, gamma, mean_rstd, ggamma, gbeta):
    _need_gpu(gy, x, gamma, mean_rstd, ggamma, gbeta)
    C = gamma.numel()
    gy, R, ld_gy = _rows(gy, C)
    x, _, ld_x = _rows(x, C)
    gx = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    _lib.call("fqss_layernorm_bwd", _p(gy), _p(x), _p(gamma), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), R, C, ld_gy, ld_x, C,
              _stream())
    return gx


def layernormq_fwd(x, gamma, beta, eps, qmin, qmax, want_codes):
    """y = fq(LN(x)) (+ its u8 codes) in one pass; the pre-quant value is not stored"""
    _need_gpu(x, gamma, beta, qmin, qmax)
    C = gamma.numel()
    x, R, ld_x = _rows(x, C)
    y = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    yc = torch.empty(*x.shape, device=x.device, dtype=torch.uint8) if want_codes else None
    mean_rstd = torch.empty(R, 2, device=x.device, dtype=torch.float32)
    _lib.call("fqss_layernormq_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(yc), _p(mean_rstd), R, C, ld_x, C, C, float(eps), _p(qmin),
              _p(qmax), _stream())
    return y, yc, mean_rstd


def layernormq_bwd(g, x, gamma, beta, mean_rstd, ggamma, gbeta, qmin, qmax, gacc):
    _need_gpu(g, x, gamma, beta, mean_rstd, ggamma, gbeta)
    C = gamma.numel()
    g, R, ld_g = _rows(g, C)
    x, _, ld_x = _rows(x, C)
    gx = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    _lib.call("fqss_layernormq_bwd", _p(g), _p(x), _p(gamma), _p(beta), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), R, C, ld_g, ld_x, C,
              _p(qmin), _p(qmax), _p(gacc), _stream())
    return gx


def add_layernorm_fwd(a, b, gamma, beta, eps, qmin=None, qmax=None, want_codes=False, qs=None, rmap=None):
    """s = a + b; y = LN(s) or fq(LN(s)) -> (s, y, codes or None, mean_rstd); qs = (min, max) of an AddQ's quantizer: y = LN(Q)(fq_s(s)),
    s stays the pre-quant sum (fqss_addq_layernorm_fwd).  rmap = (out_shape, d1, d2, t0, t1, t2) (with qs only): y and its codes leave in
    another row order -- row (i0*d1 + i1)*d2 + i2 at dense row i0*t0 + i1*t1 + i2*t2 of a tensor of `out_shape`"""
    _need_gpu(a, b, gamma, beta)
    C = gamma.numel()
    a, R, ld_a = _rows(a, C)
    b, Rb, ld_b = _rows(b, C)
    assert R == Rb and a.shape == b.shape
    s = torch.empty(*a.shape, device=a.device, dtype=torch.float32)
    yshape = tuple(a.shape) if rmap is None else tuple(rmap[0])
    y = torch.empty(*yshape, device=a.device, dtype=torch.float32)
    yc = torch.empty(*yshape, device=a.device, dtype=torch.uint8) if (want_codes and qmin is not None) else None
    mean_rstd = torch.empty(R, 2, device=a.device, dtype=torch.float32)
    if rmap is not None:
        assert y.numel() == a.numel() and yshape[-1] == C
        _lib.call("fqss_addq_layernorm_fwd_map", _p(a), _p(b), _p(gamma), _p(beta), _p(s), _p(y), _p(yc), _p(mean_rstd), R, C, ld_a, ld_b, C,
                  float(eps), _p(qmin), _p(qmax), _p(qs[0] if qs is not None else None), _p(qs[1] if qs is not None else None),
                  *[int(v) for v in rmap[1:]], _stream())
    elif qs is not None:
        _lib.call("fqss_addq_layernorm_fwd", _p(a), _p(b), _p(gamma), _p(beta), _p(s), _p(y), _p(yc), _p(mean_rstd), R, C, ld_a, ld_b, C, C, C,
                  float(eps), _p(qmin), _p(qmax), _p(qs[0]), _p(qs[1]), _stream())
    else:
        _lib.call("fqss_add_layernorm_fwd", _p(a), _p(b), _p(gamma), _p(beta), _p(s), _p(y), _p(yc), _p(mean_rstd), R, C, ld_a, ld_b, C, C, C,
                  float(eps), _p(qmin), _p(qmax), _stream())
    return s, y, yc, mean_rstd


def add_layernorm_bwd(g, gs, s, gamma, beta, mean_rstd, ggamma, gbeta, qmin=None, qmax=None, gacc=None, qs=None, gacc_s=None, rmap=None):
    """rmap (as in add_layernorm_fwd): g is dense in the forward's OUTPUT row order"""
    _need_gpu(g, gs, s, gamma, mean_rstd, ggamma, gbeta)
    C = gamma.numel()
    if rmap is not None:
        g = g.contiguous()
        R, ld_g = g.numel() // C, C
    else:
        g, R, ld_g = _rows(g, C)
    s, _, ld_s = _rows(s, C)
    ld_gs = 0
    if gs is not None:
        gs, _, ld_gs = _rows(gs, C)
    gx = torch.empty(*s.shape, device=s.device, dtype=torch.float32)
    if rmap is not None:
        assert qs is not None and gs is None and g.numel() == s.numel()
        _lib.call("fqss_addq_layernorm_bwd_map", _p(g), _p(s), _p(gamma), _p(beta), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), R, C, ld_s, C,
                  _p(qmin), _p(qmax), _p(gacc), _p(qs[0]), _p(qs[1]), _p(gacc_s), *[int(v) for v in rmap[1:]], _stream())
    elif qs is not None:
        _lib.call("fqss_addq_layernorm_bwd", _p(g), _p(gs), _p(s), _p(gamma), _p(beta), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), R, C, ld_g,
                  ld_gs, ld_s, C, _p(qmin), _p(qmax), _p(gacc), _p(qs[0]), _p(qs[1]), _p(gacc_s), _stream())
    else:
        _lib.call("fqss_add_layernorm_bwd", _p(g), _p(gs), _p(s), _p(gamma), _p(beta), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), R, C, ld_g,
                  ld_gs, ld_s, C, _p(qmin), _p(qmax), _p(gacc), _stream())
    return gx


def _ptr_array(ts):
    return (_C.c_void_p * len(ts))(*[_p(t) for t in ts])


def mha_prep_fwd(X, E, scale, ranges):
    """X [..., 3E] in-projection -> (q, k, v) [..., E]: the q / k / v quantizers on the thirds, q / scale, the div quantizer -- one pass.
    ranges: [(qmin, qmax)] of the q, k, v, div quantizers"""
    _need_gpu(X)
    assert X.is_contiguous() and X.shape[-1] == 3 * E
    R = X.numel() // (3 * E)
    q, k, v = (torch.empty(*X.shape[:-1], E, device=X.device) for _ in range(3))
    _lib.call("fqss_mha_prep_fwd", _p(X), _p(q), _p(k), _p(v), R, E, 3 * E, float(scale), _ptr_array([t for r in ranges for t in r]), _stream())
    return q, k, v


def mha_prep_fwd_c(X, E, scale, ranges):
    """mha_prep_fwd emitting the u8 CODES of q (on the div quantizer's grid), k, v: the operands of attn_long_fwd_c / _bwd_c"""
    _need_gpu(X)
    assert X.is_contiguous() and X.shape[-1] == 3 * E
    R = X.numel() // (3 * E)
    qc, kc, vc = (torch.empty(*X.shape[:-1], E, device=X.device, dtype=torch.uint8) for _ in range(3))
    _lib.call("fqss_mha_prep_fwd_c", _p(X), _p(qc), _p(kc), _p(vc), R, E, 3 * E, float(scale), _ptr_array([t for r in ranges for t in r]), _stream())
    return qc, kc, vc


def mha_prep_bwd(X, gq, gk, gv, E, scale, ranges, gaccs):
    """gradients of (q, k, v) -> gX [..., 3E]; the four quantizers' range partials go to gaccs (q, k, v, div)"""
    _need_gpu(X, gq, gk, gv)
    R = X.numel() // (3 * E)
    gq, gk, gv = gq.contiguous(), gk.contiguous(), gv.contiguous()
    gX = torch.empty_like(X)
    _lib.call("fqss_mha_prep_bwd", _p(X), _p(gq), _p(gk), _p(gv), _p(gX), R, E, 3 * E, 3 * E, float(scale),
              _ptr_array([t for r in ranges for t in r]), _ptr_array(gaccs), _stream())
    return gX


def as_rowmat_view(x):
    """(rows, cols, ld) when x [..., cols] is a row matrix whose rows are cols contiguous floats, `ld` apart, 16-B aligned and 4-float
    grouped (a last-dim slice of a contiguous tensor); else None"""
    if x.dim() < 2 or x.stride(-1) != 1 or x.shape[-1] % 4 != 0 or x.data_ptr() % 16 != 0:
        return None
    ld = x.stride(-2)
    if ld % 4 != 0 or ld < x.shape[-1]:
        return None
    exp = ld
    for d in range(x.dim() - 2, -1, -1):        # the leading dims must collapse into one row index
        if x.stride(d) != exp:
            return None
        exp *= x.shape[d]
    return x.numel() // x.shape[-1], x.shape[-1], ld


def padded_dense(x):
    """(rows, ld) when x is a row matrix -- trailing dims dense, leading dims one stride: [B, C, M] or [B, C, F, T] out of empty_act --
    whose rows are padded by less than LD_ALIGN floats and which lies inside its storage up to the last row's padding: an element-wise
    map may then run over the whole buffer, padding included (the padding holds no data by contract); else None"""
    if x.dim() < 2 or x.stride(-1) != 1 or x.data_ptr() % 16 != 0:
        return None
    rm = rowmat(x) or _rowmat_collapsed(x)
    if rm is None:
        return None
    rows, cols, ld = rm
    if ld <= cols or ld - cols >= LD_ALIGN or ld % 4 != 0:
        return None
    if (x.storage_offset() + rows * ld) * 4 > x.untyped_storage().nbytes():
        return None
    return rows, ld


def _like_padded(x, rows, ld):
    return torch.empty(rows * ld, device=x.device, dtype=torch.float32).as_strided(x.shape, x.stride())


def unary_fwd(x, kind, p=1.0):
    _need_gpu(x)
    if not x.is_contiguous() and _lib.BACKEND != "cpu":
        pd = padded_dense(x)
        if pd is not None:      # a row-padded activation ([B, C, 110250] / [B, C, F * 431] of HTDemucs): mapped where it lies, padding and all
            y = _like_padded(x, *pd)
            _lib.call("fqss_unary_fwd", _p(x), _p(y), pd[0] * pd[1], kind, float(p), _stream())
            return y
    if not x.is_contiguous() and _lib.BACKEND != "cpu" and os.environ.get("FQSS_UNARY_INPLACE", "1") != "0":      # (A/B knob)
        # a column block of a wider row matrix (the q third of an attention in-projection): read in place, written dense
        rm = as_rowmat_view(x)
        if rm is not None:
            rows, cols, ld = rm
            y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
            _lib.call("fqss_unary_rows_fwd", _p(x), _p(y), rows, cols, ld, cols, kind, float(p), _stream())
            return y
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.call("fqss_unary_fwd", _p(x), _p(y), x.numel(), kind, float(p), _stream())
    return y


UNARY_ROUND, UNARY_FLOOR, UNARY_SIGN, UNARY_CLIP = 4, 5, 6, 7


def unary2_fwd(x, kind, p=0.0, p2=0.0):
    """value maps of the STE helpers: round (half to even) / floor / sign / clip(x, p, p2)"""
    _need_gpu(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.call("fqss_unary2_fwd", _p(x), _p(y), x.numel(), kind, float(p), float(p2), _stream())
    return y


def unary_bwd(g, y, kind, p=1.0):
    _need_gpu(g, y)
    if y is not None and not y.is_contiguous() and _lib.BACKEND != "cpu":
        pd = padded_dense(y)
        if pd is not None:
            if g.shape != y.shape or g.stride() != y.stride() or padded_dense(g) != pd:      # (the gradient arrives dense or in another layout: into y's)
                gp = _like_padded(y, *pd)
                gp.copy_(g)
                g = gp
            gx = _like_padded(y, *pd)
            _lib.call("fqss_unary_bwd", _p(g), _p(y), _p(gx), pd[0] * pd[1], kind, float(p), _stream())
            return gx
        y = y.contiguous()
    g = g.contiguous()
    gx = torch.empty_like(g)
    _lib.call("fqss_unary_bwd", _p(g), _p(y), _p(gx), g.numel(), kind, float(p), _stream())
    return gx


def permute4(x, dims_out, strides_in, C, dense=True, pad_out=False):
    """y[i0][i1][i2][:C] = x[i0*s0 + i1*s1 + i2*s2 : +C]   (strides in elements; dense: they assume a contiguous x -- else they are x's
    own strides and only its last dim must have unit stride: a view of a row-padded buffer is read in place).
    pad_out: y is a row-padded activation (empty_act: rows 16-B aligned) instead of a dense tensor -- for C % 4 != 0 every element-wise
    kernel behind the move then takes its 16-B form (fqss_permute4_ld)"""
    _need_gpu(x)
    if dense or x.stride(-1) != 1:
        assert dense, "permute4: explicit strides need a unit stride along the last dim"
        x = x.contiguous()
    n0, n1, n2 = dims_out
    if pad_out and C % 4 != 0 and _lib.BACKEND != "cpu":
        y = empty_act((n0, n1, n2, C), x.device)
        _lib.call("fqss_permute4_ld", _p(x), _p(y), n0, n1, n2, C, strides_in[0], strides_in[1], strides_in[2], y.stride(-2), _stream())
        return y
    y = torch.empty(n0, n1, n2, C, device=x.device, dtype=torch.float32)
    _lib.call("fqss_permute4", _p(x), _p(y), n0, n1, n2, C, strides_in[0], strides_in[1], strides_in[2], _stream())
    return y


def dp_chunks(T, K):
    """(rest, S) of split_feature (dptnetq.py:232-259) for a length-T feature map and chunk length K"""
    P = K // 2
    rest = K - (P + T % K) % K
    return rest, 2 * (T + rest + P) // K


def dp_segment_fwd(f, K):
    """f [B][N][T] -> seg [K][B*S][N]"""
    f, B, N, T, ld = _bcm(f)
    _, S = dp_chunks(T, K)
    seg = torch.empty(K, B * S, N, device=f.device, dtype=torch.float32)
    _lib.call("fqss_dp_segment_fwd", _p(f), _p(seg), B, N, T, ld, K, S, _stream())
    return seg


def dp_segment_bwd(gseg, B, N, T, K):
    _need_gpu(gseg)
    gseg = gseg.contiguous()
    _, S = dp_chunks(T, K)
    gf = empty_act((B, N, T), gseg.device)
    _lib.call("fqss_dp_segment_bwd", _p(gseg), _p(gf), B, N, T, rowmat(gf)[2], K, S, _stream())
    return gf


def dp_merge_fwd(o, B, nspk, N, K, S):
    """o [S][B*K][nspk*N] -> a, b [B*nspk][N][Lm]"""
    _need_gpu(o)
    o = o.contiguous()
    Lm = (S // 2) * K - K // 2
    a = empty_act((B * nspk, N, Lm), o.device)
    b = empty_act((B * nspk, N, Lm), o.device)
    _lib.call("fqss_dp_merge_fwd", _p(o), _p(a), _p(b), B, nspk, N, K, S, Lm, rowmat(a)[2], _stream())
    return a, b


def dp_merge_bwd(ga, gb, B, nspk, N, K, S):
    Lm = (S // 2) * K - K // 2
    dev = (ga if ga is not None else gb).device
    if ga is None:
        ga = empty_act((B * nspk, N, Lm), dev).zero_()
    if gb is None:
        gb = empty_act((B * nspk, N, Lm), dev).zero_()
    ga, _, _, _, ld_ga = _bcm(ga)
    gb, _, _, _, ld_gb = _bcm(gb)
    go = torch.empty(S, B * K, nspk * N, device=dev, dtype=torch.float32)
    _lib.call("fqss_dp_merge_bwd", _p(ga), _p(gb), _p(go), B, nspk, N, K, S, Lm, ld_ga, ld_gb, _stream())
    return go


def ola2_fwd(y):
    """y [N][2][L] -> [N][L+1]"""
    y, N, two, L, ld = _bcm(y)
    assert two == 2
    out = torch.empty(N, L + 1, device=y.device, dtype=torch.float32)
    _lib.call("fqss_ola2_fwd", _p(y), _p(out), N, L, ld, _stream())
    return out


def ola2_bwd(g):
    _need_gpu(g)
    g = g.contiguous()
    N, L1 = g.shape
    gy = empty_act((N, 2, L1 - 1), g.device)
    _lib.call("fqss_ola2_bwd", _p(g), _p(gy), N, L1 - 1, rowmat(gy)[2], _stream())
    return gy


ATTN_STREAM = os.environ.get("FQSS_ATTN_STREAM", "1") != "0"


def _attn_stream_ok(ts, E, nh):
    """the split-bf16 streaming attention (csrc/attn_long.hip) also serves the 250-step sequences of the dual-path models: [L, B, E]
    views with 16-B aligned rows, head_dim 16 / 32 / 64 (measured against the LDS-resident kernels of csrc/attn.hip: docs/history/DESIGN_rounds_1-5.md 7/7b)"""
    return ATTN_STREAM and (E // nh) in (16, 32, 64) and all(t.dim() == 3 and t.stride(2) == 1 and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0
                                                          and t.stride(1) % 4 == 0 for t in ts)


def attn_fwd(q, k, v, L, B, nh, obs_attn=None, obs_soft=None):
    """q, k, v: [L*B rows][E] row matrices (views allowed) -> heads [L, B, E], stats"""
    E = q.shape[-1]
    if _attn_stream_ok((q, k, v), E, nh):
        return attn_long_fwd(q, k, v, nh, False, obs_attn, obs_soft)
    q, _, ld_q = _rows(q, E)
    k, _, ld_k = _rows(k, E)
    v, _, ld_v = _rows(v, E)
    o = torch.empty(L, B, E, device=q.device, dtype=torch.float32)
    stats = torch.empty(B * nh, L, 2, device=q.device, dtype=torch.float32)
    _lib.call("fqss_attn_fwd", _p(q), _p(k), _p(v), _p(o), _p(stats), L, B, nh, E // nh, ld_q, ld_k, ld_v, E, _p(obs_attn), _p(obs_soft),
              _stream())
    return o, stats


def attn_bwd(q, k, v, o, go, stats, L, B, nh):
    E = q.shape[-1]
    if _attn_stream_ok((q, k, v, o), E, nh) and go.dim() == 3:
        if not (go.stride(2) == 1 and go.data_ptr() % 16 == 0 and go.stride(0) % 4 == 0 and go.stride(1) % 4 == 0):
            go = go.contiguous()
        return attn_long_bwd(q, k, v, o, go, stats, nh, False)
    q, _, ld_q = _rows(q, E)
    k, _, ld_k = _rows(k, E)
    v, _, ld_v = _rows(v, E)
    o, _, ld_o = _rows(o, E)
    go, _, ld_go = _rows(go, E)
    gq = torch.empty(L, B, E, device=q.device, dtype=torch.float32)
    gk, gv = torch.empty_like(gq), torch.empty_like(gq)
    _lib.call("fqss_attn_bwd", _p(q), _p(k), _p(v), _p(o), _p(go), _p(stats), _p(gq), _p(gk), _p(gv), L, B, nh, E // nh, ld_q, ld_k, ld_v,
              ld_o, ld_go, E, E, E, _stream())
    return gq, gk, gv


def lstm_fwd(pre, whh, bhh, S, B, H, save=True):
    """pre [S, B, 8H] (fwd gates | reverse gates), whh [2, 4H, H], bhh [2, 4H] -> hout [S, B, 2H], (gsav, csav); save=False (no
    backward will follow: the frozen teacher): nothing is saved, (None, None)"""
    _need_gpu(pre, whh, bhh)
    assert pre.is_contiguous() and whh.is_contiguous() and bhh.is_contiguous()
    hout = torch.empty(S, B, 2 * H, device=pre.device, dtype=torch.float32)
    gsav = torch.empty(S, B, 8 * H, device=pre.device, dtype=torch.float32) if save else None
    csav = torch.empty(S, B, 2, 2 * H, device=pre.device, dtype=torch.float32) if save else None      # per direction: c | tanh(c)
    _lib.call("fqss_lstm_fwd", _p(pre), _p(whh), _p(bhh), _p(hout), _p(gsav), _p(csav), S, B, H, _stream())
    return hout, gsav, csav


def lstm_bwd(gout, whh, gsav, csav, S, B, H, gbias=None, gb4=None):
    """gbias ([8H] zeros): receives the column sums of dG (the bias gradients) from the same launch; gb4: instead, the four bias
    parameters' own gradient buffers (b_ih, b_hh forward, b_ih, b_hh reverse; [4H] each, accumulated into)"""
    _need_gpu(gout, whh)
    gout = gout.contiguous()
    dG = torch.empty(S, B, 8 * H, device=gout.device, dtype=torch.float32)
    if gb4 is not None:
        assert len(gb4) == 4 and all(t.numel() == 4 * H and t.is_contiguous() for t in gb4)
        _lib.call("fqss_lstm_bwd_b4", _p(gout), _p(whh), _p(gsav), _p(csav), _p(dG), _ptr_array(list(gb4)), S, B, H, _stream())
    elif gbias is None:
        _lib.call("fqss_lstm_bwd", _p(gout), _p(whh), _p(gsav), _p(csav), _p(dG), S, B, H, _stream())
    else:
        assert gbias.numel() == 8 * H and gbias.is_contiguous()
        _lib.call("fqss_lstm_bwd_b", _p(gout), _p(whh), _p(gsav), _p(csav), _p(dG), _p(gbias), S, B, H, _stream())
    return dG


# ================================================================== Sepformer (cfg 4)
def gnrows_fwd(x, gamma, beta, eps, RB, X, B):
    """gLN over all rows of a sample of a row matrix [..., C]; sample of row r = (r % RB) // X"""
    _need_gpu(x, gamma, beta)
    C = gamma.numel()
    x, R, ld_x = _rows(x, C)
    y = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    mean_rstd = torch.empty(B, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gnrows_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean_rstd), _p(ws), R, C, ld_x, C, RB, X, B, float(eps), _stream())
    return y, mean_rstd


def gnrows_bwd(gy, x, gamma, mean_rstd, ggamma, gbeta, RB, X, B):
    _need_gpu(gy, x, gamma, mean_rstd, ggamma, gbeta)
    C = gamma.numel()
    gy, R, ld_gy = _rows(gy, C)
    x, _, ld_x = _rows(x, C)
    gx = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    ws = torch.empty(2 * B, device=x.device, dtype=torch.float64)
    _lib.call("fqss_gnrows_bwd", _p(gy), _p(x), _p(gamma), _p(mean_rstd), _p(gx), _p(ggamma), _p(gbeta), _p(ws), R, C, ld_gy, ld_x, C,
              RB, X, B, _stream())
    return gx


def bcast_add(x, p):
    """x [L, Bp, C] + p [L, C] broadcast over Bp"""
    _need_gpu(x, p)
    x, p = x.contiguous(), p.contiguous()
    L, Bp, C = x.shape
    z = torch.empty_like(x)
    _lib.call("fqss_bcast_add", _p(x), _p(p), _p(z), L, Bp, C, _stream())
    return z


def bcast_sum(g):
    """g [L, Bp, C] -> [L, C] summed over Bp"""
    _need_gpu(g)
    g = g.contiguous()
    L, Bp, C = g.shape
    out = torch.empty(L, C, device=g.device, dtype=torch.float32)
    _lib.call("fqss_bcast_sum", _p(g), _p(out), L, Bp, C, _stream())
    return out


# ================================================================== first layer kernels of cfg 5 (HTDemucs)
UNARY_GELU = 3


def glu_fwd(x):
    """nn.GLU(dim=1) on [B, 2C, M] -> [B, C, M]"""
    x, B, C2, M, ld_x = _bcm(x)
    assert C2 % 2 == 0
    y = empty_act((B, C2 // 2, M), x.device)
    _lib.call("fqss_glu_fwd", _p(x), _p(y), B, C2 // 2, M, ld_x, rowmat(y)[2], _stream())
    return y


def glu_bwd(x, gy):
    x, B, C2, M, ld_x = _bcm(x)
    gy, _, _, _, ld_gy = _bcm(gy)
    gx = empty_act((B, C2, M), x.device)
    _lib.call("fqss_glu_bwd", _p(x), _p(gy), _p(gx), B, C2 // 2, M, ld_x, ld_gy, rowmat(gx)[2], _stream())
    return gx


def gluq_rows_ok(x):
    """fq(GLU(x)) in one pass needs 16-B aligned rows padded to a multiple of 4 (activation buffers of empty_act are)"""
    rm = rowmat(x) if x.dim() == 3 else None
    return rm is not None and rm[2] % 4 == 0 and x.data_ptr() % 16 == 0 and rm[2] >= (x.shape[-1] + 3) // 4 * 4


def gluq_fwd(x, qmode, qmin, qmax, obs_ws):
    """x [B, 2C, M] -> fq(GLU(x)) [B, C, M] (qmode QUANT), GLU(x) with the observer's min / max (OBSERVE) or plain GLU (BYPASS)"""
    _need_gpu(x)
    x, B, C2, M, ld_x = _bcm(x)
    assert C2 % 2 == 0
    y = empty_act((B, C2 // 2, M), x.device)
    _lib.call("fqss_gluq_fwd", _p(x), _p(y), B, C2 // 2, M, ld_x, rowmat(y)[2], qmode, _p(qmin), _p(qmax), _p(obs_ws), _stream())
    return y


def gluq_bwd(x, g, qmode, qmin, qmax, gacc):
    _need_gpu(x, g)
    x, B, C2, M, ld_x = _bcm(x)
    g, _, _, _, ld_g = _bcm(g)
    if ld_g % 4 != 0 or g.data_ptr() % 16 != 0:
        gp = empty_act(tuple(g.shape), g.device)
        gp.copy_(g)
        g, ld_g = gp, rowmat(gp)[2]
    gx = empty_act((B, C2, M), x.device)
    _lib.call("fqss_gluq_bwd", _p(x), _p(g), _p(gx), B, C2 // 2, M, ld_x, ld_g, rowmat(gx)[2], qmode, _p(qmin), _p(qmax), _p(gacc), _stream())
    return gx


def div_fwd(a, b):
    _need_gpu(a, b)
    assert a.shape == b.shape
    a, b = a.contiguous(), b.contiguous()
    y = torch.empty_like(a)
    _lib.call("fqss_div_fwd", _p(a), _p(b), _p(y), a.numel(), _stream())
    return y


def div_bwd(g, a, b):
    g = g.contiguous()
    ga, gb = torch.empty_like(a), torch.empty_like(b)
    _lib.call("fqss_div_bwd", _p(g), _p(a), _p(b), _p(ga), _p(gb), a.numel(), _stream())
    return ga, gb


def embedding_fwd(w, idx):
    """w [V, D] fp32, idx int64 [...] -> [..., D]"""
    _need_gpu(w)
    assert idx.dtype == torch.int64 and idx.is_cuda and w.is_contiguous()
    idx = idx.contiguous()
    V, D = w.shape
    out = torch.empty(*idx.shape, D, device=w.device, dtype=torch.float32)
    _lib.call("fqss_embedding_fwd", _p(w), _p(idx), _p(out), idx.numel(), D, V, _stream())
    return out


def embedding_bwd(g, idx, gw):
    """gw [V, D] += scatter of g [..., D] by idx"""
    _need_gpu(g, gw)
    g, idx = g.contiguous(), idx.contiguous()
    V, D = gw.shape
    _lib.call("fqss_embedding_bwd", _p(g), _p(idx), _p(gw), idx.numel(), D, V, _stream())


def qrow_eligible(Ci):
    return Ci % 16 == 0 and 16 <= Ci <= 2048


def qrow_fwd(xc, wc, bias, qmin_x, qmax_x, out=None):
    """u8 codes xc [..., Ci] (dense rows) x int8 weight codes (WCodes of a [Co, Ci] weight) -> z [..., Co] fp32 (exact integer
    sums on the int8 matrix cores); `out`: optional [..., Co] row-matrix view to write into"""
    Ci, Co = wc.Ci, wc.Co
    assert xc.dtype == torch.uint8 and xc.shape[-1] == Ci
    rm = rowmat(xc)
    assert rm is not None and rm[1] == Ci and rm[2] % 16 == 0, "activation codes need 16-B aligned rows"
    z = torch.empty(*xc.shape[:-1], Co, device=xc.device, dtype=torch.float32) if out is None else out
    zz, Rz, ld_z = _rows(z, Co)
    assert zz is z and Rz == rm[0]
    _lib.call("fqss_qrow_fwd", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias), _p(qmin_x), _p(qmax_x), _p(z), rm[0], Ci, Co, rm[2],
              ld_z, _stream())
    return z


def qrow_fwdq(xc, wc, bias, qmin_x, qmax_x, act, slope, qmin_y, qmax_y):
    """qrow_fwd with the layer's output quantizer in the GEMM epilogue -> (z, y = fq(act(z))), both [..., Co] fp32"""
    Ci, Co = wc.Ci, wc.Co
    assert xc.dtype == torch.uint8 and xc.shape[-1] == Ci
    rm = rowmat(xc)
    assert rm is not None and rm[1] == Ci and rm[2] % 16 == 0, "activation codes need 16-B aligned rows"
    z = torch.empty(*xc.shape[:-1], Co, device=xc.device, dtype=torch.float32)
    y = torch.empty_like(z)
    _lib.call("fqss_qrow_fwdq", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias), _p(qmin_x), _p(qmax_x), _p(z), _p(y), rm[0], Ci, Co, rm[2],
              Co, Co, act, _p(slope), _p(qmin_y), _p(qmax_y), _stream())
    return z, y


def qrow_fwdq2(xc, wc, bias, qmin_x, qmax_x, qmin1, qmax1, qmin2, qmax2):
    """qrow_fwd with BOTH quantizers of LinearQ -> NlQ(ReLU) in the GEMM epilogue -> (z, y = fq2(relu(fq1(z))), u8 codes of y)"""
    Ci, Co = wc.Ci, wc.Co
    assert xc.dtype == torch.uint8 and xc.shape[-1] == Ci and Co % 4 == 0
    rm = rowmat(xc)
    assert rm is not None and rm[1] == Ci and rm[2] % 16 == 0, "activation codes need 16-B aligned rows"
    z = torch.empty(*xc.shape[:-1], Co, device=xc.device, dtype=torch.float32)
    y = torch.empty_like(z)
    yc = torch.empty(*xc.shape[:-1], Co, device=xc.device, dtype=torch.uint8)
    _lib.call("fqss_qrow_fwdq2", _p(xc), _p(wc.idx), _p(wc.dw), _p(wc.rw), _p(bias), _p(qmin_x), _p(qmax_x), _p(z), _p(y), _p(yc), rm[0], Ci, Co,
              rm[2], Co, Co, Co, _p(qmin1), _p(qmax1), _p(qmin2), _p(qmax2), _stream())
    return z, y, yc


def qrow_bwd_ok(Ci, Co):
    return Ci % 4 == 0 and Co % 4 == 0


def qrow_bwd_x(gz, wc):
    """dL/dx of z = x @ w_q^T from the weight's int8 codes: gx[..., Ci] = gz[..., Co] @ (dw * wi)  (three bf16 products per k)"""
    _need_gpu(gz)
    Ci, Co = wc.Ci, wc.Co
    gz, R, ld_gz = _rows(gz, Co)
    gx = torch.empty(*gz.shape[:-1], Ci, device=gz.device, dtype=torch.float32)
    _lib.call("fqss_qrow_bwd_x", _p(gz), _p(wc.idx), _p(wc.dw), _p(gx), R, Ci, Co, ld_gz, Ci, _stream())
    return gx


def qrow_bwd_w(gz, xc, qmin_x, qmax_x, gw, gbias=None):
    """gw [Co, Ci] += gz^T @ dec(xc) from the activation's u8 codes (dense rows); gbias [Co] (optional) += column sums of gz"""
    _need_gpu(gz, gw, gbias)
    Co, Ci = gw.shape
    assert xc.dtype == torch.uint8 and xc.shape[-1] == Ci and gw.is_contiguous()
    gz, R, ld_gz = _rows(gz, Co)
    rm = rowmat(xc)
    assert rm is not None and rm[0] == R and rm[1] == Ci
    if gbias is not None:
        assert gbias.numel() == Co and gbias.is_contiguous()
        _lib.call("fqss_qrow_bwd_wb", _p(gz), _p(xc), _p(qmin_x), _p(qmax_x), _p(gw), _p(gbias), R, Ci, Co, ld_gz, rm[2], Ci, _stream())
    else:
        _lib.call("fqss_qrow_bwd_w", _p(gz), _p(xc), _p(qmin_x), _p(qmax_x), _p(gw), R, Ci, Co, ld_gz, rm[2], Ci, _stream())


class RowWgradQueue:
    """coded weight gradients of row-major linears queued over a backward segment and run by ONE launch per <= 32 of them
    (fqss_qrow_bwd_w_group): a single such GEMM is 256 workgroups of latency-bound k-tile chains, thousands overlap them"""

    def __init__(self):
        self.jobs = []

    def push(self, gz, xc, qmin_x, qmax_x, gw, gbias=None):
        _need_gpu(gz, gw, gbias)
        Co, Ci = gw.shape
        assert xc.dtype == torch.uint8 and xc.shape[-1] == Ci and gw.is_contiguous()
        gz, R, ld_gz = _rows(gz, Co)
        rm = rowmat(xc)
        assert rm is not None and rm[0] == R and rm[1] == Ci
        assert gbias is None or (gbias.numel() == Co and gbias.is_contiguous())
        self.jobs.append((gz, xc, qmin_x, qmax_x, gw, gbias, R, Ci, Co, ld_gz, rm[2]))

    def flush(self):
        if not self.jobs:
            return
        n = len(self.jobs)
        arr = (_lib.FqssRowWgradJob * n)()
        for j, (gz, xc, lo, hi, gw, gb, R, Ci, Co, ld_gz, ld_xc) in zip(arr, self.jobs):
            j.gz, j.xc, j.qmin_x, j.qmax_x, j.gw, j.gbias = _p(gz), _p(xc), _p(lo), _p(hi), _p(gw), _p(gb)
            j.R, j.Ci, j.Co, j.ld_gz, j.ld_xc, j.ld_gw = R, Ci, Co, ld_gz, ld_xc, Ci
        for n0 in range(0, n, 1024):            # (the entry point takes at most 1024 jobs: its per-shape index list)
            m = min(1024, n - n0)
            sub = (_lib.FqssRowWgradJob * m).from_buffer(arr, n0 * _C.sizeof(_lib.FqssRowWgradJob))
            _lib.call("fqss_qrow_bwd_w_group", sub, m, _stream())
        self.jobs = []


def qrow_bwd_w_pair(gz0, gz1, xc, qmin_x, qmax_x, gw0, gw1):
    """two coded weight gradients against the SAME input codes in one launch (gz0 / gz1: two column blocks of one tensor)"""
    _need_gpu(gz0, gz1, gw0, gw1)
    Co, Ci = gw0.shape
    a0, R, ld0 = _rows(gz0, Co)
    a1, R1, ld1 = _rows(gz1, Co)
    d = [a1.data_ptr() - a0.data_ptr(), gw1.data_ptr() - gw0.data_ptr()]
    if not (PAIR_WGRAD and a0 is gz0 and a1 is gz1 and R == R1 and ld0 == ld1 and gw0.shape == gw1.shape and gw0.is_contiguous() and gw1.is_contiguous()
            and d[0] % 16 == 0 and d[1] % 4 == 0):
        qrow_bwd_w(gz0, xc, qmin_x, qmax_x, gw0)
        qrow_bwd_w(gz1, xc, qmin_x, qmax_x, gw1)
        return
    assert xc.dtype == torch.uint8 and xc.shape[-1] == Ci
    rm = rowmat(xc)
    assert rm is not None and rm[0] == R and rm[1] == Ci
    _lib.call("fqss_qrow_bwd_w_batched", _p(a0), _p(xc), _p(qmin_x), _p(qmax_x), _p(gw0), R, Ci, Co, ld0, rm[2], Ci, 2, d[0] // 4, 0, d[1] // 4, _stream())


# ------------------------------------------------------------------ general convolution geometry (HTDemucs layers)
class ConvGeom:
    """kernel / stride / zero padding / dilation of a 2-D convolution over [B, C, H, W] (1-D convs run with H = 1)"""
    __slots__ = ("kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw")

    def __init__(self, k, s=(1, 1), p=(0, 0), d=(1, 1)):
        (self.kh, self.kw), (self.sh, self.sw), (self.ph, self.pw), (self.dh, self.dw) = k, s, p, d

    def out_hw(self, H, W):
        return ((H + 2 * self.ph - self.dh * (self.kh - 1) - 1) // self.sh + 1,
                (W + 2 * self.pw - self.dw * (self.kw - 1) - 1) // self.sw + 1)

    def args(self):
        return (self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, self.dh, self.dw)


def empty_sig(shape, device):
    """[B, C, H, W] signal whose (H*W) planes are dense and padded to 16 floats: viewable as [B, C, H*W] rows"""
    B, C, H, W = shape
    return empty_act((B, C, H * W), device).view(B, C, H, W)


def _sig4(x):
    """-> (x, sb, sc, sh) of a [B, C, H, W] signal with unit W stride and non-overlapping planes; foreign layouts are copied"""
    assert x.dim() == 4, "expected [B, C, H, W]"
    B, C, H, W = x.shape
    st = x.stride()
    ok = (st[3] == 1 or W == 1) and (H == 1 or st[2] >= W) and (C == 1 or st[1] >= (st[2] * (H - 1) if H > 1 else 0) + W) \
        and (B == 1 or st[0] >= (st[1] * (C - 1) if C > 1 else 0) + W)
    if not ok:
        c = empty_sig((B, C, H, W), x.device)
        c.copy_(x)
        x, st = c, c.stride()
    sh = st[2] if H > 1 else W
    sc = st[1] if C > 1 else sh * (H - 1) + W
    sb = st[0] if B > 1 else sc * (C - 1) + W
    return x, sb, sc, sh


# ---- stride-1 convolutions on halo-packed planes (fqss_halo_pack / fqss_conv2_*): no frame image
CONV_HALO = os.environ.get("FQSS_CONV_HALO", "1") != "0"      # (A/B knob: "0" = every general convolution gathers frames)
CONV_PHASE = os.environ.get("FQSS_CONV_PHASE", "1") != "0"    # (A/B knob: "0" = the strided convolutions gather frames)


class HaloPlan:
    """planes of a stride-1 convolution (ConvGeom, H x W input): xp [B, Ci, plane_x] = (H + 2 ph) rows of Wp floats with the input at
    (ph, pw); the packed output gradient gzp [B, Co, plane_g] = (Ho + 2 phg) rows with halo (phg, pwg) = ((kh-1) dh - ph, (kw-1) dw - pw);
    outputs come back on grids of pitch Wp ([.., Ho, Wp] / [.., H, Wp]) and are returned as their [.., :Wo] / [.., :W] views"""
    __slots__ = ("geom", "H", "W", "Ho", "Wo", "phg", "pwg", "Wp", "plane_x", "plane_g", "taps")

    def __init__(self, H, W, geom):
        assert geom.sh == 1 and geom.sw == 1
        self.geom, self.H, self.W = geom, H, W
        self.Ho, self.Wo = geom.out_hw(H, W)
        self.phg, self.pwg = (geom.kh - 1) * geom.dh - geom.ph, (geom.kw - 1) * geom.dw - geom.pw
        self.Wp = (max(W + 2 * geom.pw, self.Wo + 2 * max(self.pwg, 0)) + 3) // 4 * 4
        slack = (geom.kw - 1) * geom.dw + 8
        self.plane_x = ((H + 2 * geom.ph) * self.Wp + slack + 15) // 16 * 16
        self.plane_g = ((self.Ho + 2 * max(self.phg, 0)) * self.Wp + slack + 15) // 16 * 16
        self.taps = geom.kh * geom.kw

    def ok(self, Ci, Co, coded):
        """shapes the implicit kernels serve: padding no wider than the kernel reaches, planes below 2^24 floats, reductions below
        2^16 rows aligned for the weight loads (16 int8 codes / 4 floats per request)"""
        k1, k2 = Ci * self.taps, Co * self.taps
        al = 16 if coded else 4
        return (CONV_HALO and _lib.BACKEND != "cpu" and self.Ho >= 1 and self.Wo >= 1 and self.phg >= 0 and self.pwg >= 0
                and max(self.plane_x, self.plane_g) < (1 << 24) and max(k1, k2) < (1 << 16) and k1 % al == 0 and k2 % al == 0
                and (not coded or Co <= 1024))


def halo_pack(x4, ph, pw, Wp, plane):
    """x4 [B, C, H, W] (unit W stride) -> [B, C, plane]: rows of Wp floats, x at (ph, pw), zeros elsewhere"""
    _need_gpu(x4)
    x4, sb, sc, sh = _sig4(x4)
    B, C, H, W = x4.shape
    xp = torch.empty(B, C, plane, device=x4.device, dtype=torch.float32)
    _lib.call("fqss_halo_pack", _p(x4), _p(xp), B, C, H, W, sb, sc, sh, ph, pw, Wp, plane, _stream())
    return xp


def conv2_fwd(xp, plan, Co, wc, w2, bias):
    """z [B, Co, Ho, Wo] (a view of a pitch-Wp buffer) = conv(xp); wc: WCodes of the fake-quantized weight (idx [Co][Ci * taps]) or None:
    w2 float [Co][Ci * taps]"""
    g = plan.geom
    B, Ci = xp.shape[0], xp.shape[1]
    zb = torch.empty(B, Co, plan.Ho, plan.Wp, device=xp.device, dtype=torch.float32)
    N = plan.Ho * plan.Wp
    if wc is not None:
        _lib.call("fqss_conv2_fwd_wq", _p(xp), _p(wc.idx), _p(wc.dw), _p(bias), _p(zb), B, Ci, Co, plan.taps, g.kw, 0, g.dh * plan.Wp, g.dw, N,
                  plan.plane_x, N, _stream())
    else:
        assert w2.is_contiguous() and w2.shape == (Co, Ci * plan.taps)
        _lib.call("fqss_conv2_fwd_x3s", _p(xp), _p(w2), _p(bias), _p(zb), B, Ci, Co, plan.taps, g.kw, 0, g.dh * plan.Wp, g.dw, N, plan.plane_x, N,
                  _stream())
    return zb[..., :plan.Wo]


def conv2_bwd_x(gzp, plan, Ci, wcT, dw, w2T, raw=False):
    """gx [B, Ci, H, W] (a view of a pitch-Wp buffer; raw: the [B, Ci, H, Wp] buffer itself) from the packed gradient; wcT int8
    [Ci][Co * taps] + dw [Co], or w2T float"""
    g = plan.geom
    B, Co = gzp.shape[0], gzp.shape[1]
    gb = torch.empty(B, Ci, plan.H, plan.Wp, device=gzp.device, dtype=torch.float32)
    N = plan.H * plan.Wp
    base = (g.kh - 1) * g.dh * plan.Wp + (g.kw - 1) * g.dw
    if wcT is not None:
        _lib.call("fqss_conv2_bwd_x_wq", _p(gzp), _p(wcT), _p(dw), _p(gb), B, Ci, Co, plan.taps, g.kw, base, -g.dh * plan.Wp, -g.dw, N,
                  plan.plane_g, N, _stream())
    else:
        _lib.call("fqss_conv2_fwd_x3s", _p(gzp), _p(w2T), None, _p(gb), B, Co, Ci, plan.taps, g.kw, base, -g.dh * plan.Wp, -g.dw, N,
                  plan.plane_g, N, _stream())
    return gb if raw else gb[..., :plan.W]


def conv2_bwd_w(gzp, xp, gw2, plan):
    """gw2 [Co][Ci * taps] += sum over batch and positions (caller-zeroed accumulator)"""
    g = plan.geom
    B, Co = gzp.shape[0], gzp.shape[1]
    Ci = xp.shape[1]
    assert gw2.is_contiguous() and gw2.numel() == Co * Ci * plan.taps
    _lib.call("fqss_conv2_bwd_w", _p(gzp), _p(xp), _p(gw2), B, Ci, Co, plan.taps, g.kw, g.dh * plan.Wp, g.dw, plan.phg * plan.Wp + plan.pwg,
              plan.plane_g, plan.plane_x, _stream())


class PhasePlan:
    """a convolution strided along ONE axis (kernel k = T s, stride s, padding p, dilation 1; (k, 1) kernels along H of [B, C, H, W] or
    1-D along W) as a stride-1 convolution with T taps over the s C phase planes of its input (fqss_phase_pack, csrc/conv_frames.hip):
    `inner` is that convolution's HaloPlan, `perm` regroups a weight's (ci, t) columns as (ci, r, q')"""
    __slots__ = ("axis", "s", "k", "p", "T", "H", "W", "Ho", "Wo", "Hy", "inner", "perm", "inv")
    _perms = {}

    def __init__(self, H, W, geom, pad_to=None, no=None):
        """no: the frame grid along the strided axis when it is not the geometry's own (a transposed convolution's input length: its
        output may be a window, K.frames_ola)"""
        self.axis = 0 if geom.sh > 1 else 1
        self.s, self.k, self.p = (geom.sh, geom.kh, geom.ph) if self.axis == 0 else (geom.sw, geom.kw, geom.pw)
        self.T, self.H, self.W = self.k // self.s, H, W
        if no is None:
            no = ((H if self.axis == 0 else (pad_to or W)) + 2 * self.p - self.k) // self.s + 1
        self.Hy = no + self.T - 1
        if self.axis == 0:
            self.Ho, self.Wo = no, W
            self.inner = HaloPlan(self.Hy, W, ConvGeom((self.T, 1)))
        else:
            self.Ho, self.Wo = 1, no
            self.inner = HaloPlan(1, self.Hy, ConvGeom((1, self.T)))

    @staticmethod
    def serves(H, W, geom):
        """strided along exactly one axis with a kernel that is a whole number of strides long, nothing along the other axis"""
        if geom.dh != 1 or geom.dw != 1:
            return False
        if geom.sh > 1:
            return geom.sw == 1 and geom.kw == 1 and geom.pw == 0 and geom.kh % geom.sh == 0 and geom.kh > geom.sh and geom.ph < geom.kh
        return geom.sw > 1 and H == 1 and geom.kh == 1 and geom.ph == 0 and geom.kw % geom.sw == 0 and geom.kw > geom.sw and geom.pw < geom.kw

    def ok(self, Ci, Co, coded):
        return self.Hy >= self.T and self.inner.ok(Ci * self.s, Co, coded) and self.H * self.W < (1 << 24)

    def taps(self, device):
        """[s][T] tap indices: tap q' of phase r is tap t0(r) + s q' of the kernel"""
        key = ("taps", self.k, self.s, self.p, str(device))      # (cached: a host -> device copy cannot be captured into a hipGraph)
        if key not in PhasePlan._perms:
            PhasePlan._perms[key] = torch.tensor([[(r + self.p) % self.s + self.s * q for q in range(self.T)] for r in range(self.s)],
                                                 dtype=torch.int64, device=device)
        return PhasePlan._perms[key]

    def perms(self, Ci, device):
        """(perm, inv): column j = (ci s + r) T + q' of the regrouped weight is column perm[j] = ci k + t0(r) + s q' of the conv's"""
        key = (Ci, self.k, self.s, self.p, str(device))
        if key not in PhasePlan._perms:
            idx = []
            for ci in range(Ci):
                for r in range(self.s):
                    t0 = (r + self.p) % self.s
                    idx += [ci * self.k + t0 + self.s * q for q in range(self.T)]
            perm = torch.tensor(idx, dtype=torch.int64, device=device)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel(), device=device)
            PhasePlan._perms[key] = (perm, inv)
        return PhasePlan._perms[key]


def phase_pack(x4, pp):
    """x4 [B, C, H, W] -> the phase planes [B, C s, plane] of a PhasePlan (zeros outside the signal: the padding, front and back)"""
    _need_gpu(x4)
    x4, sb, sc, sh = _sig4(x4)
    B, C, H, W = x4.shape
    xp = torch.empty(B, C * pp.s, pp.inner.plane_x, device=x4.device, dtype=torch.float32)
    _lib.call("fqss_phase_pack", _p(x4), _p(xp), B, C, H, W, sb, sc, sh, pp.axis, pp.s, pp.p, pp.inner.Wp, pp.inner.plane_x, _stream())
    return xp


def phase_unpack(gy, pp, shape, off=0, bias=None):
    """the inverse move: gy [B, C s, Hy, Wp] (phase planes on their pitch-Wp grid) -> [B, C, H, W] = `shape`, a window starting `off`
    positions into the strided axis; bias [C] is added"""
    B, C, H, W = shape
    assert gy.is_contiguous() and gy.shape[0] == B and gy.shape[1] == C * pp.s
    plane = gy.shape[2] * gy.shape[3]
    out = empty_sig((B, C, H, W), gy.device)
    _, sb, sc, sh = _sig4(out)
    _lib.call("fqss_phase_unpack", _p(gy), _p(out), B, C, H, W, sb, sc, sh, pp.axis, pp.s, pp.p, pp.Hy, pp.inner.Wp, plane, off, _p(bias), _stream())
    return out


def frames_gather(x, geom, out_hw=None):
    """x [B, C, H, W] -> frames [B, C*kh*kw, Ho*Wo] (rows padded to 16 floats), Ho, Wo.
    out_hw: a frame grid other than the geometry's (the adjoint of an overlap-add that wrote a window of its signal, frames_ola): frame
    positions that reach past the signal read zeros, frames past the grid are not produced"""
    _need_gpu(x)
    B, C, H, W = x.shape
    Ho, Wo = geom.out_hw(H, W)
    if out_hw is not None:
        Ho, Wo = out_hw
    if Ho < 1 or Wo < 1:
        raise ValueError(f"convolution input {H}x{W} is shorter than the kernel")
    x, sb, sc, sh = _sig4(x)
    f = empty_act((B, C * geom.kh * geom.kw, Ho * Wo), x.device)
    _lib.call("fqss_frames_gather", _p(x), _p(f), B, C, H, W, sb, sc, sh, *geom.args(), Ho, Wo, rowmat(f)[2], _stream())
    return f, Ho, Wo


def frames_ola(frames, bias, sig_shape, geom, out_hw=None):
    """adjoint of frames_gather (+ per-channel bias): frames [B, C*kh*kw, Ho*Wo] -> y [B, C, H, W].
    out_hw: the frame grid when it is not the geometry's for this H x W (the signal is a window of a transposed convolution's output: cut
    at the back, or padded at the front by more than the kernel reaches): contributions that land outside the signal are dropped, frames
    the grid does not hold contribute nothing"""
    _need_gpu(frames, bias)
    B, C, H, W = sig_shape
    Ho, Wo = geom.out_hw(H, W)
    if out_hw is not None:
        Ho, Wo = out_hw
    frames, Bf, R, M, ld = _bcm(frames)
    assert Bf == B and R == C * geom.kh * geom.kw and M == Ho * Wo, "frames do not match the signal shape / geometry"
    y = empty_sig((B, C, H, W), frames.device)
    _, sb, sc, sh = _sig4(y)
    _lib.call("fqss_frames_ola", _p(frames), _p(bias), _p(y), B, C, H, W, sb, sc, sh, *geom.args(), Ho, Wo, ld, _stream())
    return y


def chan_sum(g, out):
    """out[C] += sum over batch and positions of g [B, C, M]"""
    _need_gpu(g, out)
    g, B, C, M, ld = _bcm(g)
    assert out.numel() == C and out.is_contiguous()
    _lib.call("fqss_chan_sum", _p(g), _p(out), B, C, M, ld, _stream())


# ------------------------------------------------------------------ streaming attention (long sequences, cross attention)
def _row_strides(t, batch_first):
    """(sl, sb) element strides of a [L, B, E] (or batch-first [B, L, E]) view with unit stride along E"""
    assert t.dim() == 3 and (t.stride(2) == 1 or t.shape[2] == 1), "expected a [., ., E] view with unit column stride"
    return (t.stride(1), t.stride(0)) if batch_first else (t.stride(0), t.stride(1))


def _stride_array(ts, batch_first):
    import ctypes
    vals = [s for t in ts for s in _row_strides(t, batch_first)]
    return (ctypes.c_int64 * len(vals))(*vals)


def attn_long_fwd(q, k, v, nh, batch_first, obs_attn=None, obs_soft=None):
    """q [Lq, B, E], k / v [Lk, B, E] views (batch_first: [B, L, E]) -> heads (same layout as q, dense), stats [B*nh, Lq, 2]"""
    _need_gpu(q, k, v)
    B, Lq = (q.shape[0], q.shape[1]) if batch_first else (q.shape[1], q.shape[0])
    Lk = k.shape[1] if batch_first else k.shape[0]
    E = q.shape[2]
    assert k.shape[2] == E and v.shape == k.shape and E % nh == 0 and (k.shape[0] if batch_first else k.shape[1]) == B
    o = torch.empty(q.shape, device=q.device, dtype=torch.float32)
    stats = torch.empty(B * nh, Lq, 2, device=q.device, dtype=torch.float32)
    _lib.call("fqss_attn_long_fwd", _p(q), _p(k), _p(v), _p(o), _p(stats), Lq, Lk, B, nh, E // nh, _stride_array((q, k, v, o), batch_first),
              _p(obs_attn), _p(obs_soft), _stream())
    return o, stats


def attn_long_bwd(q, k, v, o, go, stats, nh, batch_first):
    _need_gpu(q, k, v, o, go)
    B, Lq = (q.shape[0], q.shape[1]) if batch_first else (q.shape[1], q.shape[0])
    Lk = k.shape[1] if batch_first else k.shape[0]
    E = q.shape[2]
    if go.stride(2) != 1:
        go = go.contiguous()
    gq = torch.empty(q.shape, device=q.device, dtype=torch.float32)
    gk = torch.empty(k.shape, device=q.device, dtype=torch.float32)
    gv = torch.empty(k.shape, device=q.device, dtype=torch.float32)
    dsum = torch.empty(B * nh, Lq, device=q.device, dtype=torch.float32)
    _lib.call("fqss_attn_long_bwd", _p(q), _p(k), _p(v), _p(o), _p(go), _p(stats), _p(gq), _p(gk), _p(gv), _p(dsum), Lq, Lk, B, nh, E // nh,
              _stride_array((q, k, v, o, go, gq, gk, gv), batch_first), _stream())
    return gq, gk, gv


ATTN_CODED = os.environ.get("FQSS_ATTN_CODED", "1") != "0"


def attn_coded_ok(E, nh):
    return ATTN_CODED and (E // nh) in (16, 32, 64) and E % 8 == 0


def attn_long_fwd_c(qc, kc, vc, ranges, nh, batch_first):
    """the attention core from the u8 codes of q, k, v ([L, B, E] or [B, L, E] dense); ranges: [(qmin, qmax)] of the grids q (div), k, v
    are on (device scalars) -> heads (float, layout of qc), stats [B*nh, Lq, 2]"""
    _need_gpu(*[t for r in ranges for t in r])
    assert qc.is_cuda and kc.is_cuda and vc.is_cuda
    B, Lq = (qc.shape[0], qc.shape[1]) if batch_first else (qc.shape[1], qc.shape[0])
    Lk = kc.shape[1] if batch_first else kc.shape[0]
    E = qc.shape[2]
    assert qc.dtype == torch.uint8 and kc.dtype == torch.uint8 and vc.dtype == torch.uint8 and kc.shape == vc.shape and kc.shape[2] == E
    o = torch.empty(qc.shape, device=qc.device, dtype=torch.float32)
    stats = torch.empty(B * nh, Lq, 2, device=qc.device, dtype=torch.float32)
    _lib.call("fqss_attn_long_fwd_c", _p(qc), _p(kc), _p(vc), _ptr_array([t for r in ranges for t in r]), _p(o), _p(stats), Lq, Lk, B, nh, E // nh,
              _stride_array((qc, kc, vc, o), batch_first), _stream())
    return o, stats


def attn_long_bwd_c(qc, kc, vc, ranges, o, go, stats, nh, batch_first):
    """-> (gq, gk, gv): gradients with respect to the de-quantized q, k, v values"""
    _need_gpu(o, go)
    assert qc.is_cuda and kc.is_cuda and vc.is_cuda and qc.dtype == torch.uint8 and kc.dtype == torch.uint8 and vc.dtype == torch.uint8
    B, Lq = (qc.shape[0], qc.shape[1]) if batch_first else (qc.shape[1], qc.shape[0])
    Lk = kc.shape[1] if batch_first else kc.shape[0]
    E = qc.shape[2]
    if not (go.stride(2) == 1 and go.data_ptr() % 16 == 0 and go.stride(0) % 4 == 0 and go.stride(1) % 4 == 0):
        go = go.contiguous()
    gq = torch.empty(qc.shape, device=qc.device, dtype=torch.float32)
    gk = torch.empty(kc.shape, device=qc.device, dtype=torch.float32)
    gv = torch.empty(kc.shape, device=qc.device, dtype=torch.float32)
    dsum = torch.empty(B * nh, Lq, device=qc.device, dtype=torch.float32)
    _lib.call("fqss_attn_long_bwd_c", _p(qc), _p(kc), _p(vc), _ptr_array([t for r in ranges for t in r]), _p(o), _p(go), _p(stats), _p(gq), _p(gk),
              _p(gv), _p(dsum), Lq, Lk, B, nh, E // nh, _stride_array((qc, kc, vc, o, go, gq, gk, gv), batch_first), _stream())
    return gq, gk, gv


# ------------------------------------------------------------------ HTDemucs small ops (csrc/hd_ops.hip)
def chan_op(x, s, mode):
    """x [B, C, M] (*|+) s [C]: mode 0 multiply, 1 add"""
    _need_gpu(x, s)
    x, B, C, M, ld = _bcm(x)
    assert s.numel() == C and s.is_contiguous()
    y = empty_act((B, C, M), x.device)
    _lib.call("fqss_chan_op", _p(x), _p(s), _p(y), B, C, M, ld, rowmat(y)[2], mode, _stream())
    return y


def chan_scale_bwd(g, x, s, gs):
    """-> gx = g * s[c]; gs[c] += sum g * x"""
    _need_gpu(g, x, s, gs)
    g, B, C, M, ld_g = _bcm(g)
    x, _, _, _, ld_x = _bcm(x)
    gx = empty_act((B, C, M), g.device)
    _lib.call("fqss_chan_scale_bwd", _p(g), _p(x), _p(s), _p(gx), _p(gs), B, C, M, ld_g, ld_x, rowmat(gx)[2], _stream())
    return gx


def col_scale_fwd(x, s):
    """x [..., C] * s [C]"""
    _need_gpu(x, s)
    C = s.numel()
    xr, R, ld = _rows(x, C)
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _lib.call("fqss_col_scale_fwd", _p(xr), _p(s), _p(y), R, C, ld, C, _stream())
    return y


def col_scale_bwd(g, x, s, gs):
    _need_gpu(g, x, s, gs)
    C = s.numel()
    gr, R, ld_g = _rows(g, C)
    xr, _, ld_x = _rows(x, C)
    gx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _lib.call("fqss_col_scale_bwd", _p(gr), _p(xr), _p(s), _p(gx), _p(gs), R, C, ld_g, ld_x, C, _stream())
    return gx


def sample_meanstd(x):
    """x [B, ...] dense -> ms [B, 2] = (mean, unbiased std) over all other dims"""
    _need_gpu(x)
    x = x.contiguous()
    B = x.shape[0]
    ws = torch.zeros(B, 2, device=x.device, dtype=torch.float64)
    ms = torch.empty(B, 2, device=x.device, dtype=torch.float32)
    _lib.call("fqss_sample_meanstd", _p(x), _p(ws), _p(ms), B, x.numel() // B, _stream())
    return ms


def sample_norm(x, ms, inverse):
    """(x - mean_b) / (1e-5 + std_b), or x * std_b + mean_b when inverse"""
    _need_gpu(x, ms)
    x = x.contiguous()
    B = x.shape[0]
    assert tuple(ms.shape) == (B, 2) and ms.is_contiguous()
    y = torch.empty_like(x)
    _lib.call("fqss_sample_norm", _p(x), _p(ms), _p(y), B, x.numel() // B, 1 if inverse else 0, _stream())
    return y


# ------------------------------------------------------------------ spectrogram pair (csrc/stft.hip)
_FFT_TABLES = {}


def fft_tables(n_fft, hop, device):
    """(window, twiddles, envelope) of the n_fft-point periodic Hann STFT; the window is torch's own (bit-identical values),
    the twiddles exp(-2 pi i k / N) are rounded from fp64"""
    key = (n_fft, hop, str(device))
    t = _FFT_TABLES.get(key)
    if t is None:
        import numpy as np
        win = torch.hann_window(n_fft)
        ang = -2.0 * np.pi * np.arange(n_fft // 2, dtype=np.float64) / n_fft
        tw = torch.from_numpy(np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32))
        env = (win * win).view(n_fft // hop, hop).sum(0)
        t = _FFT_TABLES[key] = (win.to(device), tw.contiguous().to(device), env.contiguous().to(device))
    return t


def transpose2d(x):
    """x [..., R, C] dense -> [..., C, R] dense"""
    _need_gpu(x)
    x = x.contiguous()
    R, C = x.shape[-2], x.shape[-1]
    y = torch.empty(*x.shape[:-2], C, R, device=x.device, dtype=torch.float32)
    _lib.call("fqss_transpose2d", _p(x), _p(y), x.numel() // (R * C), R, C, _stream())
    return y


def stft(x, n_fft, hop, T, pad):
    """x [rows, L] -> z [rows, 2, T, n_fft/2] (re / im planes, Nyquist bin dropped), `_spec` framing (htdemucsq.py:931-950)"""
    _need_gpu(x)
    x, rows, L, ld = as_rowmat(x)
    win, tw, _ = fft_tables(n_fft, hop, x.device)
    z = torch.empty(rows, 2, T, n_fft // 2, device=x.device, dtype=torch.float32)
    _lib.call("fqss_stft", _p(x), _p(z), _p(win), _p(tw), rows, L, ld, n_fft, hop, T, pad, _stream())
    return z


def istft(z, n_fft, hop, pad, length):
    """z [rows, 2, T, n_fft/2] -> y [rows, length]  (`_ispec`, htdemucsq.py:952-960)"""
    _need_gpu(z)
    z = z.contiguous()
    rows, _, T, half = z.shape
    assert half == n_fft // 2
    win, tw, env = fft_tables(n_fft, hop, z.device)
    frames = torch.empty(rows, T, n_fft, device=z.device, dtype=torch.float32)
    y = empty_act((rows, length), z.device)
    _lib.call("fqss_istft", _p(z), _p(frames), _p(y), _p(win), _p(env), _p(tw), rows, length, rowmat(y)[2], n_fft, hop, T, pad,
              _stream())
    return y


def istft_bwd(g, n_fft, hop, pad, T):
    _need_gpu(g)
    g, rows, length, ld = as_rowmat(g)
    win, tw, env = fft_tables(n_fft, hop, g.device)
    gz = torch.empty(rows, 2, T, n_fft // 2, device=g.device, dtype=torch.float32)
    _lib.call("fqss_istft_bwd", _p(g), _p(gz), _p(win), _p(env), _p(tw), rows, length, ld, n_fft, hop, T, pad, _stream())
    return gz


def hd_kd_loss(est, fest, src, weights, kd_lambda, want_grad=True):
    """htdemucs solver loss (solver.py:333-366): est / fest / src [B, S, C, T] -> (loss [1], task [S], kd [S], w [B, S], dloss/dest)"""
    _need_gpu(est, fest, src, weights)
    est, fest, src = est.contiguous(), fest.contiguous(), src.contiguous()
    B, S = est.shape[0], est.shape[1]
    N = est.numel() // (B * S)
    assert fest.shape == est.shape and src.shape == est.shape and weights.numel() == S
    sums = torch.zeros(B * S * 5, device=est.device, dtype=torch.float64)
    out = torch.empty(1 + 2 * S + B * S, device=est.device, dtype=torch.float32)
    coef = torch.empty(B * S * 2, device=est.device, dtype=torch.float32)
    g = torch.empty_like(est) if want_grad else None
    _lib.call("fqss_hd_kd_loss", _p(est), _p(fest), _p(src), _p(weights), _p(sums), _p(out), _p(coef), _p(g), B, S, N, float(kd_lambda), _stream())
    return out[:1], out[1:1 + S], out[1 + S:1 + 2 * S], out[1 + 2 * S:].view(B, S), g


# ------------------------------------------------------------------ evaluation side (csrc/infer.hip)
def sisnr_matrix(est, ref, want_map=False):
    """est, ref [S, L] -> SI-SNR matrix [S, S] in dB (est row x ref row) and, optionally, the swap_channel_order map [S, 2]"""
    _need_gpu(est, ref)
    est, S, L, ld_e = as_rowmat(est)
    ref, S2, L2, ld_r = as_rowmat(ref)
    assert (S, L) == (S2, L2), "sisnr_matrix: shape mismatch"
    mom = torch.zeros(S * S * 5, device=est.device, dtype=torch.float64)
    db = torch.empty(S, S, device=est.device, dtype=torch.float32)
    mp = torch.empty(S, 2, device=est.device, dtype=torch.int32) if want_map else None
    _lib.call("fqss_sisnr_matrix", _p(est), _p(ref), _p(mom), _p(db), _p(mp), S, L, ld_e, ld_r, _stream())
    return (db, mp) if want_map else db


def infer_ola(chunk, mp, out, sum_weight, start, n, seg):
    """out[d, c, start:start+n] += w * sign_d * chunk[src_d, c, :n]; sum_weight[start:start+n] += w  (triangular w over `seg`)"""
    _need_gpu(chunk, out, sum_weight)
    S, C = out.shape[0], (out.shape[1] if out.dim() == 3 else 1)
    assert chunk.is_contiguous() and out.is_contiguous() and sum_weight.is_contiguous() and chunk.numel() == S * C * chunk.shape[-1]
    _lib.call("fqss_infer_ola", _p(chunk), _p(mp), _p(out), _p(sum_weight), S, C, n, seg, start, chunk.shape[-1], out.shape[-1], _stream())


def infer_normalize(out, sum_weight):
    _need_gpu(out, sum_weight)
    assert out.is_contiguous()
    L = out.shape[-1]
    _lib.call("fqss_infer_normalize", _p(out), _p(sum_weight), out.numel() // L, L, L, _stream())


# ------------------------------------------------------------------ affine form of the quantizers (csrc/export_q.hip)
def fq_affine(x, scale, zero_point, axis, qmin, qmax, want_codes=False):
    """torch.fake_quantize_per_(tensor|channel)_affine on the device; scale [C] fp32, zero_point [C] int32 (C = 1: per tensor, axis
    ignored) -> y (and the int32 codes when want_codes)"""
    _need_gpu(x, scale)
    x = x.contiguous()
    C = scale.numel()
    if C == 1:
        outer, inner = 1, x.numel()
    else:
        outer, C2, inner = _w_layout(x.shape, axis)
        assert C2 == C, "fq_affine: scale does not match the channel axis"
    y = torch.empty_like(x)
    codes = torch.empty(x.shape, device=x.device, dtype=torch.int32) if want_codes else None
    _lib.call("fqss_fq_affine", _p(x), _p(y), _p(codes), outer, C, inner, _p(scale.contiguous()), _p(zero_point.contiguous()), int(qmin), int(qmax),
              _stream())
    return (y, codes) if want_codes else y


# ------------------------------------------------------------------ data side (csrc/data_ops.hip)
def snr_mix(a, b, snr, mode=0, clip=True):
    """rows of a / b [B, T] mixed at snr [B] dB: generate_2mix_snr (mode 0) / generate_mix_noise (mode 1) + max_clip (process.py:57-103)"""
    _need_gpu(a, b, snr)
    a, B, T, ld_a = as_rowmat(a)
    b, B2, T2, ld_b = as_rowmat(b)
    assert (B, T) == (B2, T2) and snr.numel() == B
    ws = torch.zeros(2 * B, device=a.device, dtype=torch.float64)
    peak = torch.zeros(B, device=a.device, dtype=torch.int32)
    out = torch.empty(B, T, device=a.device, dtype=torch.float32)
    _lib.call("fqss_snr_mix", _p(a), _p(b), _p(snr.contiguous()), _p(ws), _p(peak), _p(out), B, T, ld_a, ld_b, T, mode, 1 if clip else 0, _stream())
    return out


# ------------------------------------------------------------------ data side: polyphase sinc resampler (librimix_dataset.py:54)
_RESAMPLE_TAPS = {}


def sinc_resample_taps(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """(taps [new][2 * width + orig] fp64, width, orig, new) with the rates reduced by their gcd -- torchaudio's published
    `_get_sinc_resample_kernel` (sinc_interp_hann), restated; computed on the host in fp64"""
    import math
    import numpy as np
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    taps = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * scale
    return taps, width, orig, new


def resample(x, orig_freq, new_freq):
    """x [..., L] fp32 on the device -> [..., ceil(new * L / orig)] (torchaudio.transforms.Resample semantics)"""
    _need_gpu(x)
    if int(orig_freq) == int(new_freq):
        return x
    key = (int(orig_freq), int(new_freq), x.device)
    ent = _RESAMPLE_TAPS.get(key)
    if ent is None:
        taps, width, orig, new = sinc_resample_taps(orig_freq, new_freq)
        ent = _RESAMPLE_TAPS[key] = (torch.from_numpy(taps).to(torch.float32).contiguous().to(x.device), width, orig, new)
    h, width, orig, new = ent
    L = x.shape[-1]
    rows = x.numel() // L
    xs = x.reshape(rows, L).contiguous()
    Lout = (new * L + orig - 1) // orig
    y = torch.empty(rows, Lout, device=x.device, dtype=torch.float32)
    _lib.call("fqss_resample_fir", _p(xs), _p(h), _p(y), rows, L, Lout, L, Lout, orig, new, width, _stream())
    return y.reshape(*x.shape[:-1], Lout)
