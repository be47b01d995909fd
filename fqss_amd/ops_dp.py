"""Autograd glue of the dual-path layers (DPTNet): row-major linears, LayerNorm rows, the attention
core with its four input-side quantizers, the bidirectional LSTM, and the chunking data movement.  Like fqss_amd/ops.py:
torch is memory / streams / autograd graph only, every op is a HIP kernel behind the C ABI (fqss_amd/kernels.py).

The output fake-quantizer of a layer is applied as its own node (`ops.NlActQ`, the same fqss_actq_fwd/bwd launches the
ConvTasNet path uses un-fused); inside the dual-path blocks tensors are sequence-first row matrices [L, B', C]."""
import math

import torch
from torch.autograd import Function

from . import kernels as K
from . import ops


def _param_grad(param, like):
    """(buffer, direct) -- accumulate straight into the arena gradient when the tensor is a leaf parameter that has one"""
    if isinstance(param, torch.nn.Parameter):
        return ops._grad_buf(param, like)
    return torch.zeros_like(like), False


def touch(*ts):
    """start the Adam clock of the leaf parameters behind these tensors (the parameter itself or a view of it): the arena
    only steps parameters that were marked as having received a gradient (torch.optim.Adam: `grad is not None`)"""
    for t in ts:
        if t is None:
            continue
        base = t._base if (t._is_view() and t._base is not None) else t
        if isinstance(base, torch.nn.Parameter):
            base._fqss_touched = True


QROW_BWD = __import__("os").environ.get("FQSS_QROW_BWD", "1") != "0"    # gradient GEMMs of coded linears on the codes; 0: fp32 x fp32 (A/B, tests)


def _rowlinear_dgrad(gz, w):
    """dL/dx of z = x @ w^T: from the weight's int8 codes when it carries them (three bf16 products per k), else fp32 x fp32 (six)"""
    wc = getattr(w, "_fqss_wcodes", None)
    if QROW_BWD and wc is not None and w.dim() == 2 and wc.idx.is_contiguous() and K.qrow_bwd_ok(wc.Ci, wc.Co):
        return K.qrow_bwd_x(gz, wc)
    return K.rowlin_bwd_x(gz, w)


WGRAD_BIAS = __import__("os").environ.get("FQSS_WGRAD_BIAS", "1") != "0"    # the bias gradient out of the coded weight-gradient launch


def _rowlinear_wgrad_into(gz, x, xq, buf, gbias=None, defer=False):
    """buf [Co, Ci] += gz^T x: from the activation's u8 codes when the forward ran on them.  gbias (optional, [Co]): the linear's bias
    gradient buffer -- the coded launch adds the column sums of gz into it on the side; returns True when it did.
    defer: buf (and gbias) are accumulation buffers that outlive this backward node (the step's dL/dW_q arena, a parameter's own
    gradient): with a runtime.QuantTables active the launch is queued and runs with the segment's other weight gradients
    (kernels.RowWgradQueue: one grouped launch instead of one 256-workgroup launch per linear)"""
    if QROW_BWD and xq is not None and buf.dim() == 2 and buf.is_contiguous() and xq.idx.is_contiguous() \
            and xq.idx.shape[-1] == buf.shape[1] and K.qrow_bwd_ok(buf.shape[1], buf.shape[0]):
        with_bias = WGRAD_BIAS and gbias is not None and gbias.is_contiguous() and gbias.numel() == buf.shape[0]
        rq = getattr(ops.DEFER, "row_wgrad_queue", None) if defer else None
        if rq is not None:
            rq.push(gz, xq.idx, xq.qmin, xq.qmax, buf, gbias if with_bias else None)
        else:
            K.qrow_bwd_w(gz, xq.idx, xq.qmin, xq.qmax, buf, gbias if with_bias else None)
        return with_bias
    K.rowlin_bwd_w(gz, x, buf)
    return False


LSTM_WIH_GROUP = __import__("os").environ.get("FQSS_LSTM_WIH_GROUP", "1") != "0"


def _rowlinear_wgrad_pair_into(gz0, gz1, x, xq, buf0, buf1):
    """both directions' W_ih gradients of a bidirectional LSTM (two column blocks of dG against one input): one launch"""
    if QROW_BWD and xq is not None and all(b.dim() == 2 and b.is_contiguous() for b in (buf0, buf1)) and xq.idx.is_contiguous() \
            and xq.idx.shape[-1] == buf0.shape[1] and K.qrow_bwd_ok(buf0.shape[1], buf0.shape[0]):
        rq = getattr(ops.DEFER, "row_wgrad_queue", None) if LSTM_WIH_GROUP else None
        if rq is not None and gz0.data_ptr() % 16 == 0 and gz1.data_ptr() % 16 == 0:
            # (buf0 / buf1 are the step's dL/dW_q arena slots: the two weight gradients join the segment's grouped launch)
            rq.push(gz0, xq.idx, xq.qmin, xq.qmax, buf0)
            rq.push(gz1, xq.idx, xq.qmin, xq.qmax, buf1)
        else:
            K.qrow_bwd_w_pair(gz0, gz1, xq.idx, xq.qmin, xq.qmax, buf0, buf1)
    else:
        K.rowlin_bwd_w_pair(gz0, x, buf0, gz1, x, buf1)


class RowLinear(Function):
    """z = x @ w^T + bias on the last dim -- F.linear of LinearQ / the MHA projections / the 1x1 Conv2dQ (qat_layers.py:521-536,
    889-901, 941).  w is the (possibly fake-quantized) weight [Co, Ci]; bias a parameter or None."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.bias = bias
        touch(w, bias)
        return K.rowlin_fwd(x, w, bias)

    @staticmethod
    def backward(ctx, gz):
        x, w = ctx.saved_tensors
        gz = gz.contiguous()
        xq = getattr(ctx, "xq", None)           # RowLinearQ: the codes the forward multiplied
        gx = _rowlinear_dgrad(gz, w) if ctx.needs_input_grad[0] else None
        gw = None
        gb, gb_direct, gb_done = None, True, False
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            gb, gb_direct = _param_grad(ctx.bias, ctx.bias)
        gwq = getattr(w, "_fqss_gwq", None)     # weight fake-quantized by runtime.QuantTables (no autograd history): dL/dW_q goes
        if gwq is not None:                     # into the step's arena, consumed by fqss_wq_multi_bwd
            gb_done = _rowlinear_wgrad_into(gz, x, xq, gwq, gb if gb_direct else None, defer=True)
        elif ctx.needs_input_grad[1]:
            gw, direct = _param_grad(w, w)
            gb_done = _rowlinear_wgrad_into(gz, x, xq, gw, gb)
            gw = None if direct else gw
        if gb is not None and not gb_done:
            K.colsum(gz, gb)
        return gx, gw, (None if gb_direct else gb)


def _rowlinear_wgrad(ctx_needs_w, x, w, gz, xq=None):
    """dL/dW of z = x @ w^T: into the step's dL/dW_q arena when w was fake-quantized by runtime.QuantTables, else the autograd way"""
    gwq = getattr(w, "_fqss_gwq", None)
    if gwq is not None:
        _rowlinear_wgrad_into(gz, x, xq, gwq, defer=True)
        return None
    if ctx_needs_w:
        gw, direct = _param_grad(w, w)
        _rowlinear_wgrad_into(gz, x, xq, gw)
        return None if direct else gw
    return None


class RowLinearActQ(Function):
    """fq(act(x @ w^T + bias)) -- LinearQ / LinearNlQ (ReLU, PReLU) in the quantizing phase as ONE autograd node: the forward is the row
    GEMM (on codes when both operands carry them) + the quantizer pass; the backward runs the quantizer's STE, its range partials AND
    the bias gradient (column sums) in one pass over the gradient (fqss_actq_bwd_colbias), then the two GEMMs -- no fqss_colsum pass."""

    @staticmethod
    def forward(ctx, x, w, bias, slope, qmin, qmax, act, q, qops, slope_param, flat):
        touch(w, bias)
        if qops is not None and FUSE_QROWQ and q.qmode == ops.Q_QUANT and q.no_codes:
            # the output quantizer rides in the int8 GEMM's epilogue: no pass over z (fqss_qrow_fwdq)
            z, y = K.qrow_fwdq(qops[0].idx, qops[1], bias, qops[0].qmin, qops[0].qmax, act, slope, q.qmin, q.qmax)
            q.carrier = False
        else:
            z = K.qrow_fwd(qops[0].idx, qops[1], bias, qops[0].qmin, qops[0].qmax) if qops is not None else K.rowlin_fwd(x, w, bias)
            y = ops._epilogue_fwd(z.view(flat) if flat is not None else z, act, slope, q).view(z.shape)     # long rows for the streaming pass
        ctx.save_for_backward(x, w, z, slope)
        ctx.bias, ctx.q, ctx.act, ctx.sp = bias, q, act, slope_param
        ctx.xq = qops[0] if qops is not None else None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, z, slope = ctx.saved_tensors
        q = ctx.q
        gb, gb_direct = _param_grad(ctx.bias, ctx.bias)
        gz = K.actq_bwd_colbias(z, g.contiguous(), ctx.act, slope, q.qmode, q.qmin, q.qmax, q.gacc, gb)
        g_slope, g_min, g_max = ops._flush_ranges(q, slope, ctx.sp, ctx.act)
        gx = _rowlinear_dgrad(gz, w) if ctx.needs_input_grad[0] else None
        gw = _rowlinear_wgrad(ctx.needs_input_grad[1], x, w, gz, ctx.xq)
        return gx, gw, (None if gb_direct else gb), g_slope, g_min, g_max, None, None, None, None, None


class RowLinearNlQ2(Function):
    """fq2(relu(fq1(x @ w^T + bias))) -- a LinearQ followed by NlQ(ReLU) (the feed-forward block of the Sepformer layer, sepformerq.py:64)
    with both quantizers in their quantizing phase and both GEMM operands on codes: ONE launch forward (int8 GEMM, both quantizers in
    its epilogue, z + y + the codes of y) and one pass + two GEMMs backward.  The fp32 image of fq1(z) is never stored."""

    @staticmethod
    def forward(ctx, x, w, bias, qmin1, qmax1, qmin2, qmax2, q1, q2, qops):
        touch(w, bias)
        z, y, yc = K.qrow_fwdq2(qops[0].idx, qops[1], bias, qops[0].qmin, qops[0].qmax, qmin1, qmax1, qmin2, qmax2)
        ctx.save_for_backward(x, w, z, qmin1, qmax1, qmin2, qmax2)
        ctx.bias, ctx.q1, ctx.q2, ctx.xq = bias, q1, q2, qops[0]
        q2.idx = yc
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, z, qmin1, qmax1, qmin2, qmax2 = ctx.saved_tensors
        q1, q2 = ctx.q1, ctx.q2
        gb, gb_direct = _param_grad(ctx.bias, ctx.bias)
        gz = K.actq2_bwd_colbias(z, g.contiguous(), qmin1, qmax1, qmin2, qmax2, q1.gacc, q2.gacc, gb)
        g_min1, g_max1 = ops._ranges_after(q1, q1.gacc)
        g_min2, g_max2 = ops._ranges_after(q2, q2.gacc)
        gx = _rowlinear_dgrad(gz, w) if ctx.needs_input_grad[0] else None
        gw = _rowlinear_wgrad(ctx.needs_input_grad[1], x, w, gz, ctx.xq)
        return gx, gw, (None if gb_direct else gb), g_min1, g_max1, g_min2, g_max2, None, None, None


QROW = __import__("os").environ.get("FQSS_QROW", "1") != "0"    # student linears on codes (csrc/qrow.hip); 0: fp32-equivalent GEMM
FUSE_QROWQ = __import__("os").environ.get("FQSS_FUSE_QROWQ", "1") != "0"    # their output quantizer in the GEMM epilogue (fqss_qrow_fwdq)


def qrow_operands(x, w):
    """(activation codes, weight codes) when the int8 q-GEMM applies to z = x @ w^T: x carries the u8 codes of the quantizer that
    produced it, w the int8 codes of its fake-quantizer (both set in the quantizing phase only), rows dense and 16-B aligned"""
    if not QROW:
        return None
    # row layers tag their outputs with `_fqss_rowq` (kept apart from `_fqss_q`, which would route element-wise consumers onto
    # ConvTasNet's coded kernels); NlQ outputs arrive through the generic tag
    xq = getattr(x, "_fqss_rowq", None) or ops.codes_of(x)
    wc = getattr(w, "_fqss_wcodes", None)
    if xq is None or wc is None or ops.is_carrier(x) or xq.idx.shape != x.shape or not xq.idx.is_contiguous():
        return None
    if w.dim() != 2 or wc.Ci != x.shape[-1] or not K.qrow_eligible(wc.Ci):
        return None
    return xq, wc


class RowLinearQ(Function):
    """RowLinear whose forward runs on the operands' codes (exact integer sums, int8 MFMA); the backward is RowLinear's"""

    @staticmethod
    def forward(ctx, x, w, bias, xq, wc):
        ctx.save_for_backward(x, w)
        ctx.bias, ctx.xq = bias, xq
        touch(w, bias)
        return K.qrow_fwd(xq.idx, wc, bias, xq.qmin, xq.qmax)

    @staticmethod
    def backward(ctx, gz):
        return RowLinear.backward(ctx, gz) + (None, None)


def row_linear(x, w, bias):
    """z = x @ w^T + bias on the last dim: on codes when both operands carry them, else the fp32-equivalent row GEMM"""
    ops_ = qrow_operands(x, w)
    if ops_ is not None:
        return RowLinearQ.apply(x, w, bias, ops_[0], ops_[1])
    return RowLinear.apply(ops.real(x), w, bias)


class LayerNormRows(Function):
    """F.layer_norm over the last dim (LayerNormQ, qat_layers.py:455-465), one wavefront per row"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean_rstd = K.layernorm_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, beta, mean_rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean_rstd = ctx.saved_tensors
        gg, d1 = _param_grad(gamma, gamma)
        gb, d2 = _param_grad(beta, beta)
        gx = K.layernorm_bwd(gy, x, gamma, mean_rstd, gg, gb)
        return gx, (None if d1 else gg), (None if d2 else gb), None


class LayerNormRowsQ(Function):
    """fq(F.layer_norm(x)) -- LayerNormQ in the quantizing phase as ONE kernel each way (csrc/dualpath.hip, k_layernorm_fwd/bwd<., true>):
    the pre-quant value is never stored, the backward recomputes it from x and the row statistics and runs the quantizer's STE and
    range-gradient partial sums in the same pass as the LayerNorm backward"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, qmin, qmax, q, want_codes):
        y, q.idx, mean_rstd = K.layernormq_fwd(x, gamma, beta, eps, qmin, qmax, want_codes)
        ctx.save_for_backward(x, gamma, beta, mean_rstd, qmin, qmax)
        ctx.q = q
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean_rstd, qmin, qmax = ctx.saved_tensors
        q = ctx.q
        gg, d1 = _param_grad(gamma, gamma)
        gb, d2 = _param_grad(beta, beta)
        gx = K.layernormq_bwd(g.contiguous(), x, gamma, beta, mean_rstd, gg, gb, qmin, qmax, q.gacc)
        _, g_min, g_max = ops._flush_ranges(q, None, None, ops.ACT_NONE)
        return gx, (None if d1 else gg), (None if d2 else gb), None, g_min, g_max, None, None


class AddLayerNormRows(Function):
    """(y, s) = (LN(a + b) or fq(LN(a + b)), a + b): the float residual add in front of a pre-norm transformer sub-layer fused into its
    LayerNorm / LayerNormQ -- one kernel each way (fqss_add_layernorm_fwd/bwd).  The backward adds the gradient that arrives over the
    residual stream `s` in its epilogue and hands the SAME tensor to both addends: the fork's sum costs no pass of its own.
    q = None: plain LayerNorm (float teacher, or a LayerNormQ outside the quantizing phase is not routed here).
    qs (an AddQ's quantizer context in its quantizing phase, else None): the add is quantized -- y = LN(Q)(fq_s(a + b)), the post-norm
    layers of DPTNet (fqss_addq_layernorm_fwd/bwd); s is then the PRE-quant sum and is not an output anyone else may consume."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps, qmin, qmax, q, want_codes, qs=None, qs_min=None, qs_max=None, rmap=None):
        """rmap (with qs): see kernels.add_layernorm_fwd -- y leaves in the row order of the NEXT layer's layout (layout_map)"""
        s, y, idx, mean_rstd = K.add_layernorm_fwd(a, b, gamma, beta, eps, qmin, qmax, want_codes, qs=None if qs is None else (qs.qmin, qs.qmax),
                                                   rmap=rmap)
        if q is not None:
            q.idx = idx
        ctx.save_for_backward(s, gamma, beta, mean_rstd, qmin, qmax)
        ctx.q, ctx.qs, ctx.rmap = q, qs, rmap
        return y, s

    @staticmethod
    def backward(ctx, gy, gs):
        s, gamma, beta, mean_rstd, qmin, qmax = ctx.saved_tensors
        q, qs = ctx.q, ctx.qs
        gg, d1 = _param_grad(gamma, gamma)
        gb, d2 = _param_grad(beta, beta)
        if gy is None:                       # the normalised branch is unused: only the residual stream carries a gradient
            assert qs is None
            return gs, gs, None, None, None, None, None, None, None, None, None, None, None
        if qs is not None:
            gs = None                        # (the pre-quant sum has no other consumer)
        gx = K.add_layernorm_bwd(gy.contiguous(), gs, s, gamma, beta, mean_rstd, gg, gb, qmin, qmax, q.gacc if q is not None else None,
                                 qs=None if qs is None else (qs.qmin, qs.qmax), gacc_s=None if qs is None else qs.gacc, rmap=ctx.rmap)
        g_min = g_max = gs_min = gs_max = None
        if q is not None:
            _, g_min, g_max = ops._flush_ranges(q, None, None, ops.ACT_NONE)
        if qs is not None:
            _, gs_min, gs_max = ops._flush_ranges(qs, None, None, ops.ACT_NONE)
        return gx, gx, (None if d1 else gg), (None if d2 else gb), None, g_min, g_max, None, None, None, gs_min, gs_max, None


class Unary(Function):
    """tanh / sigmoid (the gated output convs, dptnetq.py:286-287)"""

    @staticmethod
    def forward(ctx, x, kind):
        y = K.unary_fwd(x, kind)
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return K.unary_bwd(g, y, ctx.kind), None


class Permute4(Function):
    """y[i0, i1, i2, :] = x viewed with element strides `sin`; the backward is the inverse move (`sback` over x's own dims)"""

    @staticmethod
    def forward(ctx, x, dims_out, sin, dims_in, sback, dense=True, back_perm=None, pad_fwd=False, pad_bwd=False):
        """dense=False: `sin` are x's own strides (a view of a row-padded buffer is read in place, no contiguous() copy first).
        back_perm: for each of the three input dims the output dim it came from -- the backward then reads a row-padded gradient in
        place through ITS strides instead of a dense copy through `sback`.  pad_fwd / pad_bwd: the output / the gradient handed back
        is a row-padded activation (K.permute4 pad_out) -- the side on which rows of C floats are streamed by element-wise kernels."""
        ctx.dims_in, ctx.sback, ctx.C, ctx.xshape, ctx.back_perm, ctx.pad_bwd = dims_in, sback, x.shape[-1], tuple(x.shape), back_perm, pad_bwd
        return K.permute4(x.contiguous() if dense else x, dims_out, sin, x.shape[-1], dense, pad_out=pad_fwd)

    @staticmethod
    def backward(ctx, g):
        bp = ctx.back_perm
        if bp is not None and g.dim() == 4 and g.stride(-1) == 1 and not g.is_contiguous():
            st = g.stride()
            gx = K.permute4(g, ctx.dims_in, (st[bp[0]], st[bp[1]], st[bp[2]]), ctx.C, dense=False, pad_out=ctx.pad_bwd)
        else:
            gx = K.permute4(g.contiguous(), ctx.dims_in, ctx.sback, ctx.C, pad_out=ctx.pad_bwd)
        return (gx if tuple(gx.shape) == ctx.xshape else gx.view(ctx.xshape)), None, None, None, None, None, None, None, None


PERMUTE_CODES = __import__("os").environ.get("FQSS_PERMUTE_CODES", "1") != "0"    # the u8 codes of a row tensor travel through the layout change


def _codes_along(x, y, dims_out, strides_in):
    """x -> y was a Permute4 of rows of N features; when x carries the u8 codes of the quantizer that made it (`_fqss_rowq`), move them the
    same way (the same kernel on a 4-codes-per-float view: N / 4 floats per row) and tag y -- the linear that follows then runs on codes"""
    xq = getattr(x, "_fqss_rowq", None)
    N = x.shape[-1]
    if not PERMUTE_CODES or xq is None or N % 4 or xq.idx.shape != x.shape or not xq.idx.is_contiguous():
        return y
    idx = K.permute4(xq.idx.view(torch.float32), dims_out, tuple(s // 4 for s in strides_in), N // 4)
    y._fqss_rowq = ops.ActCodes(idx.view(torch.uint8).view(y.shape), xq.qmin, xq.qmax)
    return y


def layout_map(shape, to, B):
    """the row map (kernels.add_layernorm_fwd) of rows_to_cols (to = "cols": shape [K, B*S, N]) / cols_to_rows ("rows": [S, B*K, N])"""
    n0, n1, N = shape
    m = n1 // B
    assert n1 == B * m and to in ("cols", "rows")
    # in row r = (i0 = k or s) * (B * m) + b * m + (i2 = s or k)  ->  out row i2 * (B * n0) + b * n0 + i0
    return ((m, B * n0, N), B, m, 1, n0, B * n0)


def change_layout(x, to, B):
    """rows_to_cols / cols_to_rows by name (the un-fused form of layout_map)"""
    n0, n1, _ = x.shape
    return rows_to_cols(x, B, n1 // B) if to == "cols" else cols_to_rows(x, B, n1 // B)


def rows_to_cols(x, B, S):
    """intra-chunk layout [K, B*S, N] -> inter-chunk layout [S, B*K, N]"""
    Kc, BS, N = x.shape
    assert BS == B * S
    # out[s][b][k] = in[k][b*S + s]
    y = Permute4.apply(x, (S, B, Kc), (N, S * N, BS * N), (Kc, B, S), (N, Kc * N, B * Kc * N))
    return _codes_along(x, y.view(S, B * Kc, N), (S, B, Kc), (N, S * N, BS * N))


def cols_to_rows(x, B, Kc):
    """inter-chunk layout [S, B*K, N] -> intra-chunk layout [K, B*S, N]"""
    S, BK, N = x.shape
    assert BK == B * Kc
    y = Permute4.apply(x, (Kc, B, S), (N, Kc * N, BK * N), (S, B, Kc), (N, S * N, B * S * N))
    return _codes_along(x, y.view(Kc, B * S, N), (Kc, B, S), (N, Kc * N, BK * N))


class Segment(Function):
    """split_feature (dptnetq.py:232-259): f [B, N, T] -> intra-chunk rows [K, B*S, N]"""

    @staticmethod
    def forward(ctx, f, Kc):
        ctx.shape, ctx.Kc = tuple(f.shape), Kc
        return K.dp_segment_fwd(f, Kc)

    @staticmethod
    def backward(ctx, g):
        B, N, T = ctx.shape
        return K.dp_segment_bwd(g, B, N, T, ctx.Kc), None


class MergeStreams(Function):
    """merge_feature (dptnetq.py:261-276) up to its Add: inter-chunk rows [S, B*K, nspk*N] -> a, b [B*nspk, N, Lm]"""

    @staticmethod
    def forward(ctx, o, B, nspk, N, Kc):
        S = o.shape[0]
        ctx.geom = (B, nspk, N, Kc, S)
        return K.dp_merge_fwd(o, B, nspk, N, Kc, S)

    @staticmethod
    def backward(ctx, ga, gb):
        return K.dp_merge_bwd(ga, gb, *ctx.geom), None, None, None, None


class Ola2(Function):
    """overlap_and_add of 2-sample frames, hop 1 (dptnetq.py:140): y [N, 2, L] -> [N, L+1]"""

    @staticmethod
    def forward(ctx, y):
        return K.ola2_fwd(y)

    @staticmethod
    def backward(ctx, g):
        return K.ola2_bwd(g)


FUSE_MHA_PREP = __import__("os").environ.get("FQSS_FUSE_MHA_PREP", "1") != "0"   # q / k / v / div quantizers + the division as one pass each way


class MhaCore(Function):
    """Everything of MultiheadAttentionQ.forward between the in-projection X [L, B, 3E] and the (not yet quantized) heads
    [L, B, E] (qat_layers.py:890-911): the q / k / v quantizers (each observes the WHOLE X, each is used on its own third),
    q / sqrt(head_dim), the `div` quantizer, softmax(q k^T) v.  aqs = (aq_q, aq_k, aq_v, aq_div, aq_attn, aq_softmax) quantizer
    modules or None (float teacher); the last two only observe (their outputs are discarded by the reference)."""

    @staticmethod
    def forward(ctx, X, nh, aqs, *ranges):
        L, B, E3 = X.shape
        E = E3 // 3
        hd = E // nh
        X = X.contiguous()
        ctx.geom = (L, B, E, nh, hd)
        scale = math.sqrt(hd)
        if aqs is None:
            q = K.unary_fwd(X[..., :E], K.UNARY_DIVS, scale)
            heads, stats = K.attn_fwd(q, X[..., E:2 * E], X[..., 2 * E:], L, B, nh)
            ctx.qs = None
            ctx.save_for_backward(X, q, heads, stats)
            return heads
        qs = [a.qctx() for a in aqs[:4]]
        ctx.fused = FUSE_MHA_PREP and all(c.qmode == ops.Q_QUANT for c in qs) and E % 4 == 0
        if ctx.fused:
            # quantizing phase: the three quantizers on the thirds, q / sqrt(head_dim) and the div quantizer in ONE pass (fqss_mha_prep_fwd)
            ctx.coded = K.attn_coded_ok(E, nh)
            prep = K.mha_prep_fwd_c if ctx.coded else K.mha_prep_fwd
            q, kq, vq = prep(X, E, scale, [(c.qmin, c.qmax) for c in qs])
            for i in range(4):
                aqs[i].after_forward(qs[i])
            for i in (4, 5):
                m = aqs[i].next_mode() if hasattr(aqs[i], "next_mode") else ops.Q_BYPASS
                if m == ops.Q_OBSERVE:
                    raise RuntimeError("MultiheadAttentionQ: attn / softmax observers out of step with the q / k / v quantizers")
            if ctx.coded:
                # q, kq, vq are the quantizers' u8 CODES: the attention core runs on them (fqss_attn_long_fwd_c)
                heads, stats = K.attn_long_fwd_c(q, kq, vq, [(qs[3].qmin, qs[3].qmax), (qs[1].qmin, qs[1].qmax), (qs[2].qmin, qs[2].qmax)], nh, False)
            else:
                heads, stats = K.attn_fwd(q, kq, vq, L, B, nh, None, None)
            ctx.qs = qs
            ctx.save_for_backward(X, q, heads, stats, kq, vq)
            return heads
        parts = []
        for i in range(3):
            blk = X[..., i * E:(i + 1) * E]
            if qs[i].qmode == ops.Q_OBSERVE:
                K.minmax(X, qs[i].obs_ws)                       # the observer sees all of X (qat_layers.py:890-900)
                parts.append(blk)
            elif qs[i].qmode == ops.Q_QUANT:
                parts.append(K.actq_fwd(blk, ops.ACT_NONE, None, ops.Q_QUANT, qs[i].qmin, qs[i].qmax, None))
            else:
                parts.append(blk)
            aqs[i].after_forward(qs[i])
        qd = K.unary_fwd(parts[0], K.UNARY_DIVS, scale)
        q = K.actq_fwd(qd, ops.ACT_NONE, None, qs[3].qmode, qs[3].qmin, qs[3].qmax, qs[3].obs_ws) if qs[3].qmode != ops.Q_BYPASS else qd
        aqs[3].after_forward(qs[3])
        # the two quantizers whose outputs the reference throws away: observers only
        obs = [None, None]
        for i in (4, 5):
            m = aqs[i].next_mode() if hasattr(aqs[i], "next_mode") else ops.Q_BYPASS
            obs[i - 4] = aqs[i]._obs_ws if m == ops.Q_OBSERVE else None
        if (obs[0] is None) != (obs[1] is None):
            raise RuntimeError("MultiheadAttentionQ: attn / softmax observers out of step")
        heads, stats = K.attn_fwd(q, parts[1], parts[2], L, B, nh, obs[0], obs[1])
        if obs[0] is not None:
            K.observer_ema(aqs[4].min_range.data, aqs[4].max_range.data, aqs[4]._obs_ws, aqs[4].alpha)
            K.observer_ema(aqs[5].min_range.data, aqs[5].max_range.data, aqs[5]._obs_ws, aqs[5].alpha)
        ctx.qs = qs
        ctx.save_for_backward(X, q, heads, stats, qd, parts[1], parts[2])
        return heads

    @staticmethod
    def backward(ctx, gh):
        L, B, E, nh, hd = ctx.geom
        scale = math.sqrt(hd)
        if ctx.qs is None:
            X, q, heads, stats = ctx.saved_tensors
            gX = torch.empty_like(X)
            gq, gk, gv = K.attn_bwd(q, X[..., E:2 * E], X[..., 2 * E:], heads, gh, stats, L, B, nh)
            gX[..., :E].copy_(K.unary_bwd(gq, None, K.UNARY_DIVS, scale))
            gX[..., E:2 * E].copy_(gk)
            gX[..., 2 * E:].copy_(gv)
            return (gX, None, None)
        if ctx.fused:
            X, q, heads, stats, kq, vq = ctx.saved_tensors
            qs = ctx.qs
            if ctx.coded:
                gq, gk, gv = K.attn_long_bwd_c(q, kq, vq, [(qs[3].qmin, qs[3].qmax), (qs[1].qmin, qs[1].qmax), (qs[2].qmin, qs[2].qmax)], heads, gh,
                                               stats, nh, False)
            else:
                gq, gk, gv = K.attn_bwd(q, kq, vq, heads, gh, stats, L, B, nh)
            gaccs = [c.gacc if c.gacc is not None else torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=X.device) for c in qs]
            gX = K.mha_prep_bwd(X, gq, gk, gv, E, scale, [(c.qmin, c.qmax) for c in qs], gaccs)
            grads = []
            for c, ga in zip(qs, gaccs):
                grads += list(ops._ranges_after(c, ga))
            return (gX, None, None, *grads)
        X, q, heads, stats, qd, kq, vq = ctx.saved_tensors
        qs = ctx.qs
        gq, gk, gv = K.attn_bwd(q, kq, vq, heads, gh, stats, L, B, nh)
        grads = [None] * 8
        # div quantizer, then the division
        if qs[3].qmode == ops.Q_QUANT:
            gq, _, grads[6], grads[7], _ = ops._epilogue_bwd(qd, gq, ops.ACT_NONE, None, None, qs[3])
        gq = K.unary_bwd(gq, None, K.UNARY_DIVS, scale)
        gX = torch.empty_like(X)
        for i, g in enumerate((gq, gk, gv)):
            blk = gX[..., i * E:(i + 1) * E]
            if qs[i].qmode == ops.Q_QUANT:
                _, _, grads[2 * i], grads[2 * i + 1], _ = ops._epilogue_bwd(X[..., i * E:(i + 1) * E], g, ops.ACT_NONE, None, None, qs[i], out=blk)
            else:
                blk.copy_(g)
        return (gX, None, None, *grads)


def _lstm_pack(wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, on_codes):
    """the two directions' matrices side by side, as the kernels take them: wih [8H, I], whh [2, 4H, H], bih [8H], bhh [2, 4H] (memory
    plumbing: four small copies).  A FROZEN module (the float teacher: no gradient anywhere, parameters never rewritten) keeps the
    packed copies on its first parameter, keyed by the parameters' versions; the student's projections on codes never read bih"""
    ps = (wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r)
    frozen = not torch.is_grad_enabled() and all(isinstance(p, torch.nn.Parameter) and not p.requires_grad for p in ps)
    if frozen:
        key = tuple((p.data_ptr(), p._version) for p in ps)
        hit = getattr(wih_f, "_fqss_lstm_pack", None)
        if hit is not None and hit[0] == key:
            return hit[1]
    wih = torch.cat([wih_f, wih_r], 0)
    whh = torch.stack([whh_f, whh_r], 0).contiguous()
    bih = None if on_codes else torch.cat([bih_f, bih_r], 0)
    bhh = torch.stack([bhh_f, bhh_r], 0).contiguous()
    if frozen and not (wih.is_cuda and torch.cuda.is_current_stream_capturing()):      # (never keep memory of a graph's private pool)
        wih_f._fqss_lstm_pack = (key, (wih, whh, bih, bhh))
    return wih, whh, bih, bhh


class LstmBi(Function):
    """Bidirectional single-layer LSTM with zero initial state on sequence-first input [S, B, I] (LSTMQ, qat_layers.py:571-600).
    Weights arrive already fake-quantized (or float): w_ih [2][4H, I], w_hh [2][4H, H], biases are parameters."""

    @staticmethod
    def forward(ctx, x, wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, xq=None, wc_f=None, wc_r=None):
        S, B, I = x.shape
        H = whh_f.shape[1]
        x = x.contiguous()
        on_codes = QROW and xq is not None and wc_f is not None and wc_r is not None and K.qrow_eligible(I) \
            and xq.idx.shape == x.shape and xq.idx.is_contiguous()
        qf, qr = ((xq, wc_f), (xq, wc_r)) if on_codes else (None, None)
        wih, whh, bih, bhh = _lstm_pack(wih_f, whh_f, bih_f, bhh_f, wih_r, whh_r, bih_r, bhh_r, on_codes)
        if qf is not None and qr is not None:
            # quantizing phase: x and both W_ih sit on 8-bit grids -> the projections run on the codes (int8 MFMA), one launch per
            # direction into the two column blocks of `pre`
            pre = torch.empty(S, B, 8 * H, device=x.device, dtype=torch.float32)
            K.qrow_fwd(xq.idx, qf[1], bih_f, xq.qmin, xq.qmax, out=pre[..., :4 * H])
            K.qrow_fwd(xq.idx, qr[1], bih_r, xq.qmin, xq.qmax, out=pre[..., 4 * H:])
        else:
            pre = K.rowlin_fwd(x, wih, bih)                # [S, B, 8H]: both directions' input projections, one GEMM
        hout, gsav, csav = K.lstm_fwd(pre, whh, bhh, S, B, H, save=any(ctx.needs_input_grad))      # (the frozen teacher runs under no_grad: nothing to save)
        ctx.save_for_backward(x, wih, whh, hout, gsav, csav)
        ctx.params = (bih_f, bhh_f, bih_r, bhh_r)
        ctx.weights = (wih_f, whh_f, wih_r, whh_r)
        ctx.xq = xq if on_codes else None
        touch(wih_f, whh_f, wih_r, whh_r)
        return hout

    @staticmethod
    def backward(ctx, gout):
        x, wih, whh, hout, gsav, csav = ctx.saved_tensors
        S, B, I = x.shape
        H = whh.shape[2]
        # (the bias gradients -- column sums of dG, the same for b_ih and b_hh of a direction -- come out of the same launch, added
        #  straight into the four parameters' gradient buffers)
        gbufs = [_param_grad(p, p) for p in ctx.params]
        dG = K.lstm_bwd(gout, whh, gsav, csav, S, B, H, gb4=[g for g, _ in gbufs])      # [S, B, 8H]
        gx = K.rowlin_bwd_x(dG, wih) if ctx.needs_input_grad[0] else None
        # weights fake-quantized by runtime.QuantTables carry no autograd history: their dL/dW_q is accumulated straight into the step's
        # arena slot by the wgrad GEMM of that direction; otherwise one GEMM for both directions' W_ih into a fresh buffer
        slots = [getattr(w, "_fqss_gwq", None) for w in ctx.weights]      # wih_f, whh_f, wih_r, whh_r
        if slots[0] is not None and slots[2] is not None:
            _rowlinear_wgrad_pair_into(dG[..., :4 * H], dG[..., 4 * H:], x, ctx.xq, slots[0], slots[2])      # on the input's codes when the projection ran on them
            gws = [None, None, None, None]
        else:
            gwih = torch.zeros_like(wih)
            K.rowlin_bwd_w(dG, x, gwih)
            gws = [gwih[:4 * H], None, gwih[4 * H:], None]
        ghh = [slots[1] if slots[1] is not None else torch.zeros_like(whh[0]), slots[3] if slots[3] is not None else torch.zeros_like(whh[1])]
        if S > 1:
            # forward direction: dG_f[t] with h_f[t-1];  reverse direction: dG_r[t] with h_r[t+1]
            K.rowlin_bwd_w_pair(dG[1:, :, :4 * H], hout[:-1, :, :H], ghh[0], dG[:-1, :, 4 * H:], hout[1:, :, H:], ghh[1])    # one launch for both
        gws[1] = None if slots[1] is not None else ghh[0]
        gws[3] = None if slots[3] is not None else ghh[1]
        if (slots[0] is None) != (slots[2] is None):       # (mixed: one direction's W_ih in the tables, the other not)
            for i in (0, 2):
                if slots[i] is not None:
                    K.axpby_(slots[i], gws[i], 1.0)
                    gws[i] = None
        gbs = [None if direct else buf for buf, direct in gbufs]
        return gx, gws[0], gws[1], gbs[0], gbs[1], gws[2], gws[3], gbs[2], gbs[3], None, None, None


class GroupNormRows(Function):
    """GroupNorm(1, C) of a dual-path block on a row layout (sepformerq.py:159, 175): statistics over all rows of a sample;
    geom = (RB, X, B): sample of row r = (r % RB) // X"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, geom):
        y, mean_rstd = K.gnrows_fwd(x, gamma, beta, eps, *geom)
        ctx.save_for_backward(x, gamma, mean_rstd)
        ctx.geom, ctx.beta = geom, beta
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean_rstd = ctx.saved_tensors
        gg, d1 = _param_grad(gamma, gamma)
        gb, d2 = _param_grad(ctx.beta, gamma)
        gx = K.gnrows_bwd(gy, x, gamma, mean_rstd, gg, gb, *ctx.geom)
        return gx, (None if d1 else gg), (None if d2 else gb), None, None


class AddBcastRows(Function):
    """x [L, B', C] + p [L, C] (positional encoding broadcast over the sequences, sepformerq.py:117-118)"""

    @staticmethod
    def forward(ctx, x, p):
        return K.bcast_add(x, p)

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.needs_input_grad[0] else None), (K.bcast_sum(g) if ctx.needs_input_grad[1] else None)


# ---- first layers of cfg 5 (HTDemucs) -----------------------------------------------------------------
class Gelu(Function):
    """nn.GELU() (erf form)"""

    @staticmethod
    def forward(ctx, x):
        if K.padded_dense(x) is None:       # (a row-padded activation is mapped where it lies: kernels.unary_fwd)
            x = x.contiguous()
        ctx.save_for_backward(x)
        return K.unary_fwd(x, K.UNARY_GELU)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return K.unary_bwd(g, x, K.UNARY_GELU)


class Glu(Function):
    """nn.GLU(dim=1) on channel-first [B, 2C, M]"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.glu_fwd(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return K.glu_bwd(x, g)


class GluActQ(Function):
    """fq(nn.GLU(dim=1)(x)) on channel-first [B, 2C, M] as ONE pass each way (fqss_gluq_fwd / _bwd): the GLU map of a Conv*NlQ layer
    inside its activation quantizer's pass (quantizing or observer phase; the float modules keep the plain map)"""

    @staticmethod
    def forward(ctx, x, qmin, qmax, q):
        y = K.gluq_fwd(x, q.qmode, q.qmin, q.qmax, q.obs_ws)
        q.carrier, q.idx = False, None
        ctx.save_for_backward(x)
        ctx.q = q
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        q = ctx.q
        quant = q.qmode == ops.Q_QUANT
        gacc = None
        if quant:
            gacc = q.gacc if q.gacc is not None else torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=x.device)
        gx = K.gluq_bwd(x, g, q.qmode, q.qmin, q.qmax, gacc)
        g_min = g_max = None
        if quant:
            g_min, g_max = ops._ranges_after(q, gacc)
        return gx, g_min, g_max, None


class DivEw(Function):
    """torch.div(x1, x2), same shape"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return K.div_fwd(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga, gb = K.div_bwd(g, a, b)
        return (ga if ctx.needs_input_grad[0] else None), (gb if ctx.needs_input_grad[1] else None)


class EmbeddingRows(Function):
    """F.embedding(idx, w): w is the (possibly fake-quantized) table"""

    @staticmethod
    def forward(ctx, w, idx):
        ctx.save_for_backward(idx)
        ctx.w = w
        touch(w)
        return K.embedding_fwd(w.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gw, direct = _param_grad(ctx.w, ctx.w)
        K.embedding_bwd(g, idx, gw)
        return (None if direct else gw), None


# ----------------------------------------------------------------------------------------------
# general convolution geometry (HTDemucs layers): the frame gather and its adjoint
# ----------------------------------------------------------------------------------------------
class FramesGather(Function):
    """x [B, C, H, W] -> frames [B, C*kh*kw, Ho*Wo]; followed by a pointwise GEMM this is nn.Conv1d / nn.Conv2d (groups = 1)"""

    @staticmethod
    def forward(ctx, x, geom, out_hw=None):
        """out_hw: the frame grid of the signal zero-padded at the back (K.frames_gather): the pad is never materialised"""
        ctx.geom, ctx.shape, ctx.out_hw = geom, tuple(x.shape), out_hw
        f, _, _ = K.frames_gather(x, geom, out_hw)
        return f

    @staticmethod
    def backward(ctx, g):
        return K.frames_ola(g, None, ctx.shape, ctx.geom, ctx.out_hw), None, None


class ConvHalo(Function):
    """nn.Conv1d / nn.Conv2d with stride 1 (any kernel / dilation / zero padding, groups = 1) and a WIDE output on a halo-packed signal
    (K.halo_pack, csrc/conv_frames.hip): forward, data gradient and weight gradient are implicit GEMMs that read the packed planes in
    place (k_qgemm<.., IMP>, k_gemm_x3<.., IMP>) -- no frame image, no overlap-add.  The `rewrite` convolutions of the HTDemucs decoder
    layers (hdemucsq.py:303-347).  w [Co, Ci, kh, kw]; wc: the int8 codes of a weight fake-quantized by runtime.QuantTables (three
    products per term) or None (float weight: six, the arithmetic of the frame path's fqss_pwconv_fwd_x3s)."""

    @staticmethod
    def forward(ctx, x4, w, bias, plan, wc):
        g = plan.geom
        Co, Ci = w.shape[0], x4.shape[1]
        xp = K.halo_pack(x4, g.ph, g.pw, plan.Wp, plan.plane_x)
        w2 = None if wc is not None else w.reshape(Co, -1).contiguous()
        z = K.conv2_fwd(xp, plan, Co, wc, w2, bias)
        touch(w)
        ctx.plan, ctx.wc, ctx.bias, ctx.Ci = plan, wc, bias, Ci
        ctx.save_for_backward(xp, w)
        return z

    @staticmethod
    def backward(ctx, gz):
        xp, w = ctx.saved_tensors
        plan, wc, Ci = ctx.plan, ctx.wc, ctx.Ci
        Co, T = w.shape[0], plan.taps
        gzp = K.halo_pack(gz, plan.phg, plan.pwg, plan.Wp, plan.plane_g)
        gx = None
        if ctx.needs_input_grad[0]:
            if wc is not None:      # [Co][Ci * taps] codes regrouped per input channel: the data gradient's weight image (a few KB)
                wcT = wc.idx.view(Co, Ci, T).permute(1, 0, 2).reshape(Ci, Co * T).contiguous()
                gx = K.conv2_bwd_x(gzp, plan, Ci, wcT, wc.dw, None)
            else:
                gx = K.conv2_bwd_x(gzp, plan, Ci, None, None, w.reshape(Co, Ci, T).permute(1, 0, 2).reshape(Ci, Co * T).contiguous())
        gw = gb = None
        gwq = getattr(w, "_fqss_gwq", None)
        if ctx.needs_input_grad[1] or gwq is not None:
            gw = gwq if gwq is not None else ops._wgrad_temp(w)
            K.conv2_bwd_w(gzp, xp, gw.view(Co, -1), plan)
            ops._wgrad_temp_done(gw, gwq)
            if gwq is not None:
                gw = None
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            buf, direct = _param_grad(ctx.bias, ctx.bias)
            K.chan_sum(gzp, buf)            # (the halo holds zeros)
            gb = None if direct else buf
        return gx, gw, gb, None, None


def _frozen_cache(w, key, make):
    """the regrouped image of a weight that cannot change under the caller (no autograd in flight: the frozen float teacher,
    mysystem.py:132-133 -- runtime.KDTrainStep sets requires_grad False on its parameters and nothing but load_state_dict, which bumps the
    version, writes them): built once per (parameter, in-place version) and kept on the parameter -- the student's weights change every
    step and are regrouped inside the step"""
    base = w._base if (w._is_view() and w._base is not None) else w          # (the layer hands a view of its parameter)
    if torch.is_grad_enabled() or base.requires_grad or not isinstance(base, torch.nn.Parameter):
        return make()
    cache = getattr(base, "_fqss_regroup", None)
    if cache is None or cache[0] != base._version:
        cache = (base._version, {})
        base._fqss_regroup = cache
    if key not in cache[1]:
        cache[1][key] = make()
    return cache[1][key]


class ConvPhase(Function):
    """nn.Conv1d / nn.Conv2d strided along one axis with a kernel of T strides (the k8 s4 p2 encoder layers of HTDemucs, hdemucsq.py:72-162)
    as a stride-1 convolution with T taps over the s C PHASE planes of its input (K.PhasePlan / K.phase_pack): the implicit GEMMs of
    ConvHalo on a signal that was moved once -- the frame image is T times that, written and read back; the data gradient comes out on
    the phase planes and K.phase_unpack moves it back (the overlap-add's job), the weight gradient is regrouped with the plan's
    permutation.  w [Co, Ci, kh, kw]; wc: int8 codes of a table-quantized weight or None."""

    @staticmethod
    def forward(ctx, x4, w, bias, pp, wc):
        Co, Ci = w.shape[0], x4.shape[1]
        perm, _ = pp.perms(Ci, x4.device)
        xp = K.phase_pack(x4, pp)
        wcp = w2 = None
        if wc is not None:
            wcp = K.WCodes()
            wcp.idx, wcp.dw = wc.idx.index_select(1, perm), wc.dw
        else:
            w2 = _frozen_cache(w, ("phase", pp.k, pp.s, pp.p), lambda: w.reshape(Co, -1).index_select(1, perm))
        z = K.conv2_fwd(xp, pp.inner, Co, wcp, w2, bias)
        touch(w)
        ctx.pp, ctx.wc, ctx.bias, ctx.xshape = pp, wc, bias, tuple(x4.shape)
        ctx.save_for_backward(xp, w)
        return z

    @staticmethod
    def backward(ctx, gz):
        xp, w = ctx.saved_tensors
        pp, wc = ctx.pp, ctx.wc
        B, Ci, H, W = ctx.xshape
        Co, T, Cs = w.shape[0], pp.T, Ci * pp.s
        perm, inv = pp.perms(Ci, gz.device)
        inner = pp.inner
        gzp = K.halo_pack(gz, inner.phg, inner.pwg, inner.Wp, inner.plane_g)
        gx = None
        if ctx.needs_input_grad[0]:
            if wc is not None:
                wcT = wc.idx.index_select(1, perm).view(Co, Cs, T).permute(1, 0, 2).reshape(Cs, Co * T).contiguous()
                gy = K.conv2_bwd_x(gzp, inner, Cs, wcT, wc.dw, None, raw=True)
            else:
                w2T = w.reshape(Co, -1).index_select(1, perm).view(Co, Cs, T).permute(1, 0, 2).reshape(Cs, Co * T).contiguous()
                gy = K.conv2_bwd_x(gzp, inner, Cs, None, None, w2T, raw=True)
            gx = K.phase_unpack(gy, pp, (B, Ci, H, W))
        gw = gb = None
        gwq = getattr(w, "_fqss_gwq", None)
        if ctx.needs_input_grad[1] or gwq is not None:
            gwt = ops._wgrad_temp(w)           # (phase-ordered columns; under FQSS_DETERMINISTIC=1 cut from the integer-shadow pool)
            gwp = gwt.view(Co, Ci * pp.k)
            K.conv2_bwd_w(gzp, xp, gwp, inner)
            ops._wgrad_temp_done(gwt, None)
            gw = gwq if gwq is not None else torch.zeros_like(w)
            K.axpby_(gw.view(Co, -1), gwp.index_select(1, inv), 1.0)
            if gwq is not None:
                gw = None
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            buf, direct = _param_grad(ctx.bias, ctx.bias)
            K.chan_sum(gzp, buf)
            gb = None if direct else buf
        return gx, gw, gb, None, None


class ConvTrPhase(Function):
    """nn.ConvTranspose1d / 2d strided along one axis with a kernel of T strides (the k8 s4 decoder layers of HTDemucs, hdemucsq.py:261-347),
    the adjoint of ConvPhase: the output's s Co phase planes are a stride-1 T-tap convolution of the (halo-packed) input -- one implicit
    GEMM (fqss_conv2_fwd_x3s with negated tap steps) that writes s Co rows where the frame GEMM wrote k Co -- and K.phase_unpack lays them
    out as the signal, bias added, straight onto the window the caller keeps (`pp` carries the window's start as padding): no frame
    image, no overlap-add.  Backward: the phase planes of the output gradient are the input of ConvPhase's forward (data gradient) and
    of its weight gradient.  w [Ci, Co, kh, kw] float (six-product split: the transposed weights have no coded image)."""

    @staticmethod
    def forward(ctx, x4, w, bias, pp, out_shape):
        B, Ci = x4.shape[0], x4.shape[1]
        Co, T, s = w.shape[1], pp.T, pp.s
        inner = pp.inner
        tt = pp.taps(x4.device).reshape(-1)
        # [ci][co][r][q'] = W[ci][co][t0(r) + s q'], regrouped as rows (co, r) x columns (ci, q')
        w2t = _frozen_cache(w, ("trphase", pp.k, s, pp.p), lambda: w.reshape(Ci, Co, -1).index_select(2, tt).view(Ci, Co, s, T)
                            .permute(1, 2, 0, 3).reshape(Co * s, Ci * T).contiguous())
        xp = K.halo_pack(x4, inner.phg, inner.pwg, inner.Wp, inner.plane_g)
        gy = K.conv2_bwd_x(xp, inner, Co * s, None, None, w2t, raw=True)
        y = K.phase_unpack(gy, pp, out_shape, 0, bias)
        touch(w)
        ctx.pp, ctx.bias, ctx.Ci, ctx.Co = pp, bias, Ci, Co
        ctx.save_for_backward(xp, w)
        return y

    @staticmethod
    def backward(ctx, g):
        xp, w = ctx.saved_tensors
        pp, Ci, Co = ctx.pp, ctx.Ci, ctx.Co
        T, s, inner = pp.T, pp.s, pp.inner
        yp = K.phase_pack(g, pp)                       # [B, Co s, plane_x]: the phase planes of the output gradient
        gx = None
        if ctx.needs_input_grad[0]:
            tt = pp.taps(g.device).reshape(-1)
            w2 = w.reshape(Ci, Co, -1).index_select(2, tt).reshape(Ci, Co * s * T)     # [ci][(co, r, q')]
            gx = K.conv2_fwd(yp, inner, Ci, None, w2, None)
        gw = gb = None
        gwq = getattr(w, "_fqss_gwq", None)
        if ctx.needs_input_grad[1] or gwq is not None:
            _, inv = pp.perms(Co, g.device)
            gwt = ops._wgrad_temp(w)
            gwp = gwt.view(Ci, Co * pp.k)
            K.conv2_bwd_w(xp, yp, gwp, inner)
            ops._wgrad_temp_done(gwt, None)
            gw = gwq if gwq is not None else torch.zeros_like(w)
            K.axpby_(gw.view(Ci, -1), gwp.index_select(1, inv), 1.0)
            if gwq is not None:
                gw = None
        if ctx.bias is not None and ctx.needs_input_grad[2]:
            buf, direct = _param_grad(ctx.bias, ctx.bias)
            B, C2, H, W = g.shape
            K.chan_sum(g.reshape(B, C2, H * W), buf)
            gb = None if direct else buf
        return gx, gw, gb, None, None


class FramesOla(Function):
    """frames [B, C*kh*kw, Ho*Wo] (+ bias [C]) -> y [B, C, H, W]: the overlap-add half of nn.ConvTranspose1d / 2d"""

    @staticmethod
    def forward(ctx, frames, bias, sig_shape, geom, out_hw=None):
        ctx.geom, ctx.bias, ctx.out_hw = geom, bias, out_hw
        return K.frames_ola(frames, bias, sig_shape, geom, out_hw)

    @staticmethod
    def backward(ctx, g):
        gb = None
        if ctx.bias is not None and ctx.needs_input_grad[1]:
            buf, direct = _param_grad(ctx.bias, ctx.bias)
            B, C, H, W = g.shape
            K.chan_sum(g.reshape(B, C, H * W), buf)
            gb = None if direct else buf
        gf, _, _ = K.frames_gather(g, ctx.geom, ctx.out_hw)
        return gf, gb, None, None, None


# ----------------------------------------------------------------------------------------------
# general attention core (HTDemucs transformer): cross attention, batch-first rows, long sequences
# ----------------------------------------------------------------------------------------------
class MhaCoreX(Function):
    """MultiheadAttentionQ.forward between the in-projections and the (not yet quantized) heads for query != key (value is key),
    batch-first or sequence-first rows and any sequence lengths (qat_layers.py:878-911): Xq = in_proj(query) and Xkv = in_proj(key)
    are both the full [.., 3E] projections (the q / k / v quantizers observe ALL of their X, each is used on its own third; in the
    reference Xk and Xv are the same numbers when value is key).  Xkv None: self-attention (one projection).
    aqs = (aq_q, aq_k, aq_v, aq_div, aq_attn, aq_softmax) or None for the float module."""

    @staticmethod
    def forward(ctx, Xq, Xkv, nh, batch_first, aqs, *ranges):
        same = Xkv is None
        Xq = Xq.contiguous()
        Xkv = Xq if same else Xkv.contiguous()
        E = Xq.shape[-1] // 3
        hd = E // nh
        scale = math.sqrt(hd)
        ctx.cfg = (same, E, nh, batch_first, scale)
        srcs = (Xq, Xkv, Xkv)
        if aqs is None:
            q = K.unary_fwd(Xq[..., :E], K.UNARY_DIVS, scale)
            heads, stats = K.attn_long_fwd(q, Xkv[..., E:2 * E], Xkv[..., 2 * E:], nh, batch_first)
            ctx.qs = None
            ctx.save_for_backward(Xq, Xkv, q, heads, stats)
            return heads
        qs = [a.qctx() for a in aqs[:4]]
        ctx.coded = FUSE_MHA_PREP and all(c.qmode == ops.Q_QUANT for c in qs) and E % 4 == 0 and K.attn_coded_ok(E, nh)
        if ctx.coded:
            # quantizing phase: the quantizer chain of MhaCore's fused form emits the u8 CODES of q / k / v (for a cross attention the
            # chain runs on both projections: q from the query's, k / v from the key's) and the core runs on them (fqss_attn_long_fwd_c)
            rng = [(c.qmin, c.qmax) for c in qs]
            qc, kc, vc = K.mha_prep_fwd_c(Xq, E, scale, rng)
            if not same:
                _, kc, vc = K.mha_prep_fwd_c(Xkv, E, scale, rng)
            for i in range(4):
                aqs[i].after_forward(qs[i])
            for i in (4, 5):
                m = aqs[i].next_mode() if hasattr(aqs[i], "next_mode") else ops.Q_BYPASS
                if m == ops.Q_OBSERVE:
                    raise RuntimeError("MultiheadAttentionQ: attn / softmax observers out of step with the q / k / v quantizers")
            heads, stats = K.attn_long_fwd_c(qc, kc, vc, [rng[3], rng[1], rng[2]], nh, batch_first)
            ctx.qs = qs
            ctx.save_for_backward(Xq, Xkv, qc, kc, vc, heads, stats)
            return heads
        parts = []
        for i in range(3):
            blk = srcs[i][..., i * E:(i + 1) * E]
            if qs[i].qmode == ops.Q_OBSERVE:
                K.minmax(srcs[i], qs[i].obs_ws)
                parts.append(blk)
            elif qs[i].qmode == ops.Q_QUANT:
                parts.append(K.actq_fwd(blk, ops.ACT_NONE, None, ops.Q_QUANT, qs[i].qmin, qs[i].qmax, None))
            else:
                parts.append(blk)
            aqs[i].after_forward(qs[i])
        qd = K.unary_fwd(parts[0], K.UNARY_DIVS, scale)
        q = K.actq_fwd(qd, ops.ACT_NONE, None, qs[3].qmode, qs[3].qmin, qs[3].qmax, qs[3].obs_ws) if qs[3].qmode != ops.Q_BYPASS else qd
        aqs[3].after_forward(qs[3])
        obs = [None, None]
        for i in (4, 5):
            m = aqs[i].next_mode() if hasattr(aqs[i], "next_mode") else ops.Q_BYPASS
            obs[i - 4] = aqs[i]._obs_ws if m == ops.Q_OBSERVE else None
        if (obs[0] is None) != (obs[1] is None):
            raise RuntimeError("MultiheadAttentionQ: attn / softmax observers out of step")
        heads, stats = K.attn_long_fwd(q, parts[1], parts[2], nh, batch_first, obs[0], obs[1])
        if obs[0] is not None:
            K.observer_ema(aqs[4].min_range.data, aqs[4].max_range.data, aqs[4]._obs_ws, aqs[4].alpha)
            K.observer_ema(aqs[5].min_range.data, aqs[5].max_range.data, aqs[5]._obs_ws, aqs[5].alpha)
        ctx.qs = qs
        ctx.save_for_backward(Xq, Xkv, q, heads, stats, qd, parts[1], parts[2])
        return heads

    @staticmethod
    def backward(ctx, gh):
        same, E, nh, batch_first, scale = ctx.cfg
        if ctx.qs is not None and ctx.coded:
            Xq, Xkv, qc, kc, vc, heads, stats = ctx.saved_tensors
            qs = ctx.qs
            rng = [(c.qmin, c.qmax) for c in qs]
            gq, gk, gv = K.attn_long_bwd_c(qc, kc, vc, [rng[3], rng[1], rng[2]], heads, gh, stats, nh, batch_first)
            gaccs = [c.gacc if c.gacc is not None else torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=Xq.device) for c in qs]
            if same:
                gXq, gXkv = K.mha_prep_bwd(Xq, gq, gk, gv, E, scale, rng, gaccs), None
            else:
                zq, zk = torch.zeros_like(gq), torch.zeros_like(gk)
                gXq = K.mha_prep_bwd(Xq, gq, zq, zq, E, scale, rng, gaccs)          # (its k / v thirds: gradient 0)
                gXkv = K.mha_prep_bwd(Xkv, zk, gk, gv, E, scale, rng, gaccs)
            grads = []
            for c, ga in zip(qs, gaccs):
                grads += list(ops._ranges_after(c, ga))
            return (gXq, gXkv, None, None, None, *grads)
        if ctx.qs is None:
            Xq, Xkv, q, heads, stats = ctx.saved_tensors
            kq, vq, qs = Xkv[..., E:2 * E], Xkv[..., 2 * E:], None
        else:
            Xq, Xkv, q, heads, stats, qd, kq, vq = ctx.saved_tensors
            qs = ctx.qs
        gq, gk, gv = K.attn_long_bwd(q, kq, vq, heads, gh, stats, nh, batch_first)
        grads = [None] * 8
        if qs is not None and qs[3].qmode == ops.Q_QUANT:
            gq, _, grads[6], grads[7], _ = ops._epilogue_bwd(qd, gq, ops.ACT_NONE, None, None, qs[3])
        gq = K.unary_bwd(gq, None, K.UNARY_DIVS, scale)
        gXq = torch.empty_like(Xq) if same else torch.zeros_like(Xq)
        gXkv = gXq if same else torch.zeros_like(Xkv)
        srcs, dsts = (Xq, Xkv, Xkv), (gXq, gXkv, gXkv)
        for i, g in enumerate((gq, gk, gv)):
            blk = dsts[i][..., i * E:(i + 1) * E]
            if qs is not None and qs[i].qmode == ops.Q_QUANT:
                _, _, grads[2 * i], grads[2 * i + 1], _ = ops._epilogue_bwd(srcs[i][..., i * E:(i + 1) * E], g, ops.ACT_NONE, None, None, qs[i], out=blk)
            else:
                blk.copy_(g)
        return (gXq, None if same else gXkv, None, None, None, *grads)


# ----------------------------------------------------------------------------------------------
# LayerScale / frequency-embedding streams (csrc/hd_ops.hip)
# ----------------------------------------------------------------------------------------------
class ChanScale(Function):
    """x [B, C, M] * s [C]  (LayerScale on channel-first tensors, demucsq.py:36-39)"""

    @staticmethod
    def forward(ctx, x, s):
        s = s.contiguous()
        ctx.save_for_backward(x, s)
        ctx.s_in = s
        return K.chan_op(x, s, 0)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gs = torch.zeros_like(s)
        gx = K.chan_scale_bwd(g, x, s, gs)
        return gx, gs


class ColScale(Function):
    """x [..., C] * s [C]  (LayerScale on channel-last rows: transformer gamma_1 / gamma_2)"""

    @staticmethod
    def forward(ctx, x, s):
        s = s.contiguous()
        ctx.save_for_backward(x, s)
        return K.col_scale_fwd(x, s)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gs = torch.zeros_like(s)
        gx = K.col_scale_bwd(g, x, s, gs)
        return gx, gs


class ChanAdd(Function):
    """x [B, C, M] + e [C]  (the frequency embedding added along batch and time, htdemucsq.py:1063-1068)"""

    @staticmethod
    def forward(ctx, x, e):
        return K.chan_op(x, e.contiguous(), 1)

    @staticmethod
    def backward(ctx, g):
        ge = torch.zeros(g.shape[1], device=g.device, dtype=torch.float32)
        K.chan_sum(g, ge)
        return g, ge


class ScalarMul(Function):
    """x * python scalar (ScaledEmbedding, mul_freq: hdemucsq.py:67-69, htdemucsq.py:1068)"""

    @staticmethod
    def forward(ctx, x, c):
        ctx.c = float(c)
        return K.axpby(x, x, 0.0, sa=ctx.c)

    @staticmethod
    def backward(ctx, g):
        return K.axpby(g, g, 0.0, sa=ctx.c), None


class BatchNormFn(Function):
    """nn.BatchNorm1d / nn.BatchNorm2d on a channel-first [B, C, M] view (csrc/batchnorm.hip); the reference runs the ATen op inside
    BatchNormQ (qat_layers.py:472-486).  Training mode: batch statistics (biased variance for the normalisation, unbiased for the
    running estimate, momentum as nn.BatchNorm does it); eval mode: the running statistics.  The C-sized arithmetic is torch on
    C-element tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, training):
        B, C, M = x.shape
        n = B * M
        dev = x.device
        if training or bn.running_mean is None:
            mom = K.bn_moments(x)
            mean = mom[:, 0] / n
            var = (mom[:, 1] / n - mean * mean).clamp_(min=0.0)             # biased
            if training and bn.track_running_stats and bn.running_mean is not None:
                with torch.no_grad():
                    if bn.num_batches_tracked is not None:
                        bn.num_batches_tracked += 1
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1.0 - m).add_(mean.float(), alpha=m)
                    bn.running_var.mul_(1.0 - m).add_((var * (n / max(n - 1, 1))).float(), alpha=m)
        else:
            mean, var = bn.running_mean.double(), bn.running_var.double()
        invstd = 1.0 / torch.sqrt(var + bn.eps)
        w = weight.double() if weight is not None else torch.ones(C, device=dev, dtype=torch.float64)
        b = bias.double() if bias is not None else torch.zeros(C, device=dev, dtype=torch.float64)
        a = invstd * w
        y = K.bn_apply(x, a.float(), (b - mean * a).float())
        ctx.save_for_backward(x, mean, invstd, w)
        ctx.batch_stats, ctx.n, ctx.has_w, ctx.has_b = bool(training or bn.running_mean is None), n, weight is not None, bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, invstd, w = ctx.saved_tensors
        n = ctx.n
        red = K.bn_bwd_reduce(g, x)
        sg, sgx = red[:, 0], red[:, 1]
        ggamma = invstd * (sgx - mean * sg)                 # sum g xhat
        c1 = w * invstd
        if ctx.batch_stats:
            c2 = -c1 * invstd * ggamma / n
            c3 = c1 * (-sg / n + mean * invstd * ggamma / n)
        else:
            c2, c3 = torch.zeros_like(c1), torch.zeros_like(c1)
        gx = K.bn_bwd_apply(g, x, c1.float(), c2.float(), c3.float())
        return gx, (ggamma.float() if ctx.has_w else None), (sg.float() if ctx.has_b else None), None, None
