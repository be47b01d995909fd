"""LayerQ module library of FQSS on MI355X: each module owns the float op's parameters plus its
fake-quantizers under the reference's attribute names (=> identical state_dict keys) and runs
`fq_act(nl(op(x, fq_w(W))))` as HIP kernels through fqss_amd.ops.

Mirrors quantization/qat/qat_layers.py of the reference: markers Add/Sub/Mul/Div/Const (:8-46),
LayerQ (:49-59), AddQ/SubQ/MulQ (:62-101), Conv1dQ (:124-153), Conv1dNlQ (:188-219),
GroupNormQ (:438-452), NlQ (:511-518), Conv1dEncoderQ (:993-1046), ResidualErrorBlock (:1105-1202),
ConvTr1dDecoderQ (:1305-1361); and, for the dual-path models (DPTNet): LayerNormQ (:455-465),
LinearQ (:521-536), LSTMQ (:571-600), MultiheadAttentionQ (:865-950), Conv2dQ (1x1), LinearDecoderQ (:1256-1296) with the
nn.Linear branch of ResidualErrorBlock (:1178-1187), over the kernels of csrc/dualpath.hip, attn.hip, lstm.hip.
The Sepformer / HTDemucs classes (LinearNlQ, Conv2dNlQ, ConvTranspose*Q, Conv1dGnNlQ, ConvTr2dDecoderQ, ...) and BatchNormQ follow
further down on the kernels of csrc/conv_frames.hip, hd_ops.hip, batchnorm.hip; variants of these layers that no FQSS configuration takes
(see docs/history/DESIGN_rounds_1-5.md 8) raise NotImplementedError at construction -- there is no ATen fallback.
"""
import math
import os

import torch
import torch.nn as nn

from ... import ops, ops_dp
from ... import kernels as K
from .qat_quant import _BypassQuantizer, get_activation_quantizer, get_weight_quantizer


# ---------------------------------------------------------------------------------------------
# float marker modules (forward runs the same HIP kernels in BYPASS mode: the teacher path)
# ---------------------------------------------------------------------------------------------
def _is_row_bcast(x1, x2):
    """[L, B', C] + [L, 1, C]: the positional-encoding add of the Sepformer transformer blocks on sequence-first rows"""
    return torch.is_tensor(x2) and x1.dim() == 3 and x2.dim() == 3 and x2.shape[1] == 1 and x1.shape[1] != 1 \
        and x1.shape[0] == x2.shape[0] and x1.shape[2] == x2.shape[2]


class Add(nn.Module):
    def forward(self, x1, x2):
        if _is_row_bcast(x1, x2):
            return ops_dp.AddBcastRows.apply(ops.real(x1), ops.real(x2).reshape(x2.shape[0], x2.shape[2]))
        return ops.AddActQ.apply(x1, x2, None, None, 1.0, ops.BYPASS)


class Sub(nn.Module):
    def forward(self, x1, x2):
        return ops.AddActQ.apply(x1, x2, None, None, -1.0, ops.BYPASS)


def _mul_special(x1, x2):
    """the products of the HTDemucs layers that are not tensor x tensor: LayerScale (x * scale[:, None] channel-first, x * scale
    channel-last; demucsq.py:36-39) and x * python scalar (ScaledEmbedding, mul_freq); None when x2 is an ordinary operand"""
    if not torch.is_tensor(x2):
        return ops_dp.ScalarMul.apply(ops.real(x1), float(x2))
    if x2.dim() == 2 and x2.shape[1] == 1 and x1.dim() == 3 and x1.shape[1] == x2.shape[0]:
        return ops_dp.ChanScale.apply(ops.real(x1), x2.reshape(-1))
    if x2.dim() == 1 and x1.dim() >= 2 and x1.shape[-1] == x2.shape[0] and x1.shape != x2.shape:
        return ops_dp.ColScale.apply(ops.real(x1), x2)
    return None


def _mul_any(x1, x2, qmin, qmax, q):
    """mask[B,S,C,M] * feat[B,1,C,M] (ConvTasNet masking) or same-shape multiply"""
    if torch.is_tensor(x2) and x1.dim() == 4 and x2.dim() == 4 and x2.shape[1] == 1 and x1.shape[0] == x2.shape[0] \
            and x1.shape[2:] == x2.shape[2:]:
        return ops.MulActQ.apply(x1, x2.squeeze(1), qmin, qmax, q)
    if torch.is_tensor(x2) and x1.dim() == 4 and x2.dim() == 4 and x1.shape[1] == 1 and x1.shape[0] == x2.shape[0] \
            and x1.shape[2:] == x2.shape[2:]:
        return ops.MulActQ.apply(x2, x1.squeeze(1), qmin, qmax, q)     # feat * mask (dptnetq.py:395): same product
    if torch.is_tensor(x2) and x1.shape == x2.shape:
        M = x1.shape[-1]
        y = ops.MulActQ.apply(x1.reshape(1, 1, -1, M), x2.reshape(1, -1, M), qmin, qmax, q)
        return y.reshape(x1.shape)
    raise NotImplementedError(f"MulQ broadcast {tuple(x1.shape)} x {getattr(x2, 'shape', x2)} has no HIP kernel yet")


FUSE_MULQ = os.environ.get("FQSS_FUSE_MULQ", "1") != "0"
FUSE_DECQ = os.environ.get("FQSS_FUSE_DECQ", "1") != "0"    # decoder reads its coded input directly


def _mul_coded(x1, x2, q):
    """mask[B,S,C,M] * feat[B,1,C,M] (either order) with both operands on codes, in the quantizing phase: the codes -> codes kernel
    (ops.MulQCoded); None when it does not apply"""
    if not (FUSE_MULQ and torch.is_tensor(x2) and x1.dim() == 4 and x2.dim() == 4 and q.qmode == ops.Q_QUANT):
        return None
    if q.gacc is None and torch.is_grad_enabled() and (x1.requires_grad or x2.requires_grad):
        return None        # a backward would need the range-gradient slots (inference has none and needs none)
    if x1.shape[1] == 1 and x2.shape[1] != 1:
        x1, x2 = x2, x1
    if not (x2.shape[1] == 1 and x1.shape[0] == x2.shape[0] and x1.shape[2:] == x2.shape[2:] and 1 <= x1.shape[1] <= 4):
        return None
    mq, fq_ = ops.codes_of(x1), ops.codes_of(x2)
    if mq is None or fq_ is None or K.rowmat(mq.idx) is None or K.rowmat(fq_.idx) is None:
        return None
    return ops.MulQCoded.apply(x1, x2, q.qmin, q.qmax, q, mq, fq_)


class Mul(nn.Module):
    def forward(self, x1, x2):
        y = _mul_special(x1, x2)
        return y if y is not None else _mul_any(x1, x2, None, None, ops.BYPASS)


class Div(nn.Module):
    def forward(self, x1, x2):
        return ops_dp.DivEw.apply(ops.real(x1), ops.real(x2))


class Const(nn.Module):
    def __init__(self, shape=None):
        super().__init__()
        self.shape = shape

    def forward(self, x):
        return x


# ---------------------------------------------------------------------------------------------
class LayerQ(nn.Module):
    """common state of every quantized layer: `.activation_fake_quantize`, `.weight_fake_quantize`"""

    def __init__(self, gradient_based=True, weight_quant=False, act_quant=False, act_nl_quantizer=False,
                 weight_shape=(1, 1, 1), ch_out_idx=0, act_n_bits=8, weight_n_bits=8, do_mac_op=False):
        super().__init__()
        self.weight_quant = weight_quant
        self.act_quant = act_quant
        self.gradient_based = gradient_based
        self.activation_fake_quantize = (get_activation_quantizer(gradient_based, n_bits=act_n_bits, nl=act_nl_quantizer)
                                         if act_quant else _BypassQuantizer())
        self.weight_fake_quantize = (get_weight_quantizer(gradient_based, weight_shape, ch_out_idx=ch_out_idx, n_bits=weight_n_bits)
                                     if weight_quant else nn.Identity())
        self.do_mac_op = do_mac_op   # MAC counters are an analysis aid of the reference (never enabled); kept as attrs
        self.mac_op = 0

    def _wq(self, weight):
        return self.weight_fake_quantize(weight)


def _expect(obj, typ, what):
    if not isinstance(obj, typ):
        raise Exception(f"Quantizing wrong layer instead of {what} got:{type(obj)}")


class AddQ(LayerQ):
    def __init__(self, add, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(add, Add, "Add")
        self.add = add

    def forward(self, x1, x2):
        aq = self.activation_fake_quantize
        if _is_row_bcast(x1, x2):
            return fq_node(aq, ops_dp.AddBcastRows.apply(ops.real(x1), ops.real(x2).reshape(x2.shape[0], x2.shape[2])))
        q = aq.qctx()
        q.chain = bool(getattr(self.add, "fqss_chain", False))     # the owner declares: x1 = the previous add's output, consumed here alone
        y = ops.ew_layer(x1, x2, 1.0, ops.ACT_NONE, None, q)
        aq.after_forward(q)
        return ops.tag_codes(y, q)


class SubQ(LayerQ):
    def __init__(self, sub, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(sub, Sub, "Sub")
        self.sub = sub

    def forward(self, x1, x2):
        aq = self.activation_fake_quantize
        q = aq.qctx()
        y = ops.ew_layer(x1, x2, -1.0, ops.ACT_NONE, None, q)
        aq.after_forward(q)
        return ops.tag_codes(y, q)


class MulQ(LayerQ):
    def __init__(self, mul, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(mul, Mul, "Mul")
        self.mul = mul

    def forward(self, x1, x2):
        aq = self.activation_fake_quantize
        y = _mul_special(x1, x2)
        if y is not None:
            return fq_node(aq, y)
        q = aq.qctx()
        y = _mul_coded(x1, x2, q)
        if y is None:
            y = _mul_any(ops.real(x1), ops.real(x2), q.qmin, q.qmax, q)
        aq.after_forward(q)
        if q.idx is not None and q.idx.shape != y.shape:
            q.idx = q.idx.reshape(y.shape)       # same-shape products run on a [1, 1, rows, M] view (same row padding)
        return ops.tag_codes(y, q)


class ConstQ(LayerQ):
    def __init__(self, const, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        self.const = const

    def forward(self, x):
        return self.activation_fake_quantize(ops.real(x))


# ---------------------------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------------------------
def _act_of(nl):
    """(act code, slope parameter) of the fused non-linearity"""
    if nl is None or isinstance(nl, nn.Identity):
        return ops.ACT_NONE, None
    if isinstance(nl, nn.PReLU):
        if nl.weight.numel() != 1:
            raise NotImplementedError("per-channel PReLU has no HIP kernel (the FQSS models use nn.PReLU())")
        return ops.ACT_PRELU, nl.weight
    if isinstance(nl, nn.ReLU):
        return ops.ACT_RELU, None
    raise NotImplementedError(f"non-linearity {type(nl).__name__}: only PReLU/ReLU are on the ConvTasNet path")


def conv1d_geometry(conv):
    """classify an nn.Conv1d into one of the linear kinds served by a HIP kernel"""
    k, s, p, d, g = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0], conv.groups
    if k == 1 and s == 1 and p == 0 and g == 1:
        return ops._Lin("pw")
    if g == conv.in_channels == conv.out_channels and s == 1 and 2 * p == d * (k - 1):
        return ops._Lin("dw", dil=d, pad=p)
    if g == 1 and p == 0 and d == 1 and k > 1 and k % s == 0 and conv.in_channels <= 2 and k in (2, 16, 32):
        return ops._Lin("frames", stride=s)
    if g == 1:
        return ops._Lin("gather", stride=s, dil=d, pad=p)    # general geometry: frame gather + pointwise GEMM (conv_frames)
    raise NotImplementedError(f"Conv1d(k={k}, s={s}, p={p}, d={d}, groups={g}) has no HIP kernel")


def is_depthwise3(m):
    """a Conv1dNlQ / Conv1dQ wrapping a 3-tap depthwise 'same' convolution (the TCN block's dilated conv)"""
    conv = getattr(m, "conv1d", None)
    return (isinstance(m, (Conv1dNlQ, Conv1dQ)) and isinstance(conv, nn.Conv1d) and conv.kernel_size[0] == 3 and conv.stride[0] == 1
            and conv.groups == conv.in_channels == conv.out_channels and conv.padding[0] == conv.dilation[0])


def run_conv1d(conv, x, weight, nl, aq, pad_to=None):
    """fq_act(nl(conv1d(x, weight) + bias)) through one fused autograd node; pad_to (conv_frames): x stands for itself zero-padded on the
    right to that length (general geometry only -- the caller pads for the other kinds)"""
    if isinstance(nl, (nn.Tanh, nn.Sigmoid, nn.GELU, nn.GLU)):
        # gated output convs of the dual-path separators (dptnetq.py:286-287), GELU / GLU convs of the HTDemucs layers
        # (hdemucsq.py:126-127): conv, then the map, then the quantizer
        z = run_conv1d(conv, x, weight, None, None, pad_to)
        return fq_node(aq, apply_map(nl, ops.real(z)))
    L = conv1d_geometry(conv)
    if L.kind == "gather":
        return fq_node(aq, conv_frames(conv, x, weight, pad_to), nl)
    assert pad_to is None or pad_to == x.shape[-1], "pad_to: general-geometry convolutions only"
    act, slope = _act_of(nl)
    L.w_param, L.b_param, L.slope_param = conv.weight, conv.bias, slope
    q = aq.qctx() if aq is not None else ops.BYPASS
    xq, wc = ops.codes_of(x), getattr(weight, "_fqss_wcodes", None)
    if L.kind == "dw" and xq is not None and q.qmode == ops.Q_QUANT and conv.kernel_size[0] <= 8:
        y = ops.DwConvQ.apply(x, weight, conv.bias, slope, q.qmin, q.qmax, L, act, q, xq)
    else:
        ops.flush_defer(x)
        if not (L.kind == "pw" and xq is not None and wc is not None):
            x = ops.real(x)             # no coded-input kernel for this case: decode a carrier first
        y = ops.LinearActQ.apply(x, weight, conv.bias, slope, q.qmin, q.qmax, L, act, q, xq, wc)
    if aq is not None:
        aq.after_forward(q)
    return ops.tag_codes(y, q)


def _pair(v):
    return (1, v[0]) if len(v) == 1 else tuple(v)


def _conv_geom(conv, one_d):
    """ConvGeom of an nn.Conv1d / Conv2d / ConvTranspose1d / 2d (1-D layers run as [B, C, 1, T])"""
    if conv.groups != 1 or conv.padding_mode != "zeros" or isinstance(conv.padding, str):
        raise NotImplementedError(f"{type(conv).__name__}: only groups = 1 with explicit zero padding has a HIP path")
    pad = (0, conv.padding[0]) if one_d else tuple(conv.padding)
    return K.ConvGeom(_pair(conv.kernel_size), _pair(conv.stride) if not one_d else (1, conv.stride[0]), pad,
                      _pair(conv.dilation) if not one_d else (1, conv.dilation[0]))


def conv_frames(conv, x, weight, pad_to=None):
    """nn.Conv1d / nn.Conv2d (groups = 1, any kernel / stride / dilation / zero padding) of the HTDemucs layers (DConv's dilated k3
    convs, the k8 s4 encoders, Conv2d (8,1) along frequency, the 3x3 decoder rewrites; hdemucsq.py:72-162, 261-347,
    demucsq.py:110-182): frame gather (fqss_frames_gather) + the pointwise GEMM kernels over Ci*kh*kw channels; the data
    gradient is the overlap-add (fqss_frames_ola) of the GEMM's.  Returns the float conv output, same rank as x."""
    x = ops.real(x)
    one_d = x.dim() == 3
    geom = _conv_geom(conv, one_d)
    x4 = x.unsqueeze(2) if one_d else x
    B, C, H, W = x4.shape
    Ho, Wo = geom.out_hw(H, W)
    grid = None
    if pad_to is not None and pad_to > W:
        # the input zero-padded on the right to `pad_to` columns (`F.pad(x, (0, stride - le % stride))` in front of the strided encoder
        # convolutions, hdemucsq.py:131-135): the frame gather reads zeros past the signal, so only the frame GRID is the padded one's
        Wo = geom.out_hw(H, pad_to)[1]
        grid = (Ho, Wo)
    Co = conv.out_channels
    L = ops._Lin("pw", b_param=conv.bias, six=True)
    if K.CONV_HALO and K.CONV_PHASE and K.PhasePlan.serves(H, W, geom):
        # strided along one axis with a kernel of T strides (the k8 s4 p2 encoder layers): a stride-1 T-tap convolution over the phase
        # planes of the input (ops_dp.ConvPhase) -- the signal moves once, the frame image would be T times it
        wc = getattr(weight, "_fqss_wcodes_dgrad", None) if (ops.FRAME_CODES_FWD and ops.FRAME_CODES_DGRAD) else None
        pp = K.PhasePlan(H, W, geom, pad_to)
        if pp.ok(C, Co, wc is not None) and (wc is None or (wc.idx.is_contiguous() and wc.Ci == C * pp.k)):
            z = ops_dp.ConvPhase.apply(x4, ops.weight_view(weight, Co, C, geom.kh, geom.kw), conv.bias, pp, wc)
            return z.squeeze(2) if one_d else z
    if grid is not None:
        cols = ops_dp.FramesGather.apply(x4, geom, grid)
    elif geom.args() == (1, 1, 1, 1, 0, 0, 1, 1):
        cols = x4.reshape(B, C, H * W)
    elif one_d and geom.sw == 1 and K.conv1d_s1_ok(C, geom.kw) and geom.dw * (geom.kw - 1) - geom.pw >= 0 and Co <= K.CONV_IMPLICIT_MAX_CO:
        # stride-1 Conv1d with FEW output channels (DConv's dilated k3 convs, C -> C / 8): implicit GEMM, no frame image
        # (fqss_conv1d_s1_*).  Measured per shape (tools/conv1_probe.py): forward 1.3-1.9x, data gradient 2-4x faster than gather + GEMM
        # (+ overlap-add); the wide rewrite convs (C -> 2C) stay on the frame image, whose GEMM kernel is the faster one there
        cols = x4.reshape(B, C, W)
        L = ops._Lin("conv1", dil=geom.dw, pad=geom.pw, b_param=conv.bias, taps=geom.kw)
    else:
        if geom.sh == 1 and geom.sw == 1:
            # stride-1 convolutions with WIDE outputs (the 3 x 3 / k = 3 `rewrite` convs of the decoder layers, C -> 2C): implicit GEMMs on
            # the halo-packed signal (ops_dp.ConvHalo) -- the frame image would be 9 x / 3 x the activation, written and read back
            wc = getattr(weight, "_fqss_wcodes_dgrad", None) if (ops.FRAME_CODES_FWD and ops.FRAME_CODES_DGRAD) else None
            plan = K.HaloPlan(H, W, geom)
            if plan.ok(C, Co, wc is not None) and (wc is None or (wc.idx.is_contiguous() and wc.Ci == C * plan.taps)):
                z = ops_dp.ConvHalo.apply(x4, ops.weight_view(weight, Co, C, geom.kh, geom.kw), conv.bias, plan, wc)    # (keeps the arena slot)
                return z.squeeze(2) if one_d else z
        cols = ops_dp.FramesGather.apply(x4, geom)
    ops_dp.touch(weight)
    w3 = ops.weight_view(weight, Co, -1, 1)
    if L.kind == "pw":
        L.wc_dgrad = getattr(weight, "_fqss_wcodes_dgrad", None)      # runtime.QuantTables: int8 image of the fake-quantized conv weight
    z = ops.LinearActQ.apply(cols, w3, conv.bias, None, None, None, L, ops.ACT_NONE, ops.BYPASS)
    z = z.reshape(B, Co, Ho, Wo)
    return z.squeeze(2) if one_d else z


_OWN = object()


def convtr_frames(convtr, x, weight, bias=_OWN, window=None):
    """nn.ConvTranspose1d / 2d (groups = 1): pointwise GEMM with the [Co*kh*kw, Ci] transposed weight, then the deterministic
    overlap-add fqss_frames_ola (+ bias).  hdemucsq.py:303-347 (`conv_tr`), qat_layers.py:296-435.  Returns the float output.
    window = (dim, start, length), dim -2 | -1: only out[..., start : start + length (, :)] is wanted (the crop behind every decoder
    layer of HTDemucs, hdemucsq.py:340-345) -- the overlap-add writes just that window (a crop at the front is a padding of the transposed
    convolution, one at the back a shorter signal), its adjoint gathers from the window's gradient: no dense copy either way."""
    x = ops.real(x)
    one_d = x.dim() == 3
    geom = _conv_geom(convtr, one_d)
    x4 = x.unsqueeze(2) if one_d else x
    B, Ci, Hi, Wi = x4.shape
    op = (0, convtr.output_padding[0]) if one_d else tuple(convtr.output_padding)
    H = (Hi - 1) * geom.sh - 2 * geom.ph + geom.dh * (geom.kh - 1) + op[0] + 1
    W = (Wi - 1) * geom.sw - 2 * geom.pw + geom.dw * (geom.kw - 1) + op[1] + 1
    if geom.out_hw(H, W) != (Hi, Wi):
        raise ValueError("ConvTranspose: output_padding must be smaller than the stride")
    if window is not None:
        dim, start, length = window
        full = H if dim == -2 else W
        if dim not in (-2, -1) or (one_d and dim == -2) or start < 0 or length < 1 or start + length > full:
            raise ValueError(f"convtr_frames: window {window} outside the output ({H} x {W})")
        if dim == -2:
            geom, H = K.ConvGeom((geom.kh, geom.kw), (geom.sh, geom.sw), (geom.ph + start, geom.pw), (geom.dh, geom.dw)), length
        else:
            geom, W = K.ConvGeom((geom.kh, geom.kw), (geom.sh, geom.sw), (geom.ph, geom.pw + start), (geom.dh, geom.dw)), length
    Co = convtr.out_channels
    if K.CONV_HALO and K.CONV_PHASE and K.PhasePlan.serves(H, W, geom) and op == (0, 0):
        # strided along one axis with a kernel of T strides (the k8 s4 decoder layers): the output's phase planes from one implicit GEMM
        # on the halo-packed input, laid out as the signal -- the kept window of it -- by fqss_phase_unpack (ops_dp.ConvTrPhase)
        pp = K.PhasePlan(H, W, geom, no=(Hi if geom.sh > 1 else Wi))
        if pp.ok(Co, Ci, False) and pp.inner.Ho == (Hi if pp.axis == 0 else 1) and pp.inner.Wo == (Wi if pp.axis == 1 else W):
            b_ = convtr.bias if bias is _OWN else bias
            y = ops_dp.ConvTrPhase.apply(x4, ops.weight_view(weight, Ci, Co, geom.kh, geom.kw), b_, pp, (B, Co, H, W))
            return y.squeeze(2) if one_d else y
    ops_dp.touch(weight)
    wt = weight.reshape(Ci, -1).t().contiguous().unsqueeze(-1)           # [Co*kh*kw, Ci, 1]: a transposing copy of the (small) weight
    gwq = getattr(weight, "_fqss_gwq", None)
    if gwq is not None:
        # weight fake-quantized by runtime.QuantTables (no autograd history): the copy's gradient is transposed back into the
        # weight's dL/dW_q arena slot by the GEMM node's backward
        gwt = torch.zeros_like(wt)
        wt._fqss_gwq = gwt
        wt._fqss_gwq_done = lambda: K.axpby_(gwq.reshape(Ci, -1), K.transpose2d(gwt.reshape(-1, Ci)), 1.0)
    frames = ops.LinearActQ.apply(x4.reshape(B, Ci, Hi * Wi), wt, None, None, None, None, ops._Lin("pw", six=True), ops.ACT_NONE, ops.BYPASS)
    y = ops_dp.FramesOla.apply(frames, convtr.bias if bias is _OWN else bias, (B, Co, H, W), geom, (Hi, Wi))
    return y.squeeze(2) if one_d else y


def _add_after(add_layer, other, like_shape):
    """(codes of `other`, the AddQ's quantizer) when `add_layer(other, <conv output>)` can run in the conv's GEMM epilogue: a
    quantizing AddQ on a coded operand of the output's shape; else None (the AddQ then launches its own kernel)"""
    if other is None or type(add_layer) is not AddQ:
        return None
    aq = add_layer.activation_fake_quantize
    if not hasattr(aq, "next_mode") or (aq.observer_mode and aq.n_iter < aq.max_observations):
        return None
    oq = ops.codes_of(other)
    if oq is None or tuple(oq.idx.shape) != tuple(like_shape):
        return None
    return oq, aq


def run_conv1d_pair(l1, l2, x, sole_ew_consumers=False, adds=None):
    """(l1(x), l2(x)) for two Conv1dQ layers fed by the same tensor, as ONE fused node (ops.LinearActQPair);
    None when the fused path does not apply (observer phase, eager mode, float layers): the caller then runs
    the two layers one by one.  sole_ew_consumers: the caller guarantees that each output is consumed by exactly one
    element-wise LayerQ (AddQ), whose backward kernel may then run this layer's output-quantizer backward.
    adds = ((other1, add_layer1), (other2, add_layer2)) (entries may be None; needs sole_ew_consumers): the caller will call
    add_layer_i(other_i, output_i) next -- the forward of that AddQ then runs in this GEMM's epilogue."""
    if ops.DEFER is None or type(l1) is not Conv1dQ or type(l2) is not Conv1dQ:
        return None
    pre = getattr(l1.conv1d.weight, "_fqss_wq", None)
    pair = getattr(pre, "_fqss_pair", None)
    if pair is None or pair.partner is not getattr(l2.conv1d.weight, "_fqss_wq", None):
        return None
    aqs = (l1.activation_fake_quantize, l2.activation_fake_quantize)
    if any(not hasattr(a, "next_mode") or (a.observer_mode and a.n_iter < a.max_observations) for a in aqs):
        return None
    if l1.weight_fake_quantize.observer_mode or l2.weight_fake_quantize.observer_mode:
        return None
    xq = ops.codes_of(x)
    if xq is None:
        return None
    q1, q2 = aqs[0].qctx(), aqs[1].qctx()
    L1, L2 = conv1d_geometry(l1.conv1d), conv1d_geometry(l2.conv1d)
    for L, conv in ((L1, l1.conv1d), (L2, l2.conv1d)):
        L.w_param, L.b_param, L.slope_param = conv.weight, conv.bias, None
    after = None
    if adds is not None and sole_ew_consumers and ops.FUSE_ADD_FWD:
        after = tuple(None if a is None else _add_after(a[1], a[0], (xq.idx.shape[0], conv.out_channels, xq.idx.shape[2]))
                      for a, conv in zip(adds, (l1.conv1d, l2.conv1d)))
        if all(a is None for a in after):
            after = None
    y1, y2 = ops.LinearActQPair.apply(x, l1.conv1d.bias, l2.conv1d.bias, q1.qmin, q1.qmax, q2.qmin, q2.qmax, L1, L2, q1, q2, xq, pair,
                                      sole_ew_consumers, after)
    aqs[0].after_forward(q1)
    aqs[1].after_forward(q2)
    return ops.tag_codes(y1, q1), ops.tag_codes(y2, q2)


class Conv1dQ(LayerQ):
    def __init__(self, conv1d, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(conv1d, nn.Conv1d, "Conv1d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv1d.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.conv1d = conv1d

    def forward(self, x):
        return run_conv1d(self.conv1d, x, self._wq(self.conv1d.weight), None, self.activation_fake_quantize)


class Conv1dNlQ(LayerQ):
    def __init__(self, conv1d, nl, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(conv1d, nn.Conv1d, "Conv1d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv1d.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.conv1d = conv1d
        self.nl = nl

    def forward(self, x, pad_to=None):
        return run_conv1d(self.conv1d, x, self._wq(self.conv1d.weight), self.nl, self.activation_fake_quantize, pad_to)


class GroupNormQ(LayerQ):
    def __init__(self, groupnorm, gradient_based=True, act_quant=True, act_n_bits=8):
        _expect(groupnorm, nn.GroupNorm, "GroupNorm")
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        self.groupnorm = groupnorm

    def forward(self, x):
        return run_groupnorm(self.groupnorm, x, self.activation_fake_quantize)

    def forward_rows(self, x, geom):
        """the same layer on a dual-path row layout [L', B', C]; geom = (RB, X, B): sample of row r = (r % RB) // X"""
        return run_groupnorm_rows(self.groupnorm, x, self.activation_fake_quantize, geom)


def run_groupnorm(gn, x, aq):
    if gn.num_groups != 1 or not gn.affine:
        raise NotImplementedError("only GroupNorm(num_groups=1, affine=True) (gLN) has a HIP kernel")
    if x.dim() == 4:       # [B, C, H, W] (Sepformer's dual-path blocks in the reference's layout): statistics over C*H*W
        shp = x.shape
        return ops.reshape_tagged(run_groupnorm(gn, ops.reshape_tagged(x, shp[0], shp[1], shp[2] * shp[3]), aq), *shp)
    q = aq.qctx() if aq is not None else ops.BYPASS
    xq = ops.codes_of(x)
    if not (xq is not None and q.qmode == ops.Q_QUANT):
        x, xq = ops.real(x), None
    y = ops.GroupNormActQ.apply(x, gn.weight, gn.bias, q.qmin, q.qmax, gn.eps, q, gn.weight, gn.bias, xq)
    if aq is not None:
        aq.after_forward(q)
    return ops.tag_codes(y, q)


def run_groupnorm_rows(gn, x, aq, geom):
    if gn.num_groups != 1 or not gn.affine:
        raise NotImplementedError("only GroupNorm(num_groups=1, affine=True) (gLN) has a HIP kernel")
    return fq_node(aq, ops_dp.GroupNormRows.apply(ops.real(x), gn.weight, gn.bias, gn.eps, tuple(geom)))


def apply_map(nl, x):
    """the non-linearities that run as their own kernel in front of a quantizer"""
    if isinstance(nl, nn.Tanh):
        return ops_dp.Unary.apply(x, K.UNARY_TANH)
    if isinstance(nl, nn.Sigmoid):
        return ops_dp.Unary.apply(x, K.UNARY_SIGMOID)
    if isinstance(nl, nn.GELU):
        if getattr(nl, "approximate", "none") != "none":
            raise NotImplementedError("GELU: only the erf form has a HIP kernel")
        return ops_dp.Gelu.apply(x)
    if isinstance(nl, nn.GLU):
        if nl.dim != 1 or x.dim() < 3:
            raise NotImplementedError("GLU: only dim=1 of a channel-first tensor has a HIP kernel")
        shp = x.shape
        y = ops_dp.Glu.apply(ops.flat_cm(x))            # (no copy of a pitch-Wp / row-padded 4-D input)
        return y.reshape(shp[0], shp[1] // 2, *shp[2:])
    raise NotImplementedError(type(nl).__name__)


def run_nl(nl, x, aq):
    if isinstance(nl, (nn.Tanh, nn.Sigmoid, nn.GELU, nn.GLU)):
        return fq_node(aq, apply_map(nl, ops.real(x)))
    act, slope = _act_of(nl)
    q = aq.qctx() if aq is not None else ops.BYPASS
    y = ops.ew_layer(x, None, 0.0, act, slope, q)
    if aq is not None:
        aq.after_forward(q)
    return ops.tag_codes(y, q)


class NlQ(LayerQ):
    def __init__(self, nl, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        self.nl = nl

    def forward(self, x):
        return run_nl(self.nl, x, self.activation_fake_quantize)


# ---------------------------------------------------------------------------------------------
# 8-bit I/O blocks
# ---------------------------------------------------------------------------------------------
class Conv1dEncoderQ(LayerQ):
    """encoder conv over the n_splitter x 8-bit input channels.  At construction the extra
    splitter channels get Gaussian weights drawn around the float kernel's statistics
    (mean + randn * std**n), as the reference does (qat_layers.py:1019-1024)."""

    def __init__(self, encoder, n_splitter=1, gradient_based=True, weight_quant=True, act_quant=True, in_quant=False,
                 inout_nl_quant=False, act_n_bits=8, weight_n_bits=8, in_act_n_bits=8):
        conv = encoder[0]
        _expect(conv, nn.Conv1d, "Conv1d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.in_quantizer = (get_activation_quantizer(self.gradient_based, nl=inout_nl_quant, n_bits=in_act_n_bits)
                             if in_quant else nn.Identity())
        self.nl = nn.Identity() if len(encoder) == 1 else encoder[1]
        if n_splitter >= 2:
            w = conv.weight.detach()
            cin = conv.in_channels
            wide = nn.Conv1d(n_splitter * cin, conv.out_channels, conv.kernel_size, stride=conv.stride,
                             padding=conv.padding, bias=conv.bias is not None)
            new_w = w.repeat(1, n_splitter, 1)
            for ch in range(1, n_splitter):
                for c in range(cin):
                    base = w[:, c, :]
                    new_w[:, ch * cin + c, :] = torch.mean(base) + torch.randn_like(base) * (torch.std(base) ** ch)
            with torch.no_grad():
                wide.weight.copy_(new_w)
                if conv.bias is not None:
                    wide.bias.copy_(conv.bias)
            conv = wide
        self.conv1d = conv

    def forward(self, x):
        x = self.in_quantizer(x)
        return run_conv1d(self.conv1d, x, self._wq(self.conv1d.weight), self.nl, self.activation_fake_quantize)


class ResidualErrorBlock(LayerQ):
    """second ("LSB") output channel: re-encode the quantized output, quantize the encoding error,
    decode it with the decoder's own quantized kernel."""

    def __init__(self, decoder, gradient_based, weight_quant, act_quant, act_nl_quantizer=False, act_n_bits=8,
                 weight_n_bits=8, train_res_dec=False):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_nl_quantizer=act_nl_quantizer,
                         act_n_bits=act_n_bits)
        if type(decoder) not in (nn.ConvTranspose1d, nn.ConvTranspose2d, nn.Linear):
            raise AssertionError("Not supported residual block for type {}".format(type(decoder)))
        if train_res_dec and type(decoder) is nn.Linear:
            raise NotImplementedError("train_res_dec=True: only the transposed-convolution decoders have kernels")
        self.decoder_type = type(decoder)
        self.train_res_dec = train_res_dec
        self.decoder_bias = decoder.bias        # the reference registers the decoder's bias a second time (state_dict key `decoder_bias`)
        if self.decoder_type is nn.Linear:      # qat_layers.py:1111-1114
            self.residual_encoder = nn.Linear(decoder.out_features, decoder.in_features, bias=decoder.bias is not None)
            self.decoder_stride = None
        else:
            conv_t, convtr_t = (nn.Conv1d, nn.ConvTranspose1d) if self.decoder_type is nn.ConvTranspose1d else (nn.Conv2d, nn.ConvTranspose2d)
            self.residual_encoder = conv_t(decoder.out_channels, decoder.in_channels, decoder.kernel_size,
                                           stride=decoder.stride, bias=decoder.bias is not None)
            self.decoder_stride = decoder.stride
            object.__setattr__(self, "_decoder_geom", decoder)     # geometry donor of the shared-kernel decode (NOT a submodule: no state_dict keys)
            if train_res_dec:       # the LSB channel gets its own trainable decoder (qat_layers.py:1137-1146, 1160-1168)
                self.residual_decoder = convtr_t(decoder.in_channels, decoder.out_channels, decoder.kernel_size,
                                                 stride=decoder.stride, bias=decoder.bias is not None)
                self.weight_fake_quantize_dec = (get_weight_quantizer(gradient_based, self.residual_decoder.weight.shape, ch_out_idx=1,
                                                                      n_bits=weight_n_bits) if weight_quant else nn.Identity())
        self.weight_fake_quantize = (get_weight_quantizer(gradient_based, self.residual_encoder.weight.shape, n_bits=weight_n_bits)
                                     if weight_quant else nn.Identity())

    def forward(self, Y, y_q, w_decoder, decoder_conv=None, out_quantizer=None, window=None):
        """reference signature is (Y, y_q, w_decoder); the two optional arguments let the owning
        decoder fuse its `activation_fake_quantize_residual` into the transposed-conv node; window (convtr_frames): the slice of the
        decoded error the caller keeps (the general form only; the caller checks the output quantizer's phase)"""
        enc = self.residual_encoder
        if self.decoder_type is nn.Linear:      # qat_layers.py:1178-1187 (rows [..., E]); LinearDecoderQ inlines this sequence
            Y_q = ops_dp.RowLinear.apply(ops.real(y_q), self._wq(enc.weight), enc.bias)
            aq = self.activation_fake_quantize
            q = aq.qctx()
            Y1 = ops.AddActQ.apply(ops.real(Y), Y_q, q.qmin, q.qmax, -1.0, q)
            aq.after_forward(q)
            return ops_dp.RowLinear.apply(Y1, w_decoder, None)
        if self.decoder_type is nn.ConvTranspose2d or not _is_mono_decoder(self._decoder_geom):
            return self._forward_general(Y, y_q, w_decoder, out_quantizer, window)
        Y_q = run_conv1d(enc, y_q, self._wq(enc.weight), None, None)
        aq = self.activation_fake_quantize
        q = aq.qctx()
        Y1 = ops.tag_codes(ops.ew_layer(Y, Y_q, -1.0, ops.ACT_NONE, None, q), q)
        aq.after_forward(q)
        if decoder_conv is None and not self.train_res_dec:
            Y1 = ops.real(Y1)
        if self.train_res_dec:
            decoder_conv_, w_decoder = self.residual_decoder, self.weight_fake_quantize_dec(self.residual_decoder.weight)
            if decoder_conv is None:
                return run_convtr1d(decoder_conv_, Y1, w_decoder, None)
            return run_convtr1d(decoder_conv_, Y1, w_decoder, out_quantizer)
        if decoder_conv is None:
            L = ops._Lin("convtr", stride=self.decoder_stride[0])
            return ops.LinearActQ.apply(Y1, w_decoder, None, None, None, None, L, ops.ACT_NONE, ops.BYPASS)
        return run_convtr1d(decoder_conv, Y1, w_decoder, out_quantizer)


    def _forward_general(self, Y, y_q, w_decoder, out_quantizer, window=None):
        """stereo / biased / 2-D decoders of HTDemucs (qat_layers.py:1189-1216): the same sequence over the frame kernels.
        Quirks kept: the residual encoder ignores the decoder's padding; the 1-D decode drops the bias, the 2-D decode uses
        `residual_decoder.bias` (so the 2-D form needs train_res_dec=True, as the reference does)"""
        enc, dec = self.residual_encoder, self._decoder_geom
        Y_q = conv_frames(enc, y_q, self._wq(enc.weight))
        Y = ops.real(Y)
        shp = Y.shape
        aq = self.activation_fake_quantize
        q = aq.qctx()
        Y1 = ops.tag_codes(ops.ew_layer(Y.reshape(shp[0], shp[1], -1), Y_q.reshape(shp[0], shp[1], -1), -1.0, ops.ACT_NONE, None, q), q)
        aq.after_forward(q)
        Y1 = ops.real(Y1).reshape(shp)
        if self.decoder_type is nn.ConvTranspose2d:
            if not self.train_res_dec:
                raise AttributeError("'ResidualErrorBlock' object has no attribute 'residual_decoder'")     # qat_layers.py:1209
            bias = self.residual_decoder.bias
        else:
            bias = None
        w = self.weight_fake_quantize_dec(self.residual_decoder.weight) if self.train_res_dec else w_decoder
        return fq_node(out_quantizer, convtr_frames(dec, Y1, w, bias=bias, window=window))


def _observing(aq):
    return bool(getattr(aq, "observer_mode", False)) and aq.n_iter < aq.max_observations


def _decoder_window(layer, window, convtr):
    """the window a two-channel decoder layer hands to its residual decode, or None: two outputs (a third channel would re-encode the
    cropped second one), both output quantizers past their observer phase, the general (frame) form of the transposed convolution"""
    if window is None or layer.n_combiner != 2 or _observing(layer.activation_fake_quantize) \
            or _observing(layer.activation_fake_quantize_residual):
        return None
    if isinstance(convtr, nn.ConvTranspose1d) and _is_mono_decoder(convtr):
        return None
    return window


def _is_mono_decoder(convtr):
    """the ConvTasNet / Sepformer waveform decoder served by the dedicated overlap-add kernel (fqss_ola_convtr_fwd)"""
    return isinstance(convtr, nn.ConvTranspose1d) and convtr.out_channels == 1 and convtr.padding[0] == 0 and convtr.output_padding[0] == 0 \
        and convtr.dilation[0] == 1 and convtr.groups == 1 and convtr.bias is None


def run_convtr1d(convtr, x, weight, aq):
    if not _is_mono_decoder(convtr):
        return fq_node(aq, convtr_frames(convtr, x, weight))
    L = ops._Lin("convtr", stride=convtr.stride[0], w_param=convtr.weight)
    q = aq.qctx() if aq is not None else ops.QCtx()
    q.keep_out = True          # waveform-side outputs are the model's outputs: always real fp32
    xq = ops.codes_of(x) if (FUSE_DECQ and x.dim() == 3 and K.ola_convtr_ok(weight, convtr.stride[0])) else None
    if xq is not None and (K.rowmat(xq.idx) is None or K.rowmat(xq.idx)[2] % 16 != 0):
        xq = None
    if xq is None:
        x = ops.real(x)        # no coded-input kernel for this case: decode a carrier first
    y = ops.LinearActQ.apply(x, weight, None, None, q.qmin, q.qmax, L, ops.ACT_NONE, q, xq)
    if aq is not None:
        aq.after_forward(q)
    return ops.tag_codes(y, q)


class ConvTr1dDecoderQ(LayerQ):
    def __init__(self, decoder, n_combiner=1, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True,
                 act_n_bits=8, inout_nl_quant=False, out_quant=True, out_act_n_bits=8, train_res_dec=False):
        conv = decoder[0]
        _expect(conv, nn.ConvTranspose1d, "ConvTranspose1d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=out_quant,
                         act_nl_quantizer=inout_nl_quant, weight_shape=conv.weight.shape, ch_out_idx=1,
                         act_n_bits=out_act_n_bits, weight_n_bits=weight_n_bits)
        self.n_combiner = n_combiner
        self.convTr1d = conv
        if self.n_combiner >= 2:
            self.residual_error_block = ResidualErrorBlock(conv, gradient_based, weight_quant=weight_quant,
                                                           act_quant=act_quant, weight_n_bits=weight_n_bits,
                                                           act_n_bits=act_n_bits, train_res_dec=bool(train_res_dec))
            self.activation_fake_quantize_residual = (get_activation_quantizer(gradient_based, n_bits=out_act_n_bits)
                                                      if out_quant else _BypassQuantizer())

    def forward(self, x, window=None):
        """window (convtr_frames): HTDemucs keeps a slice of the last time-branch layer's outputs (hdemucsq.py:340-345).  With two output
        channels past their observer phase the residual decode writes just that slice and the first channel -- which the residual block
        re-encodes whole -- is cropped by the stacking copy; else the window is ignored (the caller crops)."""
        w_decoder = self._wq(self.convTr1d.weight)
        if self.n_combiner == 1:
            return run_convtr1d(self.convTr1d, x, w_decoder, self.activation_fake_quantize)
        window = _decoder_window(self, window, self.convTr1d)
        x_dec, x_res = ops.fork2(x)
        y = run_convtr1d(self.convTr1d, x_dec, w_decoder, self.activation_fake_quantize)
        outs = [y]
        for _ in range(1, self.n_combiner):
            y = self.residual_error_block(x_res, y, w_decoder, self.convTr1d, self.activation_fake_quantize_residual, window=window)
            outs.append(y)
        if window is not None:
            outs[0] = ops.real(outs[0]).narrow(*window[:2], window[2])
        return torch.stack(outs)


# ---------------------------------------------------------------------------------------------
# dual-path layers (DPTNet): row-major tensors [..., C]
# ---------------------------------------------------------------------------------------------
_FLAT_COLS = {}


def _flat2d(t):
    """a dense tensor as [rows, cols] with long rows (the element-wise kernels stream rows; 64-wide rows would idle 3/4 of
    every workgroup): cols = the largest divisor of numel in [1024, 16384] that is a multiple of 16 (rows of activation
    buffers are padded to 16 floats: a multiple of 16 keeps the kernel's output dense, so the view back is free)"""
    n = t.numel()
    if not t.is_contiguous() or n < 4096:
        return None
    c = _FLAT_COLS.get(n)
    if c is None:
        c = next((c for c in range(16384, 1008, -16) if n % c == 0), 0)
        _FLAT_COLS[n] = c
    return t.view(n // c, c) if c else None


def fq_node(aq, x, nl=None, codes=False, q=None, post_relu=False):
    """fq_act(nl(x)) as its own autograd node (fqss_actq_fwd/bwd); float modules (aq None) only apply nl.
    codes=True: also emit the u8 codes and tag the result with them (the consumer is a row linear that can run on codes);
    q: the quantizer's context when the caller already drew it (aq.qctx() advances the observer's call count);
    post_relu: relu(fq_act(x)) -- a ReLU BEHIND the quantizer (the F.relu between LSTMQ and LinearQ, dptnetq.py:92) in the same pass
    (act = ACT_POST_RELU); nl must be None"""
    if post_relu:
        assert nl is None and not codes
        if q is None:
            q = aq.qctx() if aq is not None else ops.BYPASS
        if q.qmode == ops.Q_BYPASS or not (FUSE_POSTRELU and ops.CODED):
            y = x if q.qmode == ops.Q_BYPASS else fq_node(aq, x, q=q)
            return fq_node(None, y, nn.ReLU())
        x = ops.real(x)
        flat = _flat2d(x)
        q.no_codes = True
        y = ops.NlActQ.apply(x if flat is None else flat, None, q.qmin, q.qmax, K.ACT_POST_RELU, q, None)
        if aq is not None:
            aq.after_forward(q)
        q.idx = None
        return y if flat is None else y.reshape(x.shape)
    if FUSE_GLUQ and ops.CODED and isinstance(nl, nn.GLU) and nl.dim == 1 and aq is not None and not codes and torch.is_tensor(x) and x.dim() >= 3:
        # GLU rides in the quantizer's own pass each way (fqss_gluq_fwd / _bwd) in the quantizing and the observer phase
        if q is None:
            q = aq.qctx()
        xr = ops.real(x)
        x3 = ops.flat_cm(xr)
        if q.qmode != ops.Q_BYPASS and K.gluq_rows_ok(x3):
            q.no_codes = True
            y = ops_dp.GluActQ.apply(x3, q.qmin, q.qmax, q)
            aq.after_forward(q)
            q.idx = None
            return y.reshape(xr.shape[0], xr.shape[1] // 2, *xr.shape[2:])
    gelu = FUSE_GELUQ and ops.CODED and isinstance(nl, nn.GELU) and getattr(nl, "approximate", "none") == "none"
    if gelu:
        nl = None             # GELU rides in the quantizer's own pass each way (act = ACT_GELU: fqss_actq_fwd / _bwd)
    elif isinstance(nl, (nn.Tanh, nn.Sigmoid, nn.GELU, nn.GLU)):
        x, nl = apply_map(nl, ops.real(x)), None
    act, slope = _act_of(nl)
    if gelu:
        act = K.ACT_GELU
    if q is None:
        q = aq.qctx() if aq is not None else ops.BYPASS
    if q.qmode == ops.Q_BYPASS and act == ops.ACT_NONE:
        return x
    x = ops.real(x)
    flat = _flat2d(x)
    want = codes and ops_dp.QROW and ops.CODED and q.qmode == ops.Q_QUANT
    q.no_codes = not want     # no coded consumer downstream: skip the 1 B/element side output
    # the fp32 values are written next to the codes even on the codes-only dataflow: the consuming row linear multiplies the codes in
    # its forward but its weight gradient (and any other consumer) reads the fp32 tensor -- left unwritten it was uninitialised
    # memory under KDTrainStep (found with FQSS_DEBUG_CARRIER=1: NaN weight gradients of every coded row linear)
    q.keep_out = q.keep_out or want
    y = ops.NlActQ.apply(x if flat is None else flat, slope, q.qmin, q.qmax, act, q, slope)
    if aq is not None:
        aq.after_forward(q)
    idx, q.idx = q.idx, None
    y = y if flat is None else y.reshape(x.shape)
    if want and idx is not None and idx.is_contiguous() and y.is_contiguous():
        y._fqss_rowq = ops.ActCodes(idx.view(x.shape), q.qmin.detach(), q.qmax.detach())
    return y


FUSE_POSTRELU = __import__("os").environ.get("FQSS_FUSE_POSTRELU", "1") != "0"   # 0: the ReLU behind LSTMQ's quantizer as its own pass
FUSE_GLUQ = __import__("os").environ.get("FQSS_FUSE_GLUQ", "1") != "0"     # 0: GLU as its own pass in front of the quantizer (A/B, tests)
FUSE_GELUQ = __import__("os").environ.get("FQSS_FUSE_GELUQ", "1") != "0"   # 0: GELU as its own pass in front of the quantizer (A/B, tests)
FUSE_ROWQ = __import__("os").environ.get("FQSS_FUSE_ROWQ", "1") != "0"    # 0: row linear, quantizer and bias sums as separate nodes (A/B, tests)


def run_linear(lin, x, weight, nl, aq, post_relu=False):
    return linear_fq(x, weight, lin.bias, nl, aq, post_relu)


FUSE_NLQ2 = os.environ.get("FQSS_FUSE_NLQ2", "1") != "0"    # LinearQ -> NlQ(ReLU): both quantizers in the int8 GEMM's epilogue


def _quantizing(aq):
    """a GradientActivationFakeQuantize past its observer phase with partial-sum slots (what qctx() WILL answer, without advancing it)"""
    return (getattr(aq, "observer_mode", None) is not None and not (aq.observer_mode and aq.n_iter < aq.max_observations)
            and getattr(aq, "_gacc", None) is not None)


def linear_then_relu_q(lq, nlq, x):
    """nlq(lq(x)) for a LinearQ followed by NlQ(ReLU) (the Sepformer feed-forward block): one launch forward, one pass backward when both
    quantizers quantize and the GEMM runs on codes (ops_dp.RowLinearNlQ2); any other state takes the two modules as they are"""
    aq1, aq2 = lq.activation_fake_quantize, nlq.activation_fake_quantize
    lin = lq.linear
    if (FUSE_NLQ2 and FUSE_ROWQ and ops.CODED and isinstance(nlq.nl, nn.ReLU) and lin.bias is not None and lin.out_features % 4 == 0
            and K.colbias_ok(lin.out_features) and _quantizing(aq1) and _quantizing(aq2)):
        weight = lq._wq(lin.weight)
        qops = ops_dp.qrow_operands(x, weight) if weight.dim() == 2 else None
        if qops is not None and ops_dp.FUSE_QROWQ:
            q1, q2 = aq1.qctx(), aq2.qctx()
            if q1.qmode != ops.Q_QUANT or q2.qmode != ops.Q_QUANT or q1.gacc is None or q2.gacc is None:
                raise RuntimeError("linear_then_relu_q: quantizer state changed between the check and qctx()")
            y = ops_dp.RowLinearNlQ2.apply(x, weight, lin.bias, q1.qmin, q1.qmax, q2.qmin, q2.qmax, q1, q2, qops)
            aq1.after_forward(q1)
            aq2.after_forward(q2)
            idx, q2.idx = q2.idx, None
            y._fqss_rowq = ops.ActCodes(idx.view(y.shape), q2.qmin.detach(), q2.qmax.detach())
            return y
        return nlq(run_linear(lin, x, weight, None, aq1))
    return nlq(lq(x))


def linear_fq(x, weight, bias, nl, aq, post_relu=False):
    """fq(nl(x @ weight^T + bias)) on row-major tensors: LinearQ / LinearNlQ and the attention output projection;
    post_relu: relu(fq(...)) -- the caller's F.relu / nn.ReLU on the layer's output, in the quantizer's pass (act = ACT_POST_RELU)"""
    q = None
    post_relu = post_relu and nl is None and FUSE_POSTRELU and ops.CODED
    if aq is not None and FUSE_ROWQ and bias is not None and weight.dim() == 2 and (nl is None or isinstance(nl, (nn.ReLU, nn.PReLU))):
        q = aq.qctx()
        if q.qmode == ops.Q_QUANT and q.gacc is not None and K.colbias_ok(weight.shape[0]):
            # quantizing phase: one node for linear + quantizer, the bias gradient rides in the quantizer's backward pass
            act, slope = _act_of(nl)
            if post_relu:
                act = K.ACT_POST_RELU
            qops = ops_dp.qrow_operands(x, weight)
            q.no_codes, q.carrier = True, False
            n = x.numel() // x.shape[-1] * weight.shape[0]
            cols = _FLAT_COLS.get(n)
            if cols is None:
                cols = _FLAT_COLS[n] = next((c for c in range(16384, 1008, -16) if n % c == 0), 0) if n >= 4096 else 0
            y = ops_dp.RowLinearActQ.apply(x if qops is not None else ops.real(x), weight, bias, slope, q.qmin, q.qmax, act, q, qops, slope,
                                           (n // cols, cols) if cols else None)
            aq.after_forward(q)
            q.idx = None
            return y
    if post_relu:
        return fq_node(aq, ops_dp.row_linear(x, weight, bias), q=q, post_relu=True)
    return fq_node(aq, ops_dp.row_linear(x, weight, bias), nl, q=q)


def run_layernorm(ln, x, aq):
    if len(ln.normalized_shape) != 1 or not ln.elementwise_affine:
        raise NotImplementedError("only LayerNorm over the last dim with affine parameters has a HIP kernel")
    q = None
    if aq is not None and FUSE_LNQ:
        q = aq.qctx()
        if q.qmode == ops.Q_QUANT and q.gacc is not None:
            # quantizing phase: LayerNorm + output quantizer as one kernel each way (ops_dp.LayerNormRowsQ)
            want = ops_dp.QROW and ops.CODED
            y = ops_dp.LayerNormRowsQ.apply(ops.real(x), ln.weight, ln.bias, ln.eps, q.qmin, q.qmax, q, want)
            aq.after_forward(q)
            idx, q.idx = q.idx, None
            if idx is not None:
                y._fqss_rowq = ops.ActCodes(idx.view(y.shape), q.qmin.detach(), q.qmax.detach())
            return y
    return fq_node(aq, ops_dp.LayerNormRows.apply(ops.real(x), ln.weight, ln.bias, ln.eps), codes=True, q=q)


def add_layernorm(norm, a, b):
    """(norm(a + b), a + b) for the float residual add that feeds a pre-norm sub-layer (`x = x + sublayer(...)` followed by `norm(x)`):
    fused into ONE kernel each way (ops_dp.AddLayerNormRows) for an nn.LayerNorm and for a LayerNormQ in the quantizing phase; any
    other state (observer phase, per-module quantizers without a gacc arena, FQSS_FUSE_ADDLN=0) takes the un-fused pair"""
    ln = norm.layernorm if isinstance(norm, LayerNormQ) else norm
    a, b = ops.real(a), ops.real(b)
    if FUSE_ADDLN and isinstance(ln, nn.LayerNorm) and len(ln.normalized_shape) == 1 and ln.elementwise_affine:
        if not isinstance(norm, LayerNormQ):
            y, s = ops_dp.AddLayerNormRows.apply(a, b, ln.weight, ln.bias, ln.eps, None, None, None, False)
            return y, s
        aq = norm.activation_fake_quantize
        if FUSE_LNQ and getattr(aq, "observer_mode", None) is not None and not (aq.observer_mode and aq.n_iter < aq.max_observations) \
                and getattr(aq, "_gacc", None) is not None:
            q = aq.qctx()
            if q.qmode == ops.Q_QUANT and q.gacc is not None:
                want = ops_dp.QROW and ops.CODED
                y, s = ops_dp.AddLayerNormRows.apply(a, b, ln.weight, ln.bias, ln.eps, q.qmin, q.qmax, q, want)
                aq.after_forward(q)
                idx, q.idx = q.idx, None
                if idx is not None:
                    y._fqss_rowq = ops.ActCodes(idx.view(y.shape), q.qmin.detach(), q.qmax.detach())
                return y, s
            raise RuntimeError("add_layernorm: quantizer state changed between the check and qctx()")
    s = ops.AddActQ.apply(a, b, None, None, 1.0, ops.BYPASS)
    s_n, s_res = ops.fork2(s)
    return norm(s_n), s_res


FUSE_LN_LAYOUT = os.environ.get("FQSS_FUSE_LN_LAYOUT", "1") != "0"    # the dual-path layout change inside the AddQ + LayerNormQ kernels


def addq_layernorm(add, norm, a, b, then=None):
    """norm(add(a, b)) for the post-norm layers of DPTNet (dptnetq.py:84-97: `src = norm(add_norm(src, src2))`): with an AddQ and a
    LayerNormQ both in their quantizing phase (deferred range tables) the quantized add rides in the LayerNorm kernels each way
    (ops_dp.AddLayerNormRows with a sum quantizer: fqss_addq_layernorm_fwd/bwd) -- no axpby, no quantizer pass, no fork sum; any other
    state runs the two modules"""
    if FUSE_ADDLN and FUSE_LNQ and isinstance(add, AddQ) and isinstance(norm, LayerNormQ) and not _is_row_bcast(a, b):
        ln, aq, aqs = norm.layernorm, norm.activation_fake_quantize, add.activation_fake_quantize
        ok = isinstance(ln, nn.LayerNorm) and len(ln.normalized_shape) == 1 and ln.elementwise_affine and a.shape == b.shape
        for t in (aq, aqs):
            ok = ok and getattr(t, "observer_mode", None) is not None and not (t.observer_mode and t.n_iter < t.max_observations) \
                and getattr(t, "_gacc", None) is not None
        if ok:
            q, qs = aq.qctx(), aqs.qctx()
            if q.qmode == ops.Q_QUANT and qs.qmode == ops.Q_QUANT and q.gacc is not None and qs.gacc is not None:
                want = ops_dp.QROW and ops.CODED
                # then = (to, B): the layout change that follows this layer (DPT.forward) is done by the kernel that writes y anyway
                rmap = ops_dp.layout_map(tuple(a.shape), then[0], then[1]) if (then is not None and FUSE_LN_LAYOUT and a.dim() == 3) else None
                y, _ = ops_dp.AddLayerNormRows.apply(ops.real(a), ops.real(b), ln.weight, ln.bias, ln.eps, q.qmin, q.qmax, q, want, qs, qs.qmin, qs.qmax,
                                                     rmap)
                aqs.after_forward(qs)
                aq.after_forward(q)
                idx, q.idx = q.idx, None
                if idx is not None:
                    y._fqss_rowq = ops.ActCodes(idx.view(y.shape), q.qmin.detach(), q.qmax.detach())
                return y if (then is None or rmap is not None) else ops_dp.change_layout(y, then[0], then[1])
            raise RuntimeError("addq_layernorm: quantizer state changed between the check and qctx()")
    y = norm(add(a, b))
    y = y[0] if isinstance(y, (list, tuple)) else y
    return y if then is None else ops_dp.change_layout(y, then[0], then[1])


FUSE_ADDLN = __import__("os").environ.get("FQSS_FUSE_ADDLN", "1") != "0"   # residual add + LayerNorm(Q) as one kernel each way
FUSE_LNQ = __import__("os").environ.get("FQSS_FUSE_LNQ", "1") != "0"     # 0: LayerNorm and its quantizer as separate launches (A/B, tests)


def _lstm_check(lstm):
    if lstm.num_layers != 1 or not lstm.bidirectional or lstm.batch_first or lstm.proj_size != 0 or not lstm.bias or lstm.dropout != 0:
        raise NotImplementedError("only the single-layer bidirectional sequence-first LSTM of the dual-path models has HIP kernels")


def run_lstm(lstm, x, weights, aq, post_relu=False):
    """weights: dict name -> (fake-quantized) weight for the four weight matrices; post_relu: relu(fq(lstm(x)))"""
    _lstm_check(lstm)
    y = ops_dp.LstmBi.apply(ops.real(x), weights["weight_ih_l0"], weights["weight_hh_l0"], lstm.bias_ih_l0, lstm.bias_hh_l0,
                            weights["weight_ih_l0_reverse"], weights["weight_hh_l0_reverse"], lstm.bias_ih_l0_reverse,
                            lstm.bias_hh_l0_reverse, getattr(x, "_fqss_rowq", None),
                            getattr(weights["weight_ih_l0"], "_fqss_wcodes", None), getattr(weights["weight_ih_l0_reverse"], "_fqss_wcodes", None))
    return fq_node(aq, y, post_relu=True) if post_relu else fq_node(aq, y)


def _mha_check(mha, query, key, value):
    if not (query is key and key is value):
        raise NotImplementedError("MultiheadAttentionQ: only self-attention (query is key is value) has HIP kernels")
    if mha.batch_first or mha.in_proj_weight is None or mha.bias_k is not None or mha.add_zero_attn or mha.dropout != 0:
        raise NotImplementedError("MultiheadAttention variant without a HIP kernel")


def run_mha(mha, x, w_in, w_out, aqs, aq_head, aq_out):
    """x [L, B, E] -> [L, B, E]; aqs = (q, k, v, div, attn, softmax) quantizers or None for the float module"""
    X = ops_dp.row_linear(x, w_in, mha.in_proj_bias)
    if aqs is None:
        heads = ops_dp.MhaCore.apply(X, mha.num_heads, None)
    else:
        ranges = [r for a in aqs[:4] for r in (a.min_range, a.max_range)]
        heads = ops_dp.MhaCore.apply(X, mha.num_heads, aqs, *ranges)
    heads = fq_node(aq_head, heads, codes=True)
    return linear_fq(heads, w_out, mha.out_proj.bias, None, aq_out)


def run_mha_x(mha, query, key, value, w_in, w_out, aqs, aq_head, aq_out):
    """general form of run_mha: key is value but may differ from query (cross attention), batch-first or sequence-first rows"""
    if value is not key:
        raise NotImplementedError("MultiheadAttentionQ: value must be the key tensor (self- or cross-attention)")
    if mha.in_proj_weight is None or mha.bias_k is not None or mha.add_zero_attn or mha.dropout != 0:
        raise NotImplementedError("MultiheadAttention variant without a HIP kernel")
    Xq = ops_dp.row_linear(query, w_in, mha.in_proj_bias)
    Xkv = None if key is query else ops_dp.row_linear(key, w_in, mha.in_proj_bias)
    if aqs is None:
        heads = ops_dp.MhaCoreX.apply(Xq, Xkv, mha.num_heads, bool(mha.batch_first), None)
    else:
        ranges = [r for a in aqs[:4] for r in (a.min_range, a.max_range)]
        heads = ops_dp.MhaCoreX.apply(Xq, Xkv, mha.num_heads, bool(mha.batch_first), aqs, *ranges)
    heads = fq_node(aq_head, heads, codes=True)
    return linear_fq(heads, w_out, mha.out_proj.bias, None, aq_out)


class LayerNormQ(LayerQ):
    def __init__(self, layernorm, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(layernorm, nn.LayerNorm, "LayerNorm")
        self.layernorm = layernorm

    def forward(self, x):
        return run_layernorm(self.layernorm, x, self.activation_fake_quantize)


class LinearQ(LayerQ):
    def __init__(self, linear, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(linear, nn.Linear, "Linear")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=linear.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.linear = linear

    def forward(self, x, post_relu=False):
        """post_relu (not in the reference's signature): the caller's nn.ReLU on the output, folded into the output quantizer's pass"""
        return run_linear(self.linear, x, self._wq(self.linear.weight), None, self.activation_fake_quantize, post_relu)


class Conv2dQ(LayerQ):
    """1x1 Conv2d (DPT.output[1], dptnetq.py:187).  forward() takes the reference's [B, C, H, W]; the dual-path model calls
    forward_rows() with its row-major tensors [..., C] instead (same parameters, same quantizers, no layout change)."""

    def __init__(self, conv2d, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(conv2d, nn.Conv2d, "Conv2d")
        self._is_1x1 = conv2d.kernel_size == (1, 1) and conv2d.stride == (1, 1) and conv2d.padding == (0, 0) and conv2d.groups == 1
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv2d.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.conv2d = conv2d

    def forward_rows(self, x):
        c = self.conv2d
        if not self._is_1x1:
            raise NotImplementedError("Conv2dQ.forward_rows: only the 1x1 convolution runs on row-major tensors")
        w = ops.weight_view(self._wq(c.weight), c.out_channels, c.in_channels)
        return fq_node(self.activation_fake_quantize, ops_dp.RowLinear.apply(ops.real(x), w, c.bias))

    def forward(self, x):
        if not self._is_1x1:
            return fq_node(self.activation_fake_quantize, conv_frames(self.conv2d, x, self._wq(self.conv2d.weight)))
        return run_conv2d_1x1(self.conv2d, x, self._wq(self.conv2d.weight), self.activation_fake_quantize)


def run_conv2d_1x1(c, x, weight, aq):
    B, C, H, W = x.shape
    L = ops._Lin("pw", w_param=c.weight, b_param=c.bias)
    ops_dp.touch(weight)
    z = ops.LinearActQ.apply(ops.real(x).reshape(B, C, H * W), ops.weight_view(weight, c.out_channels, c.in_channels, 1), c.bias, None, None, None,
                             L, ops.ACT_NONE, ops.BYPASS)
    return fq_node(aq, z).reshape(B, c.out_channels, H, W)


class LSTMQ(LayerQ):
    def __init__(self, lstm, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(lstm, nn.LSTM, "LSTM")
        _lstm_check(lstm)
        self.lstm = lstm
        self.num_directions = 2 if lstm.bidirectional else 1
        self.real_hidden_size = lstm.proj_size if lstm.proj_size > 0 else lstm.hidden_size
        self.weight_quantizers_dict = nn.ModuleDict()
        for name, w in zip(lstm._flat_weights_names, lstm._flat_weights):
            if name.startswith("weight"):
                self.weight_quantizers_dict[name] = (get_weight_quantizer(gradient_based, w.shape, n_bits=weight_n_bits)
                                                     if weight_quant else nn.Identity())

    def forward(self, x, post_relu=False):
        """post_relu (not in the reference's signature): the caller's F.relu on the output, folded into the output quantizer's pass"""
        weights = {n: q(getattr(self.lstm, n)) for n, q in self.weight_quantizers_dict.items()}
        return [run_lstm(self.lstm, x, weights, self.activation_fake_quantize, post_relu)]


class MultiheadAttentionQ(LayerQ):
    def __init__(self, mha, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(mha, nn.MultiheadAttention, "MultiheadAttention")
        self.mha = mha
        self.do, _ = mha.out_proj.weight.shape
        self.head_dim = mha.embed_dim // mha.num_heads

        def aq():
            return get_activation_quantizer(gradient_based, n_bits=act_n_bits) if act_quant else _BypassQuantizer()

        self.activation_fake_quantize_q = aq()
        self.activation_fake_quantize_k = aq()
        self.activation_fake_quantize_v = aq()
        self.activation_fake_quantize_div = aq()
        self.activation_fake_quantize_attn = aq()
        self.activation_fake_quantize_softmax = aq()
        self.activation_fake_quantize_head = aq()
        self.weight_fake_quantize_in = (get_weight_quantizer(gradient_based, mha.in_proj_weight.shape, n_bits=weight_n_bits)
                                        if weight_quant else nn.Identity())
        self.weight_fake_quantize_out = (get_weight_quantizer(gradient_based, mha.out_proj.weight.shape, n_bits=weight_n_bits)
                                         if weight_quant else nn.Identity())

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None, need_weights=False, is_causal=False):
        if attn_mask is not None or key_padding_mask is not None:
            raise NotImplementedError("MultiheadAttentionQ: masks are ignored by the reference's forward (qat_layers.py:878-946)")
        aqs = (self.activation_fake_quantize_q, self.activation_fake_quantize_k, self.activation_fake_quantize_v,
               self.activation_fake_quantize_div, self.activation_fake_quantize_attn, self.activation_fake_quantize_softmax)
        if isinstance(aqs[0], _BypassQuantizer):
            aqs = None
        L = query.shape[1] if self.mha.batch_first else query.shape[0]
        if not (query is key and key is value) or self.mha.batch_first or L > 512 or self.head_dim > 32:
            # cross attention / batch-first rows / long sequences (HTDemucs transformer): the streaming attention core
            y = run_mha_x(self.mha, query, key, value, self.weight_fake_quantize_in(self.mha.in_proj_weight),
                          self.weight_fake_quantize_out(self.mha.out_proj.weight), aqs, self.activation_fake_quantize_head,
                          self.activation_fake_quantize)
            return (y,)
        _mha_check(self.mha, query, key, value)
        y = run_mha(self.mha, query, self.weight_fake_quantize_in(self.mha.in_proj_weight),
                    self.weight_fake_quantize_out(self.mha.out_proj.weight), aqs, self.activation_fake_quantize_head,
                    self.activation_fake_quantize)
        return (y,)


class LinearDecoderQ(LayerQ):
    """decoder basis Linear(E, W, bias=False) over [..., E] rows + the LSB residual channel (n_combiner = 2).
    forward() takes the reference's channels-last [B, S, L, E]; forward_cf() the channel-first [B*S, E, L] tensor the
    dual-path model holds at that point (a 1x1 conv: no transposing copy of the largest activation of the network)."""

    def __init__(self, decoder, n_combiner=1, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True,
                 inout_nl_quant=False, act_n_bits=8, out_quant=True, out_act_n_bits=8, train_res_dec=False):
        lin = decoder[0]
        _expect(lin, nn.Linear, "Linear")
        if lin.bias is not None:
            raise NotImplementedError("LinearDecoderQ: the dual-path decoders are bias-free")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=out_quant,
                         act_nl_quantizer=inout_nl_quant, weight_shape=lin.weight.shape, act_n_bits=out_act_n_bits,
                         weight_n_bits=weight_n_bits)
        self.linear = lin
        self.n_combiner = n_combiner
        if self.n_combiner >= 2:
            self.residual_error_block = ResidualErrorBlock(lin, gradient_based, weight_quant, act_quant, act_n_bits=act_n_bits,
                                                           weight_n_bits=weight_n_bits, train_res_dec=bool(train_res_dec))
            self.activation_fake_quantize_residual = (get_activation_quantizer(gradient_based, n_bits=out_act_n_bits)
                                                      if out_quant else _BypassQuantizer())

    def _run(self, x, lin_fn):
        w_dec = self._wq(self.linear.weight)
        if self.n_combiner == 1:
            return fq_node(self.activation_fake_quantize, lin_fn(x, w_dec))
        x_dec, x_res = ops.fork2(x)
        y = fq_node(self.activation_fake_quantize, lin_fn(x_dec, w_dec))
        outs = [y]
        rb = self.residual_error_block
        for _ in range(1, self.n_combiner):
            Y_q = lin_fn(y, rb._wq(rb.residual_encoder.weight))
            aq = rb.activation_fake_quantize
            q = aq.qctx()
            Y1 = ops.AddActQ.apply(ops.real(x_res), Y_q, q.qmin, q.qmax, -1.0, q)
            aq.after_forward(q)
            y = fq_node(self.activation_fake_quantize_residual, lin_fn(Y1, w_dec))
            outs.append(y)
        return outs

    def forward(self, x):
        outs = self._run(x, lambda t, w: ops_dp.RowLinear.apply(ops.real(t), w, None))
        return outs if self.n_combiner == 1 else torch.stack(outs)

    def forward_cf(self, x):
        """x [B', E, L] -> list of n_combiner tensors [B', W, L] (channel-first frames)"""
        def pw(t, w):
            L = ops._Lin("pw")
            ops_dp.touch(w)
            return ops.LinearActQ.apply(ops.real(t), ops.weight_view(w, w.shape[0], w.shape[1], 1), None, None, None, None, L, ops.ACT_NONE, ops.BYPASS)
        outs = self._run(x, pw)
        return [outs] if self.n_combiner == 1 else outs


# ---------------------------------------------------------------------------------------------
# first layers of cfg 5 (HTDemucs): the model is not built yet
# ---------------------------------------------------------------------------------------------
class LinearNlQ(LayerQ):
    """fq(nl(linear(x)))  (qat_layers.py:539-561; transformer feed-forward `linear1` + activation)"""

    def __init__(self, linear, nl, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(linear, nn.Linear, "Linear")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=linear.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.linear = linear
        self.nl = nl

    def forward(self, x):
        z = ops_dp.row_linear(x, self._wq(self.linear.weight), self.linear.bias)
        if isinstance(self.nl, (nn.GELU, nn.Tanh, nn.Sigmoid)):
            return fq_node(self.activation_fake_quantize, apply_map(self.nl, z))
        return fq_node(self.activation_fake_quantize, z, self.nl)


class DivQ(LayerQ):
    def __init__(self, div, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        _expect(div, Div, "Div")
        self.div = div

    def forward(self, x1, x2):
        if not torch.is_tensor(x2) or x1.shape != x2.shape:
            raise NotImplementedError("DivQ: only same-shape tensor operands have a HIP kernel")
        return fq_node(self.activation_fake_quantize, ops_dp.DivEw.apply(ops.real(x1), ops.real(x2)))


class EmbeddingQ(LayerQ):
    def __init__(self, embedding, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(embedding, nn.Embedding, "Embedding")
        if embedding.padding_idx is not None or embedding.max_norm is not None or embedding.scale_grad_by_freq or embedding.sparse:
            raise NotImplementedError("EmbeddingQ: only the plain lookup (no padding_idx / max_norm / sparse) has a HIP kernel")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=embedding.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.embedding = embedding

    def forward(self, x):
        return fq_node(self.activation_fake_quantize, ops_dp.EmbeddingRows.apply(self._wq(self.embedding.weight), x))


class Conv1dGnNlQ(LayerQ):
    """fq(nl(GroupNorm(conv1d(x))))  (qat_layers.py:222-259; DConv of the HTDemucs layers, demucsq.py:163-169)"""

    def __init__(self, conv1d, gn, nl, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(conv1d, nn.Conv1d, "Conv1d")
        _expect(gn, nn.GroupNorm, "GroupNorm")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv1d.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.conv1d = conv1d
        self.gn = gn
        self.nl = nl

    def forward(self, x):
        y = run_conv1d(self.conv1d, x, self._wq(self.conv1d.weight), None, None)
        z = run_groupnorm(self.gn, y, None)
        return fq_node(self.activation_fake_quantize, z, self.nl)


# ---------------------------------------------------------------------------------------------
# layers of the later §8 rows (HTDemucs)
# ---------------------------------------------------------------------------------------------
class Conv2dNlQ(LayerQ):
    """fq(nl(conv2d(x)))  (qat_layers.py:261-293; the frequency-branch encoder / rewrite convs of HTDemucs)"""

    def __init__(self, conv2d, nl, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(conv2d, nn.Conv2d, "Conv2d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv2d.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.conv2d = conv2d
        self.nl = nl

    def forward(self, x):
        return fq_node(self.activation_fake_quantize, conv_frames(self.conv2d, x, self._wq(self.conv2d.weight)), self.nl)


class _ConvTrQ(LayerQ):
    """fq(nl(conv_transpose(x)))  (qat_layers.py:296-435): per-channel weight ranges along dim 1 (ch_out_idx = 1)"""
    _attr, _typ = None, None

    def __init__(self, convtr, nl=None, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        _expect(convtr, self._typ, self._typ.__name__)
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=convtr.weight.shape, ch_out_idx=1, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        setattr(self, self._attr, convtr)
        if self._has_nl:
            self.nl = nl

    def forward(self, x, window=None):
        """window (convtr_frames): the caller keeps only that slice of the output.  Taken into the transposed convolution when the output
        quantizer is past its observer phase (the observers see the WHOLE output, qat_layers.py:296-435 + hdemucsq.py:340-345; fake-quant
        and the non-linearity are element-wise and commute with the crop); the caller crops whatever comes back un-cropped."""
        c = getattr(self, self._attr)
        aq = self.activation_fake_quantize
        if window is not None and _observing(aq):
            window = None
        return fq_node(aq, convtr_frames(c, x, self._wq(c.weight), window=window), self.nl if self._has_nl else None)


class ConvTranspose1dQ(_ConvTrQ):
    _attr, _typ, _has_nl = "convTr1d", nn.ConvTranspose1d, False

    def __init__(self, convTr1d, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        super().__init__(convTr1d, None, gradient_based, weight_quant, act_quant, act_n_bits, weight_n_bits)


class ConvTranspose2dQ(_ConvTrQ):
    _attr, _typ, _has_nl = "convTr2d", nn.ConvTranspose2d, False

    def __init__(self, convTr2d, gradient_based=True, weight_quant=True, act_quant=True, act_n_bits=8, weight_n_bits=8):
        super().__init__(convTr2d, None, gradient_based, weight_quant, act_quant, act_n_bits, weight_n_bits)


class ConvTranspose1dNlQ(_ConvTrQ):
    _attr, _typ, _has_nl = "convTr1d", nn.ConvTranspose1d, True


class ConvTranspose2dNlQ(_ConvTrQ):
    _attr, _typ, _has_nl = "convTr2d", nn.ConvTranspose2d, True


class Conv2dEncoderQ(LayerQ):
    """first frequency-branch conv with the n_splitter-wide input (qat_layers.py:1049-1102): channels >= in_channels of the widened
    kernel are drawn at random at construction, like Conv1dEncoderQ"""

    def __init__(self, encoder, n_splitter=1, gradient_based=True, weight_quant=True, act_quant=True, inout_nl_quant=False,
                 in_quant=False, act_n_bits=8, weight_n_bits=8, in_act_n_bits=8):
        conv = encoder[0]
        _expect(conv, nn.Conv2d, "Conv2d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                         weight_shape=conv.weight.shape, act_n_bits=act_n_bits, weight_n_bits=weight_n_bits)
        self.in_quantizer = (get_activation_quantizer(gradient_based, nl=inout_nl_quant, n_bits=in_act_n_bits)
                             if in_quant else nn.Identity())
        self.nl = nn.Identity() if len(encoder) == 1 else encoder[1]
        if n_splitter >= 2:
            w = conv.weight.detach()
            cin = conv.in_channels
            wide = nn.Conv2d(n_splitter * cin, conv.out_channels, conv.kernel_size, stride=conv.stride, padding=conv.padding,
                             bias=conv.bias is not None)
            new_w = w.repeat(1, n_splitter, 1, 1)
            for ch in range(1, n_splitter):
                for c in range(cin):
                    base = w[:, c, ...]
                    new_w[:, ch * cin + c, ...] = torch.mean(base) + torch.randn_like(base) * (torch.std(base) ** ch)
            with torch.no_grad():
                wide.weight.copy_(new_w)
                if conv.bias is not None:
                    wide.bias.copy_(conv.bias)
            conv = wide
        self.conv2d = conv

    def forward(self, x):
        x = self.in_quantizer(x)
        nl = None if isinstance(self.nl, nn.Identity) else self.nl
        return fq_node(self.activation_fake_quantize, conv_frames(self.conv2d, x, self._wq(self.conv2d.weight)), nl)


class ConvTr2dDecoderQ(LayerQ):
    """last frequency-branch transposed conv with the n_combiner residual outputs (qat_layers.py:1364-1418)"""

    def __init__(self, decoder, n_combiner=1, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True,
                 inout_nl_quant=False, act_n_bits=8, out_quant=True, out_act_n_bits=8, train_res_dec=False):
        conv = decoder[0]
        _expect(conv, nn.ConvTranspose2d, "ConvTranspose2d")
        super().__init__(gradient_based=gradient_based, weight_quant=weight_quant, act_quant=out_quant,
                         act_nl_quantizer=inout_nl_quant, weight_shape=conv.weight.shape, ch_out_idx=1,
                         act_n_bits=out_act_n_bits, weight_n_bits=weight_n_bits)
        self.n_combiner = n_combiner
        self.convTr2d = conv
        if self.n_combiner >= 2:
            self.residual_error_block = ResidualErrorBlock(conv, gradient_based, weight_quant=weight_quant, act_quant=act_quant,
                                                           weight_n_bits=weight_n_bits, act_n_bits=act_n_bits,
                                                           train_res_dec=bool(train_res_dec))
            self.activation_fake_quantize_residual = (get_activation_quantizer(gradient_based, n_bits=out_act_n_bits)
                                                      if out_quant else _BypassQuantizer())

    def forward(self, x, window=None):
        """window: as ConvTr1dDecoderQ.forward"""
        w_decoder = self._wq(self.convTr2d.weight)
        y = fq_node(self.activation_fake_quantize, convtr_frames(self.convTr2d, x, w_decoder))
        if self.n_combiner == 1:
            return y
        window = _decoder_window(self, window, self.convTr2d)
        outs = [y]
        for _ in range(1, self.n_combiner):
            y = self.residual_error_block(x, y, w_decoder, self.convTr2d, self.activation_fake_quantize_residual, window=window)
            outs.append(y)
        if window is not None:
            outs[0] = ops.real(outs[0]).narrow(*window[:2], window[2])
        return torch.stack(outs)


class BatchNormQ(LayerQ):
    """y = fq_act(batchnorm(x))  (qat_layers.py:472-486 of the reference; quantize_modules maps nn.BatchNorm1d / nn.BatchNorm2d here,
    qat_utils.py:163, 381-382).  None of the shipped configurations builds one; csrc/batchnorm.hip + ops_dp.BatchNormFn."""

    def __init__(self, batchnorm, gradient_based=True, act_quant=True, act_n_bits=8):
        super().__init__(gradient_based=gradient_based, act_quant=act_quant, act_n_bits=act_n_bits)
        if not isinstance(batchnorm, nn.BatchNorm1d) and not isinstance(batchnorm, nn.BatchNorm2d):
            raise Exception(f'Quantizing wrong layer instead of BatchNorm got:{type(batchnorm)}')
        self.batchnorm = batchnorm

    def forward(self, x):
        bn = self.batchnorm
        x = ops.real(x)
        shp = x.shape
        if x.dim() == 2:                                   # BatchNorm1d on [B, C]
            x3 = x.reshape(shp[0], shp[1], 1)
        elif x.dim() == 4:                                 # BatchNorm2d on [B, C, H, W]
            x3 = x.reshape(shp[0], shp[1], shp[2] * shp[3])
        else:
            x3 = x
        y = ops_dp.BatchNormFn.apply(x3, bn.weight, bn.bias, bn, bn.training)
        return fq_node(self.activation_fake_quantize, y.reshape(shp) if y.shape != shp else y)
