"""Autograd glue between the `quantization.qat` module API and the HIP kernels (fqss_amd/kernels.py).

One ``torch.autograd.Function`` per LayerQ-granularity op: forward = linear kernel + act/fake-quant
epilogue kernel, backward = STE/range-gradient kernel + input/weight gradient kernels.  No tensor
arithmetic is done with ATen here; torch is used for memory, streams and the autograd graph only.

Parameter gradients: when a parameter carries ``_fqss_direct = True`` and a pre-allocated ``.grad``
(see fqss_amd.runtime.ParamArena) the kernels accumulate ("+=") straight into it and the Function
returns ``None`` for that input; otherwise a fresh zero buffer is filled and returned to autograd.
"""
import os

import torch
from torch.autograd import Function

from . import kernels as K

Q_BYPASS, Q_OBSERVE, Q_QUANT = K.Q_BYPASS, K.Q_OBSERVE, K.Q_QUANT
ACT_NONE, ACT_PRELU, ACT_RELU = K.ACT_NONE, K.ACT_PRELU, K.ACT_RELU


class QCtx:
    """what the epilogue needs to know about the activation quantizer for ONE call (host-side only)"""
    __slots__ = ("qmode", "qmin", "qmax", "obs_ws", "gacc", "owner", "idx", "carrier", "keep_out", "prod", "no_codes", "stats", "defer", "hand", "presum", "chain", "link")

    def __init__(self, qmode=Q_BYPASS, qmin=None, qmax=None, obs_ws=None, gacc=None, owner=None):
        self.qmode, self.qmin, self.qmax, self.obs_ws, self.gacc, self.owner = qmode, qmin, qmax, obs_ws, gacc, owner
        self.idx = None         # u8 codes of the output produced by this call (QUANT mode)
        self.carrier = False    # True: the fp32 output of this call is an uninitialised carrier (codes-only fast path)
        self.keep_out = False   # force a real fp32 output even in the fast path (model outputs)
        self.prod = None        # _Producer: lets the NEXT layer's backward run this layer's epilogue backward (see below)
        self.no_codes = False   # the caller has no coded consumer (dual-path row layers): do not emit the u8 codes at all
        self.stats = None       # kernels.CodeStats of the output codes, emitted by the producing kernel for a GroupNormQ consumer
        self.defer = None       # GroupNormQ in front of a depthwise layer: its launch record, run by that layer's kernel (GroupNormActQ)
        self.hand = None        # _GnHand: backward hand-over between a GroupNormQ and the depthwise layer next to it
        self.presum = None      # _PreSum: the AddQ that consumes this output was already evaluated by the producing GEMM
        self.chain = False      # AddQ: the caller declares that the first operand is the previous add's output and has no other consumer
        self.link = None        # _ChainLink of this call's output (EwQ): the next add of the chain finds it on the tensor


class ActCodes:
    """u8 bin indices of a fake-quantized activation + the ranges of the quantizer that made them;
    travels as the `_fqss_q` attribute of the fp32 tensor so that a consuming q-GEMM can use it"""
    __slots__ = ("idx", "qmin", "qmax")

    def __init__(self, idx, qmin, qmax):
        self.idx, self.qmin, self.qmax = idx, qmin, qmax


def tag_codes(y, q):
    """attach the codes produced by the epilogue of this call to its output tensor"""
    if q.idx is not None and not CODED:
        assert not q.carrier, "codes-only carriers need the coded dataflow"
        q.idx = q.prod = q.stats = q.hand = q.presum = q.link = None          # un-fused reference dataflow (tests): consumers see plain fp32 tensors
    if q.idx is not None:
        y._fqss_q = ActCodes(q.idx, q.qmin.detach(), q.qmax.detach())
        y._fqss_carrier = q.carrier
        q.idx = None
        if q.prod is not None:
            y._fqss_prod, q.prod = q.prod, None
        if q.stats is not None:
            y._fqss_stats, q.stats = q.stats, None
        if q.defer is not None:
            y._fqss_defer, q.defer = q.defer, None
        if q.hand is not None:
            y._fqss_hand, q.hand = q.hand, None
        if q.presum is not None:
            y._fqss_presum, q.presum = q.presum, None
        if q.link is not None:
            y._fqss_chain, q.link = q.link, None
    return y


class _Producer:
    """What the NEXT layer needs in order to run THIS layer's epilogue backward (output fake-quant STE + non-linearity)
    inside its own backward kernel: handed over only along a HipSequential edge whose consumer is a GroupNormQ
    (NEXT_IS_GROUPNORM), i.e. where the produced tensor provably has that single consumer."""
    __slots__ = ("z", "act", "slope", "slope_param", "q", "bias_param", "bias", "fused")

    def __init__(self, z, act, slope, slope_param, q, bias_param, bias):
        self.z, self.act, self.slope, self.slope_param, self.q, self.bias_param, self.bias = z, act, slope, slope_param, q, bias_param, bias
        self.fused = False   # set by the consumer's backward: the gradient it returned already IS this layer's gz


class _GnHand:
    """Backward hand-over between a GroupNormQ and the 3-tap depthwise Conv1dNlQ next to it along a HipSequential edge (one consumer),
    both quantizing (round 5; csrc/fused_q.hip k_dwq_bwd<3, GA, GB>).  The GroupNorm's two-pass backward is bound by its bytes, and the
    depthwise backward owns a whole (b, c) row per workgroup:
      kind "after"  (tagged on the DEPTHWISE layer's output, found by the GroupNormQ that consumes it): that GroupNormQ's backward
                    runs its rows pass only, leaves `rec` (its parameters + row sums) here and returns the INCOMING gradient
                    unchanged; the depthwise backward applies the second pass while it loads that gradient;
      kind "before" (tagged on the GROUPNORM's output, found by the depthwise layer that consumes it): `fwd` holds what the
                    GroupNorm's rows pass needs; the depthwise backward takes that pass on the gx it produces and leaves the row
                    sums in `rec`; the GroupNorm's backward then only runs its apply pass."""
    __slots__ = ("kind", "fwd", "rec")

    def __init__(self, kind, fwd=None):
        self.kind, self.fwd, self.rec = kind, fwd, None


FUSE_GN_BWD_DW = os.environ.get("FQSS_FUSE_GN_BWD_DW", "1") != "0"
_GN_HAND = os.environ.get("FQSS_GN_HAND", "both")          # experiments: "after" / "before" alone
FUSE_GN = os.environ.get("FQSS_FUSE_GN", "1") != "0"
FUSE_EW = os.environ.get("FQSS_FUSE_EW", "1") != "0"
# round 5: the forward of the AddQ behind each output of a res | skip pair runs in the pair GEMM's epilogue (fqss_qpw_fwdq_add)
FUSE_ADD_FWD = os.environ.get("FQSS_FUSE_ADD_FWD", "1") != "0"


# round 5: the backward of a CHAIN of AddQ layers (the skip sum of the TCN stack) as ONE launch (fqss_add_chain_bwd)
FUSE_ADD_CHAIN = os.environ.get("FQSS_FUSE_ADD_CHAIN", "1") != "0"


class _ChainLink:
    """One AddQ of a chain out_l = fq(out_{l-1} + b_l) whose out_{l-1} has no other consumer (declared by the caller: QCtx.chain):
    what the chain kernel needs of this level, the link of the level below, and -- once the backward of a level ABOVE has run the
    whole chain -- this level's finished results (`pre`), which its own autograd node then only hands on."""
    __slots__ = ("prev", "ac", "amin", "amax", "bc", "bmin", "bmax", "qmin", "qmax", "q", "prod_a", "prod_b", "pre", "dummy")

    def __init__(self, prev, ac, amin, amax, bc, bmin, bmax, qmin, qmax, q, prod_a, prod_b):
        self.prev, self.ac, self.amin, self.amax, self.bc, self.bmin, self.bmax = prev, ac, amin, amax, bc, bmin, bmax
        self.qmin, self.qmax, self.q, self.prod_a, self.prod_b = qmin, qmax, q, prod_a, prod_b
        self.pre = self.dummy = None


class _PreSum:
    """Codes of fq(dec(a) + dec(y)) that the GEMM which produced y already wrote for the AddQ `owner` (its quantizer module) whose
    other operand has the codes tensor `a_idx`; rides on y as `_fqss_presum`, EwQ.forward takes it when exactly that layer arrives"""
    __slots__ = ("codes", "a_idx", "owner")

    def __init__(self, codes, a_idx, owner):
        self.codes, self.a_idx, self.owner = codes, a_idx, owner
FUSE_STATS = os.environ.get("FQSS_FUSE_STATS", "1") != "0"   # gLN statistics from the epilogue of the kernel that makes its input codes
NEXT_IS_GROUPNORM = False   # set by HipSequential around the forward of a module followed by a GroupNormQ
NEXT_IS_DW3 = False         # ... around the forward of a GroupNormQ followed by a 3-tap depthwise Conv1dNlQ (one launch for both: FUSE_GN_DW)
# OPT-IN (FQSS_FUSE_GN_DW=1), measured and NOT the default: the one-launch form is bit-identical (tests/test_gpu_kernels.py::
# test_gn_dw_fused_bit_identical) and SLOWER -- 39.5 us against 34.5 us for fqss_gnq_fwd + fqss_dwq_fwd at the cfg-2 shape (tools/
# gndw_probe.py), the cfg-2 step unchanged (15.17 vs 15.16 ms on one box): both passes are bound by vector-ALU issue, so a fusion saves
# the launch and a 16-MB re-read but not the instructions, and the LDS row (two barriers per row, two aligned reads + a byte-align per
# four codes instead of one unaligned global load) adds some.  VERDICT r03 item 1(d) asked for this fusion.
# Since round 5 the kernel lives in the experiments build only (include/fqss_experiments.h; `make -C fqss_amd/csrc experiments` + FQSS_LIB).
FUSE_GN_DW = __import__("os").environ.get("FQSS_FUSE_GN_DW", "0") != "0"


def flush_defer(x):
    """a GroupNormQ output whose launch was left to a depthwise consumer that cannot take it after all: launch it now"""
    d = getattr(x, "_fqss_defer", None)
    if d is not None and not d["done"]:
        K.gnq_fwd_into(d["xc"], d["qmin_x"], d["qmax_x"], d["gamma"], d["beta"], d["eps"], d["qmin"], d["qmax"], d["stats"], d["yc"], d["mean_rstd"])
        d["done"] = True

NEXT_TAKES_PRODUCER = False  # set by HipSequential around its last module when the sequence's output has ONE consumer, a coded MulQ
FUSE_MULQ_PROD = os.environ.get("FQSS_FUSE_MULQ_PROD", "1") != "0"


def _producer_args(pr):
    """(z, act, slope, gacc, gbias) for the fused-producer kernels, or None when the partials cannot go in place"""
    if pr is None or pr.q.gacc is None:
        return None
    pgb, direct = (None, True) if pr.bias is None else _grad_buf(pr.bias_param, pr.bias)
    return (pr.z, pr.act, pr.slope, pr.q.gacc, pgb) if direct else None


def codes_of(x):
    return getattr(x, "_fqss_q", None)


def is_carrier(x):
    return getattr(x, "_fqss_carrier", False)


# ---- codes-only fast path -----------------------------------------------------------------------
# With FAST on, a quantizing layer writes ONLY the u8 codes of its output; the fp32 tensor it returns is
# an uninitialised "carrier" that keeps shapes/autograd intact.  Consumers with a coded-input kernel read
# the codes; every other consumer calls real(x) first.  Off by default (module outputs are then real
# fp32, like the reference's); fqss_amd.runtime.KDTrainStep turns it on around the student forward.
FAST = False
CODED = True            # False (tests, `coded_dataflow(False)`): no layer output carries codes -> every layer runs its un-fused fp32 kernels
DEBUG_POISON = bool(int(__import__("os").environ.get("FQSS_DEBUG_CARRIER", "0")))   # NaN-fill carriers (tests)


# ---- deferred per-quantizer work ------------------------------------------------------------------
# When a runtime.QuantTables is active (DEFER), the per-call gacc flush and the per-weight fake-quant
# backward are NOT launched by the Functions; the stepper runs them once per step as multi-tensor kernels.
DEFER = None


class deferred:
    def __init__(self, tables):
        self.t = tables

    def __enter__(self):
        global DEFER
        self.prev, DEFER = DEFER, self.t

    def __exit__(self, *a):
        global DEFER
        DEFER = self.prev


class fast_codes:
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global FAST
        self.prev, FAST = FAST, self.on

    def __exit__(self, *a):
        global FAST
        FAST = self.prev


class coded_dataflow:
    """coded_dataflow(False): layer outputs do not carry their u8 codes, so every consumer takes its un-fused fp32 kernels (the
    per-layer path the G1 fixtures pin); used by the tests that pin the fused codes-only step against it"""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global CODED
        self.prev, CODED = CODED, self.on

    def __exit__(self, *a):
        global CODED
        CODED = self.prev


class poison_carriers:
    """NaN-fill every codes-only carrier inside the block (what FQSS_DEBUG_CARRIER=1 does process-wide)"""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global DEBUG_POISON
        self.prev, DEBUG_POISON = DEBUG_POISON, self.on

    def __exit__(self, *a):
        global DEBUG_POISON
        DEBUG_POISON = self.prev


def _carrier(out):
    if DEBUG_POISON:
        out.fill_(float("nan"))
    return out


class Materialize(Function):
    """carrier -> real fp32 tensor (decode kernel); gradient passes through"""

    @staticmethod
    def forward(ctx, x, xq):
        return K.decode(xq.idx, xq.qmin, xq.qmax)

    @staticmethod
    def backward(ctx, g):
        return g, None


def real(x):
    """the fp32 values of x (decodes a carrier; no-op for ordinary tensors)"""
    if torch.is_tensor(x) and is_carrier(x):
        flush_defer(x)
        xq = codes_of(x)
        y = Materialize.apply(x, xq)
        y._fqss_q, y._fqss_carrier = xq, False
        return y
    return x


def reshape_tagged(x, *shape):
    """x.reshape(shape) that keeps the codes (reshaped alike) and the carrier flag"""
    y = x.reshape(*shape)
    xq = codes_of(x)
    if xq is not None:
        y._fqss_q = ActCodes(xq.idx.reshape(*shape), xq.qmin, xq.qmax)
        y._fqss_carrier = is_carrier(x)
        pr = getattr(x, "_fqss_prod", None)
        if pr is not None:
            y._fqss_prod = pr        # a reshape is not a second consumer: the producer record travels with the codes
    return y


# ---- backward segments ------------------------------------------------------------------------------
# At world > 1 runtime.KDTrainStep splits the backward at the model's cut points into separately launched segments, so that the
# gradient all-reduce of a finished segment (RCCL, side stream) overlaps the backward of the next one (the reference: DDP's
# bucketed all-reduce overlapped with backward, asteroid_librimix_trainer.py:125-135).  cut() is the identity otherwise.
CUTS = None


class cut_recorder:
    def __enter__(self):
        global CUTS
        self.prev, CUTS = CUTS, []
        return CUTS

    def __exit__(self, *a):
        global CUTS
        CUTS = self.prev


_TAGS = ("_fqss_q", "_fqss_carrier", "_fqss_prod", "_fqss_stats", "_fqss_rowq", "_fqss_defer", "_fqss_hand")


def cut(*tensors, late=False):
    """a point where the backward may be split: returns fresh autograd leaves (same storage, same code tags); the recorder keeps
    (original tensors, leaves, late) so the stepper can push the leaves' gradients into the original graph later.  late=True: a
    side edge that skips over all segments (ConvTasNet: the encoder output feeding the mask multiply at the end of the network);
    its gradient is pushed with the LAST backward segment, together with that segment's own roots"""
    if CUTS is None or not torch.is_grad_enabled():
        return tensors
    leaves = []
    for t in tensors:
        if t is None or not t.requires_grad:
            leaves.append(t)
            continue
        leaf = t.detach().requires_grad_(True)
        for a in _TAGS:
            if hasattr(t, a):
                setattr(leaf, a, getattr(t, a))
        leaves.append(leaf)
    CUTS.append((tensors, tuple(leaves), late))
    return tuple(leaves)


FLAT_CM = os.environ.get("FQSS_FLAT_CM", "1") != "0"        # (A/B knob: "0" = reshape(B, C, -1), copying what is not a view)


def flat_cm(x):
    """a channel-first tensor [B, C, ...] as [B, C', M] WITHOUT a copy, for maps that pair or walk channels (GLU, per-channel streams):
    [B, C, H W] when the trailing dims are dense; [B, C H, W] when rows of W are `ld` apart with the planes dense in rows -- the
    pitch-Wp outputs of the halo-packed convolutions (ops_dp.ConvHalo), row-padded activations -- where `reshape(B, C, -1)` would copy the
    tensor.  Channel halves stay halves (C H = 2 (C / 2) H)."""
    if x.dim() <= 3:
        return x
    B, C = x.shape[0], x.shape[1]
    if x.dim() == 4 and FLAT_CM:
        _, _, H, W = x.shape
        st = x.stride()
        if st[3] == 1 and st[2] == W and st[1] >= H * W:
            return x.reshape(B, C, H * W)            # (a view)
        if st[3] == 1 and st[1] == H * st[2] and (B == 1 or st[0] >= C * st[1]):
            return x.as_strided((B, C * H, W), (st[0], st[2], 1))
    return x.reshape(B, C, -1)


def weight_view(w, *shape):
    """w.view(shape) of a (possibly fake-quantized) weight that keeps the dL/dW_q arena slot of a weight fake-quantized by
    runtime.QuantTables: such a tensor has no autograd history, its consumers accumulate its gradient into `_fqss_gwq`"""
    v = w.view(*shape)
    gwq = getattr(w, "_fqss_gwq", None)
    if gwq is not None:
        v._fqss_gwq = gwq.view(*shape)
    return v


BYPASS = QCtx()


def _grad_buf(param, like):
    """(buffer, direct): direct -> accumulate in place into param.grad and return None to autograd"""
    if param is not None:
        param._fqss_touched = True      # the arena starts this parameter's Adam step count (torch: grad is not None)
        if getattr(param, "_fqss_direct", False) and param.grad is not None:
            return param.grad, True
    return torch.zeros_like(like), False


def _wgrad_temp(w):
    """zeroed accumulator for a weight gradient that autograd carries on (un-fused layers: no arena slot to add into).  Under
    FQSS_DETERMINISTIC=1 it is cut from the DetMode's pool, so that the wgrad kernel's split adds land on the integer shadow"""
    d = K.DetMode.owner
    t = d.temp_like(w) if d is not None else None
    return t if t is not None else torch.zeros_like(w)


def _wgrad_temp_done(gw, gwq):
    """behind the kernels that added into a _wgrad_temp: round its integer sums into it before autograd hands it on"""
    d = K.DetMode.owner
    if d is not None and gwq is None and gw is not None:
        pool = d.arenas[2]
        if pool.data_ptr() <= gw.data_ptr() < pool.data_ptr() + 4 * pool.numel():
            d.finish_temp(gw)


def _epilogue_fwd(z, act, slope, q):
    if q.qmode == Q_QUANT and q.no_codes:
        q.carrier = False
        return K.actq_fwd(z, act, slope, q.qmode, q.qmin, q.qmax, q.obs_ws)
    if q.qmode == Q_QUANT:
        q.carrier = FAST and not q.keep_out
        out, q.idx = K.actq_fwd(z, act, slope, q.qmode, q.qmin, q.qmax, q.obs_ws, want_idx=True, write_out=not q.carrier)
        return _carrier(out) if q.carrier else out
    return K.actq_fwd(z, act, slope, q.qmode, q.qmin, q.qmax, q.obs_ws)


def _fuse_out_quant(q):
    """codes-only mode with a learned-range quantizer: the q-GEMM epilogue can emit the output codes itself"""
    return q.qmode == Q_QUANT and FAST and not q.keep_out


def _touch(*params):
    for p in params:
        if p is not None:
            p._fqss_touched = True


def _flush_ranges(q, slope, slope_param, act):
    """fp64 partial slots -> fp32 parameter gradients (ranges always, slope for PReLU)"""
    if q.owner is not None and getattr(q.owner, "_fqss_deferred", False):
        _touch(slope_param if act == ACT_PRELU else None, q.owner.min_range, q.owner.max_range)
        return None, None, None        # flushed once per step by fqss_gacc_flush_multi
    s_buf = None
    s_direct = True
    if act == ACT_PRELU:
        s_buf, s_direct = _grad_buf(slope_param, slope)
    mn_buf, mn_direct = _grad_buf(q.owner.min_range if q.owner is not None else None, q.qmin)
    mx_buf, mx_direct = _grad_buf(q.owner.max_range if q.owner is not None else None, q.qmax)
    K.gacc_flush(q.gacc, mn_buf, mx_buf, s_buf)
    return (None if s_direct else s_buf), (None if mn_direct else mn_buf), (None if mx_direct else mx_buf)


def _ranges_after(q, gacc):
    """what _epilogue_bwd does behind its kernel, for a quantizer (no non-linearity) whose range partials ANOTHER kernel left in gacc:
    deferred tables -> mark the range parameters touched; else flush into the parameters' gradients -> (g_min, g_max) for autograd"""
    if q.owner is not None and getattr(q.owner, "_fqss_deferred", False):
        _touch(None, q.owner.min_range, q.owner.max_range)
        return None, None
    mn_buf, mn_direct = _grad_buf(q.owner.min_range if q.owner is not None else None, q.qmin)
    mx_buf, mx_direct = _grad_buf(q.owner.max_range if q.owner is not None else None, q.qmax)
    K.gacc_flush(gacc, mn_buf, mx_buf, None)
    return (None if mn_direct else mn_buf), (None if mx_direct else mx_buf)


def _epilogue_bwd(z, g, act, slope, slope_param, q, bias_param=None, bias_like=None, C=0, out=None):
    """returns gz and the autograd gradients (g_slope, g_qmin, g_qmax, g_bias); out: row-matrix view that receives gz"""
    need_acc = (q.qmode == Q_QUANT) or (act == ACT_PRELU)
    gacc = None
    if need_acc:
        gacc = q.gacc if q.gacc is not None else torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=z.device)
    gb, gb_direct = (None, True)
    if bias_like is not None:
        gb, gb_direct = _grad_buf(bias_param, bias_like)
    gz = K.actq_bwd(z, g, act, slope, q.qmode, q.qmin, q.qmax, gacc, gbias=gb, C=C, out=out)
    g_slope = g_min = g_max = None
    if need_acc and q.owner is not None and getattr(q.owner, "_fqss_deferred", False):
        _touch(slope_param if act == ACT_PRELU else None, q.owner.min_range if q.qmode == Q_QUANT else None,
               q.owner.max_range if q.qmode == Q_QUANT else None)
    elif need_acc:
        s_buf = mn_buf = mx_buf = None
        s_direct = mn_direct = mx_direct = True
        if act == ACT_PRELU:
            s_buf, s_direct = _grad_buf(slope_param, slope)
        if q.qmode == Q_QUANT:
            mn_buf, mn_direct = _grad_buf(q.owner.min_range if q.owner is not None else None, q.qmin)
            mx_buf, mx_direct = _grad_buf(q.owner.max_range if q.owner is not None else None, q.qmax)
        K.gacc_flush(gacc, mn_buf, mx_buf, s_buf)
        g_slope = None if s_direct else s_buf
        g_min = None if mn_direct else mn_buf
        g_max = None if mx_direct else mx_buf
    return gz, g_slope, g_min, g_max, (None if gb_direct else gb)


def _plain_or_bwd(z, g, q):
    """float (BYPASS) producers have no epilogue: the gradient passes through untouched"""
    if q.qmode == Q_BYPASS:
        return g, None, None, None, None
    return _epilogue_bwd(z, g, ACT_NONE, None, None, q)


# ----------------------------------------------------------------------------------------------
# linear kinds
# ----------------------------------------------------------------------------------------------
class _Lin:
    """geometry of the linear op in front of the epilogue"""
    __slots__ = ("kind", "stride", "dil", "pad", "w_param", "b_param", "slope_param", "six", "taps", "wc_dgrad")

    def __init__(self, kind, stride=1, dil=1, pad=0, w_param=None, b_param=None, slope_param=None, six=False, taps=1):
        self.kind, self.stride, self.dil, self.pad = kind, stride, dil, pad
        self.six = six          # "pw": forward on the six-product split GEMM (K.pwconv_fwd)
        self.taps = taps        # "conv1": stride-1 1-D convolution as an implicit GEMM (K.conv1d_s1_*), weight [Co, Ci * taps, 1]
        self.wc_dgrad = None    # "pw", float input: int8 codes of the (fake-quantized) weight for the DATA gradient alone (three products per k)
        self.w_param, self.b_param, self.slope_param = w_param, b_param, slope_param


FRAME_CODES_FWD = os.environ.get("FQSS_FRAME_CODES_FWD", "1") != "0"


def _lin_fwd(L, x, w, bias):
    if L.kind == "pw":
        if L.wc_dgrad is not None and FRAME_CODES_FWD and x.dim() == 3:
            # the weight is on its int8 grid (runtime.QuantTables), the input a plain float tensor: three products per k (k_qgemm<4>)
            z = K.pwconv_fwd_wq(x, L.wc_dgrad, bias)
            if z is not None:
                return z
        return K.pwconv_fwd(x, w, bias, L.six)
    if L.kind == "conv1":
        return K.conv1d_s1_fwd(x, w.reshape(w.shape[0], -1), bias, L.taps, L.dil, L.pad)
    if L.kind == "dw":
        return K.dwconv_fwd(x, w, bias, L.dil, L.pad)
    if L.kind == "frames":       # strided framing conv (encoder), no bias in the networks served
        assert bias is None
        return K.frames_conv_fwd(x, w, L.stride)
    if L.kind == "convtr":       # transposed conv + overlap-add (decoder), Co = 1
        assert bias is None and w.shape[1] == 1
        return K.ola_convtr_fwd(x, w, L.stride)
    raise NotImplementedError(L.kind)


def _lin_bwd_x(L, gz, w, x_shape):
    if L.kind == "pw":
        return K.pwconv_bwd_x(gz, w, x_shape[1])
    if L.kind == "conv1":
        # the transposed convolution of a stride-1 conv is a stride-1 conv: taps flipped, weight [Ci][Co * taps], pad' = dil (taps - 1) - pad
        Co, Ci, T = w.shape[0], x_shape[1], L.taps
        wt = w.reshape(Co, Ci, T).flip(2).permute(1, 0, 2).reshape(Ci, Co * T).contiguous()
        return K.conv1d_s1_fwd(gz, wt, None, T, L.dil, L.dil * (T - 1) - L.pad)
    if L.kind == "dw":
        return K.dwconv_bwd_x(gz, w, L.dil, L.pad)
    if L.kind == "frames":
        # transposed conv of gz, one input channel at a time (Ci = 1 on the training path: the residual
        # encoder; the waveform-side encoder input never needs a gradient)
        parts = [K.ola_convtr_fwd(gz, w[:, ci, :].contiguous(), L.stride) for ci in range(x_shape[1])]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    if L.kind == "convtr":
        return K.frames_conv_fwd(gz, w.reshape(w.shape[0], 1, w.shape[2]), L.stride)
    raise NotImplementedError(L.kind)


def _lin_bwd_w(L, gz, x, gw):
    if L.kind == "pw":
        K.pwconv_bwd_w(gz, x, gw)
    elif L.kind == "conv1":
        K.conv1d_s1_bwd_w(gz, x, gw, L.taps, L.dil, L.pad)
    elif L.kind == "dw":
        K.dwconv_bwd_w(gz, x, gw, L.dil, L.pad)
    elif L.kind == "frames":
        K.frames_wgrad(gz, x, gw, L.stride)
    elif L.kind == "convtr":
        K.frames_wgrad(x, gz, gw, L.stride)
    else:
        raise NotImplementedError(L.kind)


FRAME_CODES_DGRAD = os.environ.get("FQSS_FRAME_CODES_DGRAD", "1") != "0"
PLAIN_BIAS_SUM = os.environ.get("FQSS_PLAIN_BIAS_SUM", "1") != "0"       # (A/B knob: "0" = the bias gradient of a float conv from the epilogue pass)


class LinearActQ(Function):
    """out = fq(act(linear(x, w) + bias))  -- Conv1dQ / Conv1dNlQ / Conv1dEncoderQ / decoder convT"""

    @staticmethod
    def forward(ctx, x, w, bias, slope, qmin, qmax, L, act, q, xq=None, wc=None):
        # grid-valued operands (student, quantizing phase): exact bf16-MFMA GEMM on the codes
        ctx.xq = xq if (L.kind == "pw" and xq is not None and K.q_eligible(w.shape[1], w.shape[0])) else None
        ctx.wc = wc if (L.kind == "pw" and wc is not None and K.q_eligible(w.shape[1], w.shape[0])) else None
        ctx.plain = (q.qmode == Q_BYPASS and act == ACT_NONE)   # float linear op: no epilogue pass at all
        # decoder on a coded input (MulQ's output, the residual block's quantized error): de-quantised inside the overlap-add kernel
        ctx.xq_tr = xq if (L.kind == "convtr" and xq is not None and bias is None and K.ola_convtr_ok(w, L.stride)) else None
        if ctx.xq is not None and ctx.wc is not None and _fuse_out_quant(q):
            # the layer's own non-linearity + fake-quant ride in the GEMM epilogue: z (for the backward) and the
            # output codes come out of one launch
            # ... and, for a GroupNormQ consumer, the integer statistics of those codes (no statistics pass in the gLN)
            q.stats = K.new_stats("qpw", x.shape[0], ctx.wc.Co, x.shape[-1], x.device) if (FUSE_STATS and NEXT_IS_GROUPNORM) else None
            z, q.idx = K.qpw_fwdq(ctx.xq.idx, ctx.wc, bias, None, ctx.xq.qmin, ctx.xq.qmax, ctx.wc.Co, act, slope, (q.qmin, q.qmax),
                                  stats=q.stats)
            q.carrier = True
            out = _carrier(K.empty_act(tuple(z.shape), z.device))
        else:
            if ctx.xq is not None and ctx.wc is not None:
                z = K.qpw_fwd(ctx.xq.idx, ctx.wc, bias, ctx.xq.qmin, ctx.xq.qmax)
            elif ctx.xq_tr is not None:
                z = K.ola_convtr_fwd_q(ctx.xq_tr.idx, ctx.xq_tr.qmin, ctx.xq_tr.qmax, w, L.stride)
            else:
                z = _lin_fwd(L, x, w, bias)
            out = z if ctx.plain else _epilogue_fwd(z, act, slope, q)
        ctx.prod = None
        if ((FUSE_GN and NEXT_IS_GROUPNORM) or (FUSE_MULQ_PROD and NEXT_TAKES_PRODUCER and act in (ACT_NONE, ACT_RELU, ACT_PRELU))) \
                and DEFER is not None and q.qmode == Q_QUANT and not ctx.plain and q.owner is not None \
                and getattr(q.owner, "_fqss_deferred", False):
            ctx.prod = q.prod = _Producer(z, act, slope, L.slope_param, q, L.b_param, bias)
        ctx.x_shape = x.shape
        ctx.fork = getattr(x, "_fqss_fork", None) if ((ctx.wc is not None and ctx.wc.idxT is not None) or L.kind == "convtr") else None
        ctx.save_for_backward(None if ((ctx.xq is not None and ctx.wc is not None) or ctx.xq_tr is not None) else x, w,
                              None if ctx.plain else z, slope)
        ctx.L, ctx.act, ctx.q, ctx.has_bias = L, act, q, bias is not None
        ctx.bias_like = bias
        ctx.C = z.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, z, slope = ctx.saved_tensors
        L, act, q = ctx.L, ctx.act, ctx.q
        if ctx.prod is not None and ctx.prod.fused:
            # the consumer's backward (fqss_gnq_bwd_p) already pushed its gradient through this layer's output quantizer
            # and non-linearity and accumulated the range / slope / bias partials: g IS gz
            gz, g_slope, g_min, g_max, g_bias = g, None, None, None, None
            _touch(L.slope_param if act == ACT_PRELU else None, q.owner.min_range, q.owner.max_range, L.b_param)
        elif ctx.plain and not ctx.has_bias:
            gz, g_slope, g_min, g_max, g_bias = g, None, None, None, None
        elif ctx.plain and PLAIN_BIAS_SUM and g.dim() == 3 and g.is_cuda and L.kind in ("pw", "conv1") and K.rowmat(g) is not None \
                and K.rowmat(g)[2] % 4 == 0 and g.data_ptr() % 16 == 0:
            # a float (BYPASS) channel-first conv with a bias -- the DConv / frame-path convolutions of HTDemucs: the bias gradient is the
            # channel sum of g (one read-only pass, fqss_chan_sum) and g itself is gz; the epilogue kernel wrote a COPY of g to get it
            gb, gb_direct = _grad_buf(L.b_param, ctx.bias_like)
            K.chan_sum(g, gb)
            gz, g_slope, g_min, g_max, g_bias = g, None, None, None, (None if gb_direct else gb)
        else:
            gz, g_slope, g_min, g_max, g_bias = _epilogue_bwd(
                g if ctx.plain else z, g, act, slope, L.slope_param, q, bias_param=L.b_param,
                bias_like=ctx.bias_like if ctx.has_bias else None, C=ctx.C)
        gx = None
        if ctx.needs_input_grad[0]:
            fk = ctx.fork
            other = fk[0].take(fk[1], ctx.x_shape) if (fk is not None and (ctx.wc is not None or L.kind == "convtr")) else None
            if ctx.wc is not None and other is not None:
                gx = K.qpw_bwd_x(gz, ctx.wc, add=other)     # + the gradient of the fork's other branch: no separate sum pass
            elif L.kind == "convtr" and other is not None:
                gx = K.frames_conv_fwd(gz, w.reshape(w.shape[0], 1, w.shape[2]), L.stride, add=other)   # decoder: same, in its dgrad
            elif ctx.wc is not None:
                gx = K.qpw_bwd_x(gz, ctx.wc)
            elif L.kind == "pw" and L.wc_dgrad is not None and FRAME_CODES_DGRAD and gz.dim() == 3:
                # round 5 (cfg 5): the frame-path convolutions of HTDemucs read FLOAT inputs, but their weight is fake-quantized: W_q =
                # dw * Wi with int8 Wi, so the data gradient W_q^T gz is the q-GEMM of the ConvTasNet path (k_qgemm<1>: gz in three exact
                # bf16 pieces x ONE exact plane of codes = three products per k instead of the six of the float x float form)
                gx = K.qpw_bwd_x(gz, L.wc_dgrad)
            else:
                gx = _lin_bwd_x(L, gz, w, ctx.x_shape)
        gw = None
        gwq = getattr(w, "_fqss_gwq", None)    # deferred mode: dL/dW_q accumulates in the step's arena
        if ctx.needs_input_grad[1] or gwq is not None:
            # w here is the fake-quantized weight (a non-leaf): its gradient goes back through autograd, or
            # (deferred) into the arena consumed by fqss_wq_multi_bwd
            gw = gwq if gwq is not None else _wgrad_temp(w)
            if ctx.xq is not None:
                wq = getattr(DEFER, "wgrad_queue", None) if (gwq is not None and getattr(w, "_fqss_gwq_done", None) is None) else None
                if wq is not None:     # the segment's weight gradients run together when it is done (runtime.QuantTables.finish_backward)
                    wq.push(gz, None, ctx.xq.idx, ctx.xq.qmin, ctx.xq.qmax, gw)
                else:
                    K.qpw_bwd_w(gz, ctx.xq.idx, ctx.xq.qmin, ctx.xq.qmax, gw)
            else:
                if x is None and K.frames_wgrad1_q(ctx.xq_tr.idx, ctx.xq_tr.qmin, ctx.xq_tr.qmax, gz, gw, L.stride):
                    pass             # coded decoder input: dL/dW straight from the codes
                else:
                    if x is None:
                        x = K.decode(ctx.xq_tr.idx, ctx.xq_tr.qmin, ctx.xq_tr.qmax)
                    _lin_bwd_w(L, gz, x, gw)
            _wgrad_temp_done(gw, gwq)
            if L.w_param is not None and w is L.w_param:
                L.w_param._fqss_touched = True
            if gwq is not None:
                gw = None
                done = getattr(w, "_fqss_gwq_done", None)     # a re-laid-out copy of the weight (convtr_frames): fold its gradient back
                if done is not None:
                    done()
        return gx, gw, g_bias, g_slope, g_min, g_max, None, None, None, None, None


class LinearActQPair(Function):
    """Two quantized pointwise convs on the SAME coded input (res | skip of a TCN block) as one node:
    one forward GEMM over the concatenated output channels, one dgrad GEMM whose reduction runs over both
    layers' gradients (= autograd's accumulation at the fork, without the extra pass), one wgrad launch.
    Only in the quantizing phase with a runtime.QuantTables active (concatenated weight codes)."""

    @staticmethod
    def forward(ctx, x, b1, b2, qmin1, qmax1, qmin2, qmax2, L1, L2, q1, q2, xq, pair, sole_ew=False, adds=None):
        Co1 = pair.Co1
        if _fuse_out_quant(q1) and _fuse_out_quant(q2) and Co1 % 32 == 0:
            kadds = None
            if adds is not None and FUSE_ADD_FWD:
                # adds[i] = (codes of the other operand, the AddQ's quantizer module): that AddQ is the only consumer of output i and
                # runs in this GEMM's epilogue; its own forward finds the finished codes on the tensor (_fqss_presum, EwQ.forward)
                kadds = tuple(None if a is None else (a[0].idx, a[0].qmin, a[0].qmax, a[1].min_range, a[1].max_range) for a in adds)
            res = K.qpw_fwdq(xq.idx, pair.wc, b1, b2, xq.qmin, xq.qmax, Co1, ACT_NONE, None, (q1.qmin, q1.qmax), (q2.qmin, q2.qmax),
                             adds=kadds)
            z1, z2, q1.idx, q2.idx = res[:4]
            if kadds is not None and len(res) == 5:
                q1.presum, q2.presum = (None if sc is None else _PreSum(sc, a[0].idx, a[1]) for sc, a in zip(res[4], adds))
            q1.carrier = q2.carrier = True
            out1 = _carrier(K.empty_act(tuple(z1.shape), z1.device))
            out2 = _carrier(K.empty_act(tuple(z2.shape), z2.device))
        else:
            z1, z2 = K.qpw_fwd2(xq.idx, pair.wc, b1, b2, xq.qmin, xq.qmax, Co1)
            out1 = _epilogue_fwd(z1, ACT_NONE, None, q1)
            out2 = _epilogue_fwd(z2, ACT_NONE, None, q2)
        ctx.save_for_backward(z1, z2)
        ctx.L, ctx.q, ctx.b, ctx.xq, ctx.pair = (L1, L2), (q1, q2), (b1, b2), xq, pair
        ctx.prods = (None, None)
        if FUSE_EW and sole_ew and all(getattr(q.owner, "_fqss_deferred", False) for q in (q1, q2)):
            # each output has ONE consumer, an element-wise LayerQ (residual AddQ / skip sum): that layer's backward
            # kernel will also run this layer's output-quantizer backward (ops._Producer)
            ctx.prods = (_Producer(z1, ACT_NONE, None, None, q1, L1.b_param, b1), _Producer(z2, ACT_NONE, None, None, q2, L2.b_param, b2))
            q1.prod, q2.prod = ctx.prods
        return out1, out2

    @staticmethod
    def backward(ctx, g1, g2):
        z = ctx.saved_tensors
        g_in = (g1, g2)
        gs, gz, gbias, gmin, gmax = [g1, g2], [], [], [], []
        for i in range(2):
            if gs[i] is None:      # an unused output (the last block's residual): its gradient is zero
                gs[i] = K.empty_act(tuple(z[i].shape), z[i].device).zero_()
            L, q, b = ctx.L[i], ctx.q[i], ctx.b[i]
            if ctx.prods[i] is not None and ctx.prods[i].fused and g_in[i] is not None:
                # the consumer's backward already ran this layer's epilogue backward: the incoming gradient IS gz
                _touch(q.owner.min_range, q.owner.max_range, L.b_param)
                gz.append(gs[i]); gbias.append(None); gmin.append(None); gmax.append(None)
                continue
            gzi, _, g_min, g_max, g_bias = _epilogue_bwd(z[i], gs[i], ACT_NONE, None, None, q, bias_param=L.b_param,
                                                          bias_like=b, C=z[i].shape[1])
            gz.append(gzi); gbias.append(g_bias); gmin.append(g_min); gmax.append(g_max)
        gx = K.qpw_bwd_x2(gz[0], gz[1], ctx.pair.wc) if ctx.needs_input_grad[0] else None
        wq = getattr(DEFER, "wgrad_queue", None)
        if wq is not None:
            wq.push(gz[0], gz[1], ctx.xq.idx, ctx.xq.qmin, ctx.xq.qmax, ctx.pair.gw)
        else:
            K.qpw_bwd_w2(gz[0], gz[1], ctx.xq.idx, ctx.xq.qmin, ctx.xq.qmax, ctx.pair.gw)
        return gx, gbias[0], gbias[1], gmin[0], gmax[0], gmin[1], gmax[1], None, None, None, None, None, None, None, None


FUSE_GNQ_F = os.environ.get("FQSS_FUSE_GNQ_F", "1") != "0"    # GroupNormQ on a float input: the quantizer inside the GroupNorm's own passes


class GroupNormActQ(Function):
    """out = fq(GroupNorm(1, C)(x))  -- GroupNormQ.  With coded input in the quantizing phase the layer
    runs codes -> codes (csrc/fused_q.hip) and saves nothing but the input codes and the statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, qmin, qmax, eps, q, gamma_param, beta_param, xq=None):
        ctx.q, ctx.gp, ctx.bp = q, gamma_param, beta_param
        ctx.coded = xq is not None and q.qmode == Q_QUANT
        ctx.prod = getattr(x, "_fqss_prod", None) if ctx.coded else None
        ctx.hand_in = ctx.hand_out = None
        if ctx.coded:
            h = getattr(x, "_fqss_hand", None)
            ctx.hand_in = h if (h is not None and h.kind == "after") else None     # x comes out of a depthwise layer that takes our apply pass
            q.carrier = FAST and not q.keep_out
            st = getattr(x, "_fqss_stats", None)
            if FUSE_GN_DW and NEXT_IS_DW3 and q.carrier and st is not None and x.shape[-1] <= 4096:
                # the 3-tap depthwise layer behind this GroupNorm runs both as ONE kernel (fqss_gndwq_fwd): buffers now, launch there
                out, q.idx, mean_rstd = K.gnq_fwd_deferred(xq.idx)
                q.defer = dict(xc=xq.idx, qmin_x=xq.qmin, qmax_x=xq.qmax, gamma=gamma, beta=beta, eps=eps, qmin=qmin, qmax=qmax, stats=st,
                               yc=q.idx, mean_rstd=mean_rstd, done=False)
            else:
                out, q.idx, mean_rstd = K.gnq_fwd(xq.idx, xq.qmin, xq.qmax, gamma, beta, eps, qmin, qmax, write_out=not q.carrier, stats=st)
            ctx.save_for_backward(gamma, beta, mean_rstd, xq.idx, xq.qmin, xq.qmax, qmin, qmax)
            if FUSE_GN_BWD_DW and _GN_HAND != "after" and NEXT_IS_DW3 and q.gacc is not None and x.dim() == 3 and x.shape[-1] <= K.DWQ_ROW_MAX:
                # the depthwise layer behind this GroupNorm takes our backward's rows pass (ops._GnHand "before")
                ctx.hand_out = q.hand = _GnHand("before", dict(xc0=xq.idx, qmin0=xq.qmin, qmax0=xq.qmax, gamma=gamma, beta=beta,
                                                                mean_rstd=mean_rstd, gacc=q.gacc))
            return _carrier(out) if q.carrier else out
        ctx.fused_f = FUSE_GNQ_F and q.qmode == Q_QUANT and q.gacc is not None and x.dim() == 3 and x.is_cuda      # (the CPU backend: un-fused)
        if ctx.fused_f:
            # float input, quantizing phase: the quantizer inside the GroupNorm's apply pass (fqss_gnq_fwd_f): y AND its codes from one
            # pass, no pre-quant z (the un-fused chain in the codes-only dataflow: z, a quantizer pass for the codes, a decode pass for
            # the float consumer behind it)
            out, q.idx, mean_rstd = K.gnq_fwd_f(x, gamma, beta, eps, qmin, qmax, want_idx=not q.no_codes)
            q.carrier = False
            ctx.save_for_backward(x, gamma, beta, mean_rstd, qmin, qmax)
            return out
        z, mean_rstd = K.gn_fwd(x, gamma, beta, eps)
        plain = q.qmode == Q_BYPASS
        out = z if plain else _epilogue_fwd(z, ACT_NONE, None, q)
        ctx.save_for_backward(x, gamma, None if plain else z, mean_rstd)
        return out

    @staticmethod
    def backward(ctx, g):
        q = ctx.q
        if ctx.coded:
            gamma, beta, mean_rstd, xc, xmin, xmax, qmin, qmax = ctx.saved_tensors
            gg, gg_direct = _grad_buf(ctx.gp, gamma)
            gb, gb_direct = _grad_buf(ctx.bp, gamma)
            producer = _producer_args(ctx.prod) if ctx.needs_input_grad[0] else None
            if producer is not None:
                ctx.prod.fused = True
            hi, ho = ctx.hand_in, ctx.hand_out
            if hi is not None and producer is None and gg_direct and gb_direct and ctx.needs_input_grad[0]:
                # rows pass here; the apply pass runs inside the depthwise layer's backward, on this gradient as it loads it: what
                # goes back to autograd is the INCOMING gradient, not dL/dx (the edge has that one consumer)
                gx, ws = K.gnq_bwd_rows(xc, xmin, xmax, g, gamma, beta, mean_rstd, qmin, qmax, q.gacc)
                hi.rec = dict(gamma=gamma, beta=beta, mean_rstd=mean_rstd, ws=ws, qmin=qmin, qmax=qmax, ggamma=gg, gbeta=gb)
            elif ho is not None and ho.rec is not None:
                # the depthwise layer behind us took the rows pass on the gradient it produced (row sums + our range partials)
                gx = K.gnq_bwd_apply(xc, xmin, xmax, g, gamma, beta, mean_rstd, qmin, qmax, ho.rec["ws"], gg, gb, producer=producer)
                ho.rec = None
            else:
                gx = K.gnq_bwd(xc, xmin, xmax, g, gamma, beta, mean_rstd, qmin, qmax, q.gacc, gg, gb, producer=producer)
            _, g_min, g_max = _flush_ranges(q, None, None, ACT_NONE)
            return gx, (None if gg_direct else gg), (None if gb_direct else gb), g_min, g_max, None, None, None, None, None
        if ctx.fused_f:
            x, gamma, beta, mean_rstd, qmin, qmax = ctx.saved_tensors
            gg, gg_direct = _grad_buf(ctx.gp, gamma)
            gb, gb_direct = _grad_buf(ctx.bp, gamma)
            gx = K.gnq_bwd_f(g.contiguous(), x, gamma, beta, mean_rstd, qmin, qmax, q.gacc, gg, gb)
            g_min, g_max = _ranges_after(q, q.gacc)
            return gx, (None if gg_direct else gg), (None if gb_direct else gb), g_min, g_max, None, None, None, None, None
        x, gamma, z, mean_rstd = ctx.saved_tensors
        gz, _, g_min, g_max, _ = _plain_or_bwd(z, g, q)
        gg, gg_direct = _grad_buf(ctx.gp, gamma)
        gb, gb_direct = _grad_buf(ctx.bp, gamma)
        gx = K.gn_bwd(gz, x, gamma, mean_rstd, gg, gb)
        return gx, (None if gg_direct else gg), (None if gb_direct else gb), g_min, g_max, None, None, None, None, None


class DwConvQ(Function):
    """out = fq(act(depthwise_conv(x, w) + bias)) on coded input, codes -> codes; the backward
    recomputes the pre-quant value from the input codes (csrc/fused_q.hip)"""

    @staticmethod
    def forward(ctx, x, w, bias, slope, qmin, qmax, L, act, q, xq):
        q.carrier = FAST and not q.keep_out
        d = getattr(x, "_fqss_defer", None)
        if d is not None and not d["done"] and q.carrier and w.shape[-1] == 3 and L.pad == L.dil:
            # the GroupNormQ in front left its launch to this layer: gLN + quantizer + depthwise + PReLU + quantizer as one kernel
            out, q.idx, q.stats = K.gndwq_fwd(d, w, bias, L.dil, L.pad, act, slope, qmin, qmax, FUSE_STATS and NEXT_IS_GROUPNORM)
            d["done"] = True
        else:
            flush_defer(x)
            q.stats = K.new_stats("dwq", x.shape[0], x.shape[1], x.shape[2], x.device) if (FUSE_STATS and NEXT_IS_GROUPNORM) else None
            out, q.idx = K.dwq_fwd(xq.idx, xq.qmin, xq.qmax, w, bias, L.dil, L.pad, act, slope, qmin, qmax, write_out=not q.carrier,
                                   stats=q.stats)
        ctx.save_for_backward(w, bias, slope, xq.idx, xq.qmin, xq.qmax, qmin, qmax)
        ctx.L, ctx.act, ctx.q = L, act, q
        ctx.hand_a = ctx.hand_b = None
        if FUSE_GN_BWD_DW and w.shape[-1] == 3 and x.shape[-1] <= K.DWQ_ROW_MAX and q.gacc is not None:
            h = getattr(x, "_fqss_hand", None)
            ctx.hand_b = h if (h is not None and h.kind == "before") else None
            if NEXT_IS_GROUPNORM and q.qmode == Q_QUANT and _GN_HAND != "before":
                ctx.hand_a = q.hand = _GnHand("after")       # the GroupNormQ behind this layer may leave its apply pass to our backward
        return _carrier(out) if q.carrier else out

    @staticmethod
    def backward(ctx, g):
        w, bias, slope, xc, xmin, xmax, qmin, qmax = ctx.saved_tensors
        L, act, q = ctx.L, ctx.act, ctx.q
        gb, gb_direct = (None, True)
        if bias is not None:
            gb, gb_direct = _grad_buf(L.b_param, bias)
        gw = None
        gwq = getattr(w, "_fqss_gwq", None)
        want_gw = ctx.needs_input_grad[1] or gwq is not None
        if want_gw:
            gw = gwq if gwq is not None else _wgrad_temp(w)
        if xc.shape[-1] <= K.DWQ_ROW_MAX:
            # one launch: gz stays in LDS (csrc/fused_q.hip k_dwq_bwd)
            after = before = None
            if ctx.hand_a is not None and ctx.hand_a.rec is not None:
                after, ctx.hand_a.rec = ctx.hand_a.rec, None      # g is the gradient w.r.t. that GroupNormQ's OUTPUT
            if ctx.hand_b is not None and ctx.needs_input_grad[0]:
                before = dict(ctx.hand_b.fwd)
            gx = K.dwq_bwd(xc, xmin, xmax, w, bias, g, L.dil, L.pad, act, slope, qmin, qmax, q.gacc, gb, gw,
                           want_gx=ctx.needs_input_grad[0], after=after, before=before)
            if before is not None:
                ctx.hand_b.rec = dict(ws=before["ws"])
        else:
            gz = K.dwq_bwd_z(xc, xmin, xmax, w, bias, g, L.dil, L.pad, act, slope, qmin, qmax, q.gacc, gb)
            gx = K.dwconv_bwd_x(gz, w, L.dil, L.pad) if ctx.needs_input_grad[0] else None
            if want_gw:
                K.dwq_bwd_w(gz, xc, xmin, xmax, gw, L.dil, L.pad)
        g_slope, g_min, g_max = _flush_ranges(q, slope, L.slope_param, act)
        if want_gw:
            _wgrad_temp_done(gw, gwq)
            if L.w_param is not None and w is L.w_param:
                L.w_param._fqss_touched = True
            if gwq is not None:
                gw = None
        return gx, gw, (None if gb_direct else gb), g_slope, g_min, g_max, None, None, None, None


class EwQ(Function):
    """out = fq(act(a + sb*b)) on coded a (b: coded | real fp32 | absent), codes -> codes; the backward
    recomputes z from the codes.  AddQ / Sub of the residual block / NlQ in the quantizing phase."""

    @staticmethod
    def forward(ctx, a, b, slope, qmin, qmax, sb, act, q, aq_, bq_, slope_param):
        q.carrier = FAST and not q.keep_out
        bf = None if (b is None or bq_ is not None) else b
        pre = getattr(b, "_fqss_presum", None) if bq_ is not None else None
        if (pre is not None and q.carrier and pre.owner is q.owner and pre.a_idx is aq_.idx and sb == 1.0 and act == ACT_NONE):
            out, q.idx = K.empty_act(tuple(aq_.idx.shape), aq_.idx.device), pre.codes      # evaluated by b's producer (LinearActQPair)
        else:
            out, q.idx = K.ewq_fwd(aq_.idx, aq_.qmin, aq_.qmax, bq_.idx if bq_ else None, bq_.qmin if bq_ else None,
                                   bq_.qmax if bq_ else None, bf, sb, act, slope, qmin, qmax, write_out=not q.carrier)
        ctx.save_for_backward(aq_.idx, aq_.qmin, aq_.qmax, bq_.idx if bq_ else None, bq_.qmin if bq_ else None,
                              bq_.qmax if bq_ else None, bf, slope, qmin, qmax)
        ctx.sb, ctx.act, ctx.q, ctx.sp, ctx.has_b = sb, act, q, slope_param, b is not None
        # operands that are fresh outputs of pointwise convs with no other consumer (tagged by run_conv1d_pair)
        ctx.prod_a = getattr(a, "_fqss_prod", None)
        ctx.prod_b = getattr(b, "_fqss_prod", None) if (b is not None and bq_ is not None and sb == 1.0) else None
        ctx.fork_a = getattr(a, "_fqss_fork", None)      # operand a is one branch of a residual fork (ops._ForkState)
        ctx.link = None
        if (FUSE_ADD_CHAIN and q.chain and ctx.prod_b is not None and ctx.fork_a is None and act == ACT_NONE and aq_.idx.dim() == 3
                and q.gacc is not None):
            prev = getattr(a, "_fqss_chain", None)
            if prev is not None and tuple(prev.ac.shape) != tuple(aq_.idx.shape):
                prev = None
            ctx.link = q.link = _ChainLink(prev, aq_.idx, aq_.qmin, aq_.qmax, bq_.idx, bq_.qmin, bq_.qmax, qmin, qmax, q, ctx.prod_a, ctx.prod_b)
        return _carrier(out) if q.carrier else out

    @staticmethod
    def backward(ctx, g):
        ac, amin, amax, bc, bmin, bmax, bf, slope, qmin, qmax = ctx.saved_tensors
        q = ctx.q
        link = ctx.link
        if link is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            res = _chain_backward(link, g)
            if res is not None:
                ga, gb, a_fused = res
                ctx.prod_b.fused = True
                if a_fused:
                    ctx.prod_a.fused = True
                g_slope, g_min, g_max = _flush_ranges(q, slope, ctx.sp, ctx.act)
                return ga, gb, g_slope, g_min, g_max, None, None, None, None, None, None
        pa = _producer_args(ctx.prod_a) if ctx.needs_input_grad[0] else None
        pb = _producer_args(ctx.prod_b) if (ctx.has_b and ctx.needs_input_grad[1]) else None
        if (pa is not None or pb is not None) and bf is None and ac.dim() == 3:
            # the producers' epilogue backward (output fake-quant STE, range / bias partials) rides in this kernel
            gz, gza, gzb = K.ewq_bwd_p(ac, amin, amax, bc, bmin, bmax, ctx.sb, g, ctx.act, slope, qmin, qmax, q.gacc, ac.shape[1],
                                       prod_a=pa, prod_b=pb)
            if pa is not None:
                ctx.prod_a.fused = True
            if pb is not None:
                ctx.prod_b.fused = True
        else:
            gz, gza, gzb = K.ewq_bwd(ac, amin, amax, bc, bmin, bmax, bf, ctx.sb, g, ctx.act, slope, qmin, qmax, q.gacc), None, None
        g_slope, g_min, g_max = _flush_ranges(q, slope, ctx.sp, ctx.act)
        ga = (gza if gza is not None else gz) if ctx.needs_input_grad[0] else None
        if ga is not None and ctx.fork_a is not None:
            ctx.fork_a[0].leave(ga, ctx.fork_a[1])      # the conv on the fork's other branch may add it in its dgrad epilogue
        gb = None
        if ctx.has_b and ctx.needs_input_grad[1]:
            if gzb is not None:
                gb = gzb
            else:
                gb = gz if ctx.sb == 1.0 else K.axpby(gz, gz, 0.0, sa=float(ctx.sb))
        return ga, gb, g_slope, g_min, g_max, None, None, None, None, None, None


def _chain_backward(link, g):
    """EwQ.backward of a chained AddQ: -> (ga, gb, producer-of-a fused) or None (not part of a chain launch: the per-level kernel).
    The TOP level of a chain (the first whose backward runs) launches fqss_add_chain_bwd for itself and every level below it and leaves
    the lower levels' results on their links; their own nodes -- which autograd still runs, in order, each on the placeholder the
    level above returned -- hand them on."""
    if link.pre is not None:
        (ga, gb, a_fused), link.pre = link.pre, None
        if g.data_ptr() != link.dummy.data_ptr():
            raise RuntimeError("fqss_amd.ops: a chained AddQ received a gradient that is not the one its chain handed down -- its "
                               "first operand has another consumer (QCtx.chain was declared wrongly)")
        return ga, gb, a_fused
    levels, cur = [], link
    while cur is not None and len(levels) < K.ADD_CHAIN_MAX:
        pb = _producer_args(cur.prod_b)
        if pb is None or pb[1] != ACT_NONE:
            break
        levels.append((cur, pb))
        cur = cur.prev
    B, C, M = link.ac.shape
    if len(levels) < 2 or not K.add_chain_ok(B * C, M, C, len(levels)):
        return None
    bottom = levels[-1][0]
    pa = _producer_args(bottom.prod_a) if bottom.prod_a is not None else None
    if pa is not None and pa[1] != ACT_NONE:
        pa = None
    levels.reverse()       # forward order: levels[0] = the bottom
    outs, ga_bottom = K.add_chain_bwd([dict(ac=l.ac, amin=l.amin, amax=l.amax, bc=l.bc, bmin=l.bmin, bmax=l.bmax, qmin=l.qmin, qmax=l.qmax,
                                            gacc=l.q.gacc, prod_b=pb) for l, pb in levels], g, prod_a=pa)
    dummy = K.empty_act((B, C, M), link.ac.device)      # what travels along the chain's autograd edges: never read
    n = len(levels)
    for i, ((l, _), o) in enumerate(zip(levels, outs)):
        below = (ga_bottom, o, pa is not None) if i == 0 else (dummy, o, False)
        if i == n - 1:
            return below          # this call's own level (the top)
        l.pre, l.dummy = below, dummy
    return None


def ew_layer(x1, x2, sb, act, slope, q):
    """dispatch an element-wise quantized layer: coded kernel when the first operand carries codes and the
    layer quantizes, the fp32 kernels otherwise (decoding carriers first)"""
    aq_ = codes_of(x1)
    if aq_ is not None and q.qmode == Q_QUANT and q.gacc is not None:
        bq_ = codes_of(x2) if x2 is not None else None
        b = x2 if (x2 is None or bq_ is not None) else real(x2)
        return EwQ.apply(x1, b, slope, q.qmin, q.qmax, sb, act, q, aq_, bq_, slope)
    if x2 is None:
        return NlActQ.apply(real(x1), slope, q.qmin, q.qmax, act, q, slope)
    return AddActQ.apply(real(x1), real(x2), q.qmin, q.qmax, sb, q)


class AddActQ(Function):
    """out = fq(a + sign*b)  -- AddQ (sign=+1), ResidualErrorBlock's Y - Y_q (sign=-1)"""

    @staticmethod
    def forward(ctx, a, b, qmin, qmax, sign, q):
        z = K.axpby(a, b, sign)
        plain = q.qmode == Q_BYPASS
        out = z if plain else _epilogue_fwd(z, ACT_NONE, None, q)
        ctx.save_for_backward(None if plain else z)
        ctx.q, ctx.sign = q, sign
        return out

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        gz, _, g_min, g_max, _ = _plain_or_bwd(z, g, ctx.q)
        ga = gz if ctx.needs_input_grad[0] else None
        gb = None
        if ctx.needs_input_grad[1]:
            gb = gz if ctx.sign == 1.0 else K.axpby(gz, gz, 0.0, sa=float(ctx.sign))
        return ga, gb, g_min, g_max, None, None


class MulActQ(Function):
    """out = fq(mask[B,S,C,M] * feat[B,1,C,M])  -- MulQ of ConvTasNetQ.forward"""

    @staticmethod
    def forward(ctx, mask, feat, qmin, qmax, q):
        z = K.mul_bcast_fwd(mask, feat)
        plain = q.qmode == Q_BYPASS
        out = z if plain else _epilogue_fwd(z, ACT_NONE, None, q)
        ctx.save_for_backward(mask, feat, None if plain else z)
        ctx.q = q
        return out

    @staticmethod
    def backward(ctx, g):
        mask, feat, z = ctx.saved_tensors
        gz, _, g_min, g_max, _ = _plain_or_bwd(z, g, ctx.q)
        gmask, gfeat = K.mul_bcast_bwd(gz, mask, feat)
        return gmask, gfeat, g_min, g_max, None


class MulQCoded(Function):
    """out = fq(mask[B,S,C,M] * feat[B,1,C,M]) on coded operands, codes -> codes (MulQ of ConvTasNetQ.forward in the quantizing
    phase): one launch instead of decode x 2 + product + quantizer, nothing saved but the operands' codes; the backward
    recomputes the product (csrc/fused_q.hip fqss_mulq_fwd / fqss_mulq_bwd)."""

    @staticmethod
    def forward(ctx, mask, feat, qmin, qmax, q, mq, fq_):
        B, S, C, M = mask.shape
        q.carrier = FAST and not q.keep_out
        fidx = fq_.idx.reshape(B, C, M)
        out, q.idx = K.mulq_fwd(mq.idx, mq.qmin, mq.qmax, fidx, fq_.qmin, fq_.qmax, qmin, qmax, write_out=not q.carrier)
        ctx.save_for_backward(mq.idx, mq.qmin, mq.qmax, fidx, fq_.qmin, fq_.qmax, qmin, qmax)
        ctx.q, ctx.feat_shape = q, feat.shape
        ctx.prod = getattr(mask, "_fqss_prod", None)     # the mask is the fresh output of a pointwise conv with no other consumer
        return _carrier(out) if q.carrier else out

    @staticmethod
    def backward(ctx, g):
        mc, mmin, mmax, fc, fmin, fmax, qmin, qmax = ctx.saved_tensors
        q = ctx.q
        pa = _producer_args(ctx.prod) if ctx.needs_input_grad[0] else None
        gmask, gfeat = K.mulq_bwd(mc, mmin, mmax, fc, fmin, fmax, g, qmin, qmax, q.gacc, want_gfeat=ctx.needs_input_grad[1], prod=pa)
        if pa is not None:
            ctx.prod.fused = True
        _, g_min, g_max = _flush_ranges(q, None, None, ACT_NONE)
        return (gmask if ctx.needs_input_grad[0] else None, gfeat.reshape(ctx.feat_shape) if gfeat is not None else None,
                g_min, g_max, None, None, None)


class NlActQ(Function):
    """out = fq(act(x))  -- NlQ, and the bare GradientActivationFakeQuantize module (act = NONE)"""

    @staticmethod
    def forward(ctx, x, slope, qmin, qmax, act, q, slope_param):
        out = _epilogue_fwd(x, act, slope, q)
        ctx.save_for_backward(x, slope)
        ctx.act, ctx.q, ctx.sp = act, q, slope_param
        return out

    @staticmethod
    def backward(ctx, g):
        x, slope = ctx.saved_tensors
        gz, g_slope, g_min, g_max, _ = _epilogue_bwd(x, g, ctx.act, slope, ctx.sp, ctx.q)
        return gz, g_slope, g_min, g_max, None, None, None


class WeightFq(Function):
    """w_q = per-channel symmetric fake-quant of w  -- GradientWeightFakeQuantize (quantizing call)"""

    @staticmethod
    def forward(ctx, w, qmin, qmax, axis, owner, w_param):
        wq = K.wq_fwd(w, axis, qmin, qmax)
        ctx.save_for_backward(w, qmin, qmax)
        ctx.axis, ctx.owner, ctx.w_param = axis, owner, w_param
        return wq

    @staticmethod
    def backward(ctx, g):
        w, qmin, qmax = ctx.saved_tensors
        gw, d0 = _grad_buf(ctx.w_param, w)
        gmn, d1 = _grad_buf(ctx.owner.min_range if ctx.owner is not None else None, qmin)
        gmx, d2 = _grad_buf(ctx.owner.max_range if ctx.owner is not None else None, qmax)
        K.wq_bwd(w, g, ctx.axis, qmin, qmax, out=(gw, gmn, gmx))
        return (None if d0 else gw), (None if d1 else gmn), (None if d2 else gmx), None, None, None


class Combine2(Function):
    """y = x0 + x1 * 2^-8  -- process.postprocess with n_combiner = 2"""

    @staticmethod
    def forward(ctx, x0, x1):
        return K.axpby(x0, x1, 0.00390625)

    @staticmethod
    def backward(ctx, g):
        g1 = K.axpby(g, g, 0.0, sa=0.00390625) if ctx.needs_input_grad[1] else None
        return g, g1


def splitter2(x):
    with torch.no_grad():
        return K.splitter2(x)


class _ForkState:
    """lets the dgrad q-GEMM of ONE branch of a two-way fork add the gradient of the OTHER branch in its epilogue (FUSE_FORK): the other
    branch's backward (an element-wise LayerQ, created later in the forward => run earlier in the backward) leaves its gradient in
    `other` (tagged with its branch index, ACCUMULATED if that branch has several such consumers); the conv's LinearActQ.backward
    consumes it only if it sits on the opposite branch and records which branch it was -- and WHAT it added (`taken`, a snapshot:
    a further element-wise consumer of that branch whose backward runs AFTER the conv's keeps accumulating into `other`, which must not
    change what Fork2.backward subtracts); Fork2.backward then passes the fused branch's gradient through -- after checking that what
    was added really is the other branch's complete gradient."""
    __slots__ = ("other", "other_branch", "fused_branch", "n_other", "taken", "n_taken")

    def __init__(self):
        self.other, self.other_branch, self.fused_branch, self.n_other, self.taken, self.n_taken = None, None, None, 0, None, 0

    def leave(self, g, branch):
        """an element-wise consumer on `branch` hands over its input gradient"""
        if self.other is None:
            self.other, self.other_branch, self.n_other = g, branch, 1
        elif self.other_branch == branch:
            self.other, self.n_other = K.axpby(self.other, g, 1.0), self.n_other + 1
        else:                       # consumers on BOTH branches left gradients: nothing to fuse
            self.other, self.other_branch, self.n_other = None, -1, 0

    def take(self, branch, shape):
        """the conv on `branch` asks for the opposite branch's gradient to add in its dgrad epilogue (None: sum the usual way)"""
        if self.other is None or self.other_branch in (None, -1, branch) or self.fused_branch is not None \
                or tuple(self.other.shape) != tuple(shape):
            return None
        self.fused_branch, self.taken, self.n_taken = branch, self.other, self.n_other
        return self.other


FUSE_FORK = os.environ.get("FQSS_FUSE_FORK", "1") != "0"


class Fork2(Function):
    """identity with two consumers: the two incoming gradients are summed by the padded HIP axpby
    (autograd's own accumulation would densify the row stride and push consumers onto the scalar path)"""

    @staticmethod
    def forward(ctx, x, fk):
        ctx.fk = fk
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        fk = ctx.fk
        fb, other, n_other, n_left = fk.fused_branch, fk.taken, fk.n_taken, fk.n_other
        fk.fused_branch, fk.other, fk.other_branch, fk.n_other, fk.taken, fk.n_taken = None, None, None, 0, None, 0
        gs = (g1, g2)
        if fb is None:
            if g1 is None:
                return g2, None
            if g2 is None:
                return g1, None
            return K.axpby(g1, g2, 1.0), None
        gf, go = gs[fb], gs[1 - fb]              # gf already holds (its own gradient + `other`)
        if go is None or (n_other == 1 and n_left == 1 and go.data_ptr() == other.data_ptr()):
            return gf, None                      # `other` WAS the opposite branch's complete gradient (fqss_qpw_bwd_x_add)
        # the opposite branch had further consumers -- before or AFTER the conv's backward took its snapshot -- and autograd summed all
        # of them into `go`: add what the epilogue missed (go - the snapshot that was added)
        return K.axpby(K.axpby(gf, go, 1.0), other, -1.0), None


def fork2(x):
    if torch.is_grad_enabled() and x.requires_grad:
        fk = _ForkState()
        a, b = Fork2.apply(x, fk)
        if FUSE_FORK:
            a._fqss_fork, b._fqss_fork = (fk, 0), (fk, 1)
        c = codes_of(x)
        if c is not None:
            a._fqss_q = b._fqss_q = c
            a._fqss_carrier = b._fqss_carrier = is_carrier(x)
        rq = getattr(x, "_fqss_rowq", None)       # codes of a dual-path row tensor (read by ops_dp.row_linear only)
        if rq is not None:
            a._fqss_rowq = b._fqss_rowq = rq
        return a, b
    return x, x


class KDLoss(Function):
    """loss = -10 log10((1-l)*task + l*kd + eps) with SDR-weighted KD and 2-speaker PIT
    (mysystem.py:124-151); forward and backward come out of ONE kernel sequence."""

    @staticmethod
    def forward(ctx, est, fest, tgt, kd_lambda):
        out, w, sisdr, gest = K.kd_loss(est, fest, tgt, kd_lambda, want_grad=True)
        ctx.save_for_backward(gest)
        ctx.mark_non_differentiable(w, sisdr)
        return out[0], out[1].detach(), w, sisdr

    @staticmethod
    def backward(ctx, g_loss, g_kd, g_w, g_s):
        (gest,) = ctx.saved_tensors
        return gest * g_loss, None, None, None


class HdKDLoss(Function):
    """loss of the htdemucs solver (solver.py:333-366): (1 - lambda) L1(est, src) + lambda w L1(est, fest) per source, source-weighted"""

    @staticmethod
    def forward(ctx, est, fest, src, weights, kd_lambda):
        loss, task, kd, w, g = K.hd_kd_loss(est, fest, src, weights, kd_lambda, want_grad=True)
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(task, kd, w)
        return loss.reshape(()), task, kd, w

    @staticmethod
    def backward(ctx, gl, *_):
        (g,) = ctx.saved_tensors
        return g * gl, None, None, None, None
