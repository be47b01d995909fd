"""Learned-range fake-quantizers of FQSS on MI355X.

Same public names, constructor signatures, parameter names/shapes (=> same state_dict keys) and
host-side state (`n_iter`, `observer_mode`, `max_observations`, `alpha`) as the reference's
quantization/qat/qat_quant.py; the arithmetic runs in the HIP kernels of csrc/fq.hip:

  linear_quantize                    qat_quant.py:125-147  -> fqss_actq_fwd/bwd, fqss_wq_fwd/bwd
  GradientActivationFakeQuantize     qat_quant.py:206-242  -> observer: fqss_actq_fwd(OBSERVE)+fqss_observer_ema
  GradientWeightFakeQuantize         qat_quant.py:350-381  -> fqss_wq_observe / fqss_wq_fwd / fqss_wq_bwd

Deliberate differences (documented in DESIGN.md):
  * no host syncs: the reference does `.item()` + `assert max >= min` on every training forward
    (qat_quant.py:235-238); here nothing reads device memory from the host.
  * 8-bit only (every shipped config); other widths raise NotImplementedError.
"""
import torch
import torch.nn as nn

from ... import kernels as K
from ... import ops


def ops_dp_qrow():
    from ... import ops_dp
    return ops_dp.QROW


def _check_bits(n_bits):
    if n_bits != 8:
        raise NotImplementedError("fqss_amd kernels implement the 8-bit quantizers of the shipped FQSS configs")


# ---- public STE helpers (qat_quant.py:88-107 of the reference): `(f(x) - x).detach() + x` -> value f(x), gradient of x ------------
class _Ste(torch.autograd.Function):
    """y = f(x) (one HIP map) with the gradient of `gscale * x`: the straight-through estimators users build new quantizers from"""

    @staticmethod
    def forward(ctx, x, kind, p, p2, gscale):
        ctx.gscale = float(gscale)
        return K.unary2_fwd(ops.real(x), kind, p, p2) if kind >= 0 else ops.real(x).clone()

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.gscale == 1.0 else K.axpby(g, g, 0.0, sa=ctx.gscale)), None, None, None, None


def round_ste(x):
    """(torch.round(x) - x).detach() + x  (qat_quant.py:88-89)"""
    return _Ste.apply(x, K.UNARY_ROUND, 0.0, 0.0, 1.0)


def floor_ste(x):
    """(torch.floor(x) - x).detach() + x  (:92-93)"""
    return _Ste.apply(x, K.UNARY_FLOOR, 0.0, 0.0, 1.0)


def grad_sign(x, scale=1.0):
    """value sign(x), gradient scale * g  (:96-98)"""
    return _Ste.apply(x, K.UNARY_SIGN, 0.0, 0.0, scale)


def grad_scale(x, scale):
    """value x, gradient scale * g  (:101-103)"""
    return _Ste.apply(x, -1, 0.0, 0.0, scale)


def clip_ste(x, min_val=-1.0, max_val=1.0):
    """value clip(x, min_val, max_val), identity gradient  (:106-107)"""
    return _Ste.apply(x, K.UNARY_CLIP, min_val, max_val, 1.0)


def linear_quantize(x, min_range, max_range, n_bits, sign=True, sym=False, scale_grad=False):
    """Functional form (differentiable w.r.t. x, min_range, max_range) on device tensors."""
    _check_bits(n_bits)
    if scale_grad:
        raise NotImplementedError("scale_grad=True is not used by any FQSS config")
    if sym:
        shape = tuple(min_range.shape)
        axis = next((i for i, s in enumerate(shape) if s != 1), 0)
        return ops.WeightFq.apply(x, min_range, max_range, axis, None, None)
    q = ops.QCtx(ops.Q_QUANT, min_range, max_range, None, None, None)
    return ops.NlActQ.apply(x, None, min_range, max_range, ops.ACT_NONE, q, None)


class GradientActivationFakeQuantize(nn.Module):
    """Per-tensor asymmetric activation quantizer with a 50-call EMA observer, then learned ranges."""

    def __init__(self, gradient_based, n_bits=8, sym=False, scale_grad=False):
        super().__init__()
        _check_bits(n_bits)
        if sym or scale_grad:
            raise NotImplementedError("sym/scale_grad activation quantizers are not reachable from the FQSS configs")
        self.n_bits = n_bits
        self.sym = sym
        self.min_range = nn.Parameter(torch.tensor([-0.5]), requires_grad=gradient_based)
        self.max_range = nn.Parameter(torch.tensor([0.5]), requires_grad=gradient_based)
        self.max_observations = 50
        self.observer_mode = True
        self.alpha = 0.9
        self.n_iter = 0
        self.sign = True
        self.scale_grad = scale_grad
        # device scratch: not part of the state_dict (persistent=False keeps the reference's key set)
        self.register_buffer("_obs_ws", torch.tensor([-1, 0], dtype=torch.int32), persistent=False)
        self.register_buffer("_gacc", torch.zeros(K.GACC_DOUBLES, dtype=torch.float64), persistent=False)

    def enable_observer(self, observer_mode):
        self.observer_mode = observer_mode

    # -- fused entry points used by the LayerQ modules ---------------------------------------
    def next_mode(self):
        """advance the host-side observer counter exactly like qat_quant.py:228-229"""
        if self.observer_mode and self.n_iter < self.max_observations:
            self.n_iter += 1
            return ops.Q_OBSERVE
        return ops.Q_QUANT

    def qctx(self):
        return ops.QCtx(self.next_mode(), self.min_range, self.max_range, self._obs_ws, self._gacc, self)

    def after_forward(self, q):
        if q.qmode == ops.Q_OBSERVE:
            K.observer_ema(self.min_range.data, self.max_range.data, self._obs_ws, self.alpha)

    def forward(self, x):
        q = self.qctx()
        y = ops.NlActQ.apply(ops.real(x), None, self.min_range, self.max_range, ops.ACT_NONE, q, None)
        self.after_forward(q)
        return ops.tag_codes(y, q)


class GradientWeightFakeQuantize(nn.Module):
    """Per-output-channel symmetric weight quantizer; the first call only records amax/amin."""

    def __init__(self, gradient_based, weight_shape, n_bits=8, sym=True, ch_out_idx=0, scale_grad=False):
        super().__init__()
        _check_bits(n_bits)
        if not sym or scale_grad:
            raise NotImplementedError("asymmetric/scale_grad weight quantizers are not reachable from the FQSS configs")
        self.n_bits = n_bits
        self.sym = sym
        self.axis = ch_out_idx
        init_shape = [1] * len(weight_shape)
        init_shape[ch_out_idx] = weight_shape[ch_out_idx]
        self.min_range = nn.Parameter(-0.5 * torch.ones(init_shape), requires_grad=gradient_based)
        self.max_range = nn.Parameter(0.5 * torch.ones(init_shape), requires_grad=gradient_based)
        self.observer_mode = True
        self.sign = True
        self.scale_grad = scale_grad

    def enable_observer(self, observer_mode):
        self.observer_mode = observer_mode

    def forward(self, x, w_param=None):
        if self.observer_mode:
            K.wq_observe(x.detach(), self.axis, self.min_range.data, self.max_range.data)
            self.observer_mode = False
            return x
        pre = getattr(x, "_fqss_wq", None)
        if pre is not None and ops.DEFER is not None:
            return pre        # already fake-quantized this step by fqss_wq_multi_fwd (runtime.QuantTables)
        wq = ops.WeightFq.apply(x, self.min_range, self.max_range, self.axis, self, w_param if w_param is not None else x)
        if self.axis == 0 and x.dim() == 3 and x.shape[2] == 1 and K.q_eligible(x.shape[1], x.shape[0]):
            # pointwise-conv weight: also hand its int8 codes to the bf16-MFMA q-GEMMs (csrc/qgemm.hip)
            wq._fqss_wcodes = K.wq_codes(x.detach(), self.min_range.detach(), self.max_range.detach())
        elif self.axis == 0 and x.dim() == 2 and K.qrow_eligible(x.shape[1]) and ops_dp_qrow():
            # row-major linear weight (LinearQ, attention projections, LSTM input projection): codes for csrc/qrow.hip
            wq._fqss_wcodes = K.wq_codes(x.detach(), self.min_range.detach(), self.max_range.detach())
        return wq


class _BypassQuantizer(nn.Identity):
    """act_quant / weight_quant = False (the reference installs nn.Identity, qat_layers.py:56-57)"""

    def qctx(self):
        return ops.BYPASS

    def after_forward(self, q):
        pass


def get_activation_quantizer(gradient_based=True, nl=False, n_bits=8):
    if nl:
        raise NotImplementedError("mu-law (inout_nl_quant) quantizer: unreachable from the FQSS configs (load_model.py:61)")
    return GradientActivationFakeQuantize(gradient_based, n_bits=n_bits)


def get_weight_quantizer(gradient_based=True, weight_shape=(1, 1, 1), n_bits=8, ch_out_idx=0):
    return GradientWeightFakeQuantize(gradient_based, weight_shape, n_bits=n_bits, ch_out_idx=ch_out_idx)


def get_dym_activation_quantizer(n_bits=8, factor=0.99):
    raise NotImplementedError("dynamic activation quantizer: no call site in the FQSS training path")


# ---------------------------------------------------------------------------------------------------------------------------
# true-integer export: the affine (scale, zero-point) wrappers of the reference, qat_quant.py:15-72
# ---------------------------------------------------------------------------------------------------------------------------
class TorchWeightFakeQuantize(nn.Module):
    """per-channel symmetric weight quantizer in torch's affine form (qat_quant.py:15-37): scale = max(|min|, |max|) / 2^(n-1),
    zero point 0; `integer(w)` returns the int8 codes a true-integer kernel consumes"""

    def __init__(self, quantizer):
        super().__init__()
        mn, mx = quantizer.min_range.detach(), quantizer.max_range.detach()
        max_abs = torch.maximum(torch.abs(mn), torch.abs(mx))
        scales = max_abs / (2 ** (quantizer.n_bits - int(quantizer.sign)))
        self.scales = scales.flatten()
        self.zero_points = torch.zeros_like(self.scales)
        self.axis, self.sign, self.n_bits = quantizer.axis, quantizer.sign, quantizer.n_bits

    def _lim(self):
        return (-2 ** (self.n_bits - 1), 2 ** (self.n_bits - 1) - 1) if self.sign else (0, 2 ** self.n_bits - 1)

    def forward(self, x, w_param=None):
        return K.fq_affine(x, self.scales.to(x.device), self.zero_points.to(x.device).int(), self.axis, *self._lim())

    def integer(self, x):
        _, codes = K.fq_affine(x, self.scales.to(x.device), self.zero_points.to(x.device).int(), self.axis, *self._lim(), want_codes=True)
        return codes.to(torch.int8 if self.sign else torch.uint8)


class TorchActivationFakeQuantize(nn.Module):
    """per-tensor asymmetric activation quantizer in torch's affine form (qat_quant.py:40-56), quirks included: the zero point is
    |round(min / scale)| whatever the sign of min"""

    def __init__(self, quantizer):
        super().__init__()
        mn, mx = quantizer.min_range.detach().cpu(), quantizer.max_range.detach().cpu()
        self.scale = float((mx - mn) / (2 ** quantizer.n_bits - 1))
        zp = int(torch.round(mn / self.scale))
        self.zero_point = -zp if mn < 0 else zp
        self.n_bits = quantizer.n_bits

    def _args(self, dev):
        return torch.tensor([self.scale], device=dev), torch.tensor([self.zero_point], device=dev, dtype=torch.int32)

    def forward(self, x):
        x = ops.real(x)
        return K.fq_affine(x, *self._args(x.device), 0, 0, 2 ** self.n_bits - 1)

    def integer(self, x):
        x = ops.real(x)
        return K.fq_affine(x, *self._args(x.device), 0, 0, 2 ** self.n_bits - 1, want_codes=True)[1].to(torch.uint8)

    # the LayerQ forwards drive their quantizer through qctx(): an exported model runs un-fused, layer by layer
    def qctx(self):
        raise NotImplementedError("TorchActivationFakeQuantize: exported models evaluate through `forward`, not the fused training kernels")


class TorchDymActivationFakeQuantize(nn.Module):
    """dynamic per-call activation quantizer in torch's affine form (qat_quant.py:56-72): the range is factor * (min, max) of the
    tensor at hand, found by the device min/max reduction; scale and zero point are host scalars, as in the reference"""

    def __init__(self, quantizer):
        super().__init__()
        self.n_bits = quantizer.n_bits
        self.factor = getattr(quantizer, "factor", 1.0)

    def forward(self, x):
        x = ops.real(x)
        ws = torch.tensor([-1, 0], dtype=torch.int32, device=x.device)
        K.minmax(x, ws)
        mn, mx = torch.zeros(1, device=x.device), torch.zeros(1, device=x.device)
        K.observer_ema(mn, mx, ws, 0.0)           # alpha = 0: takes the observed pair
        mn, mx = self.factor * float(mn), self.factor * float(mx)
        scale = float((mx - mn) / (2 ** self.n_bits - 1))
        zp = int(round(mn / scale))               # python rounds half to even like torch.round
        zp = -zp if mn < 0 else zp
        return K.fq_affine(x, torch.tensor([scale], device=x.device), torch.tensor([zp], device=x.device, dtype=torch.int32), 0, 0,
                           2 ** self.n_bits - 1)


def export_integer_state(model):
    """{quantizer path: affine parameters} + {weight path: int8 codes} of a trained model: what a true-integer (int8 MFMA) deployment
    loads.  Weights are located through the LayerQ that owns the quantizer (its float submodule's `.weight`)."""
    out = {}
    for name, m in model.named_modules():
        if isinstance(m, GradientActivationFakeQuantize):
            t = TorchActivationFakeQuantize(m)
            out[name] = {"scale": t.scale, "zero_point": t.zero_point, "quant_min": 0, "quant_max": 2 ** t.n_bits - 1}
        elif isinstance(m, GradientWeightFakeQuantize):
            t = TorchWeightFakeQuantize(m)
            out[name] = {"scales": t.scales.cpu(), "zero_points": t.zero_points.cpu(), "axis": t.axis, "quant_min": t._lim()[0], "quant_max": t._lim()[1]}
    return out
