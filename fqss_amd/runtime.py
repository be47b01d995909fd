"""Step runtime: flat parameter arena, fused clip+Adam, and the KD training step.

MI355X-first choices:
  * ONE flat fp32 buffer each for parameters, gradients and the two Adam moments: kernels accumulate
    parameter gradients straight into the flat gradient buffer (no per-parameter AccumulateGrad
    kernels), the optimizer is two launches over the whole model, and data-parallel training
    all-reduces that one buffer over RCCL/xGMI.
  * nothing on the step reads device memory from the host (the reference syncs ~400x per forward,
    qat_quant.py:235-238), so the whole step can be captured in a hipGraph.
"""
import os

import torch

from . import _lib
from . import kernels as K
from . import ops
from . import ops_dp


class ParamArena:
    def __init__(self, params, align=64):
        params = [p for p in params if p.requires_grad]
        assert params, "no trainable parameters"
        dev = params[0].device
        assert dev.type == ("cpu" if _lib.BACKEND == "cpu" else "cuda"), "ParamArena: parameters must live where the selected backend computes"
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.params, self.offsets, self.numel = params, offs, n
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.sumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.step_t = torch.zeros(1, device=dev, dtype=torch.int32)
        self.gnorm = torch.zeros(1, device=dev)
        # torch.optim.Adam counts steps per parameter, from the first step the parameter has a gradient
        self.t0 = torch.full((n,), 2 ** 31 - 1, device=dev, dtype=torch.int32)
        self._inactive = list(zip(params, offs))
        self._host_step = 0
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = self.flat_p[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                p._fqss_direct = True

    def zero_grad(self):
        self.flat_g.zero_()

    def _activate_touched(self):
        if self._inactive:
            still = []
            for p, o in self._inactive:
                if getattr(p, "_fqss_touched", False):
                    self.t0[o:o + p.numel()] = self._host_step
                else:
                    still.append((p, o))
            self._inactive = still
        self._host_step += 1

    def clip_adam_step(self, lr, max_norm=5.0, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8, activate=True):
        if activate:
            self._activate_touched()
        self.sumsq.zero_()
        K.sumsq(self.flat_g, self.sumsq)
        K.adam_clip(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.sumsq, self.step_t, self.gnorm,
                    max_norm, grad_scale, lr, betas[0], betas[1], eps, t0=self.t0)


TEACHER_STREAM = os.environ.get("FQSS_TEACHER_STREAM", "1") != "0"
TEACHER_PRIO = int(os.environ.get("FQSS_TEACHER_PRIO", "0"))        # experiment knob: priority of the teacher stream (0 = default / lowest)


def _teacher_stream():
    return torch.cuda.Stream(priority=TEACHER_PRIO)


class TeacherRunner:
    """Frozen float ConvTasNet teacher as a fused inference chain (csrc/teacher.hip): 3 kernels per TCN
    block, weights pre-split once into exact bf16 planes.  Falls back to the module forward for any other
    teacher structure."""

    def __init__(self, fmodel):
        import torch.nn as nn
        from .quantization.qat.models.convtasnetq import ConvBlock, ConvTasNetQ
        self.fmodel = fmodel
        self.ok = False
        m = fmodel
        try:
            ok = isinstance(m, ConvTasNetQ) and type(m.encoder) is nn.Conv1d and type(m.decoder) is nn.ConvTranspose1d
            mk = m.masker
            ok = ok and type(mk.bottleneck[0]) is nn.GroupNorm and type(mk.bottleneck[1]) is nn.Conv1d
            ok = ok and type(mk.mask_net[0]) is nn.PReLU and type(mk.mask_net[1]) is nn.Conv1d and type(mk.mask_net[2]) is nn.ReLU
            for blk in mk.TCN:
                sb = blk.shared_block
                ok = ok and isinstance(blk, ConvBlock) and [type(t) for t in sb] == [nn.Conv1d, nn.PReLU, nn.GroupNorm, nn.Conv1d,
                                                                                   nn.PReLU, nn.GroupNorm]
                ok = ok and type(blk.res_conv) is nn.Conv1d and type(blk.skip_conv) is nn.Conv1d and sb[3].kernel_size[0] <= 8
                ok = ok and sb[0].in_channels % 8 == 0 and sb[0].out_channels % 8 == 0
            ok = ok and m.n_splitter == 1 and m.n_combiner == 1 and m.encoder.bias is None and m.decoder.bias is None
            ok = ok and m.decoder.out_channels == 1 and m.encoder.in_channels == 1 and m.encoder.kernel_size[0] in (16, 32)
            self.ok = bool(ok)
        except AttributeError:
            self.ok = False
        self._planes = None

    def _prepare(self):
        mk = self.fmodel.masker
        P = lambda conv: K.split3_planes(conv.weight.detach().reshape(conv.out_channels, conv.in_channels))
        blocks = []
        for blk in mk.TCN:
            sb = blk.shared_block
            rs_w = torch.cat([blk.res_conv.weight.detach(), blk.skip_conv.weight.detach()], 0)
            rs_b = torch.cat([blk.res_conv.bias.detach(), blk.skip_conv.bias.detach()], 0).contiguous()
            blocks.append((P(sb[0]), K.split3_planes(rs_w.reshape(rs_w.shape[0], rs_w.shape[1])), rs_b))
        self._planes = dict(bn=P(mk.bottleneck[1]), mask=P(mk.mask_net[1]), blocks=blocks)

    @torch.no_grad()
    def __call__(self, x):
        if not self.ok or (x.shape[-1] - self.fmodel.encoder.kernel_size[0]) // self.fmodel.encoder.stride[0] + 1 > 12 * 1024:
            return self.fmodel(x)     # (the fused depthwise stage keeps a whole row of frames in LDS)
        if self._planes is None:
            self._prepare()
        m, mk, pl = self.fmodel, self.fmodel.masker, self._planes
        if x.dim() == 2:
            x = x.unsqueeze(1)
        B = x.shape[0]
        stride = m.encoder.stride[0]
        feats = K.frames_conv_fwd(x, m.encoder.weight, stride)                       # [B, F, M]
        nb = len(mk.TCN)
        st = K.tstat_buffer(1 + 2 * nb, B, x.device)                                 # GroupNorm (sum, sum^2) partials per sample
        K.tstats(feats, st[0])
        gn0, bn = mk.bottleneck[0], mk.bottleneck[1]
        h = K.tgemm(pl["bn"], feats, bn.bias, pro=1, pro_stats=st[0], pro_gamma=gn0.weight, pro_beta=gn0.bias, pro_eps=gn0.eps)
        acc = None
        nfeat = h.shape[1]
        for i, blk in enumerate(mk.TCN):
            sb = blk.shared_block
            p_c1, p_rs, b_rs = pl["blocks"][i]
            y1 = K.tgemm(p_c1, h, sb[0].bias, act=K.ACT_PRELU, slope=sb[1].weight, stats_out=st[1 + 2 * i])
            y3 = K.tdw(y1, st[1 + 2 * i], sb[2].weight, sb[2].bias, sb[2].eps, sb[3].weight, sb[3].bias, sb[4].weight,
                       st[2 + 2 * i], sb[3].dilation[0], sb[3].padding[0])
            h, acc = K.tgemm(p_rs, y3, b_rs, pro=1, pro_stats=st[2 + 2 * i], pro_gamma=sb[5].weight, pro_beta=sb[5].bias,
                             pro_eps=sb[5].eps, M1=nfeat, r1=h, r2=acc)
        mask = K.tgemm(pl["mask"], acc, mk.mask_net[1].bias, act=K.ACT_RELU, pro=2, pro_slope=mk.mask_net[0].weight)
        F_ = feats.shape[1]
        if K.ola_convtr_ok(m.decoder.weight, stride) and os.environ.get("FQSS_FUSE_TMUL", "1") != "0":
            # the masking product is formed inside the decoder kernel (no [B, S, F, M] intermediate): same values, same sums
            dec = K.ola_convtr_mul_fwd(mask.reshape(B, m.n_srcs, F_, -1), feats, m.decoder.weight, stride)
        else:
            masked = K.mul_bcast_fwd(mask.reshape(B, m.n_srcs, F_, -1), feats)
            dec = K.ola_convtr_fwd(masked.reshape(B * m.n_srcs, F_, -1), m.decoder.weight, stride)
        return dec.reshape(B, m.n_srcs, -1)


class _LinearPair:
    """concatenated weight codes / dL/dW_q view of two same-input pointwise layers (ops.LinearActQPair)"""
    __slots__ = ("wc", "gw", "Co1", "partner")


class QuantTables:
    """Device-side descriptor tables that let ONE launch each do, for the whole model: the weight
    fake-quant forward (+ int8 codes), its backward, and the range/slope gradient flush (csrc/multi.hip).
    Built once the observer phase is over and the parameters live in a ParamArena (stable addresses)."""

    def __init__(self, model, arena, segments=None):
        """segments: list of parameter-id sets (forward order) -> finish_backward(k) only handles the quantizers whose parameters lie
        in segment k (the gradient all-reduce of a segment needs its weight / range / slope gradients final)"""
        import torch.nn as nn
        from .quantization.qat.qat_layers import LayerQ
        from .quantization.qat.qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize
        dev = arena.flat_p.device
        # ---- activation quantizers: one gacc arena, flush table ---------------------------------
        aqs = [m for m in model.modules() if isinstance(m, GradientActivationFakeQuantize)]
        slope_of = {}
        for layer in model.modules():
            if isinstance(layer, LayerQ) and isinstance(getattr(layer, "nl", None), nn.PReLU):
                slope_of[id(layer.activation_fake_quantize)] = layer.nl.weight
        self.gacc = torch.zeros(len(aqs), K.GACC_DOUBLES, dtype=torch.float64, device=dev)
        rows = []
        for i, m in enumerate(aqs):
            m._buffers["_gacc"] = self.gacc[i]
            m._fqss_deferred = True
            sl = slope_of.get(id(m))
            rows.append([self.gacc[i].data_ptr(), m.min_range.grad.data_ptr(), m.max_range.grad.data_ptr(),
                         sl.grad.data_ptr() if sl is not None else 0])
        self.flush_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.aqs = aqs
        # ---- weight quantizers --------------------------------------------------------------------
        from .quantization.qat.qat_layers import LSTMQ, MultiheadAttentionQ

        def fits(wqm, w):
            return isinstance(wqm, GradientWeightFakeQuantize) and isinstance(w, torch.nn.Parameter) and tuple(wqm.min_range.shape) == tuple(
                1 if d != wqm.axis else w.shape[d] for d in range(w.dim()))

        owners = []
        for layer in model.modules():
            wqm = getattr(layer, "weight_fake_quantize", None)
            if isinstance(wqm, GradientWeightFakeQuantize):
                # convolutions (channel-first and frame path) and the row-major linears of the dual-path / transformer layers
                for cand in ("conv1d", "convTr1d", "residual_encoder", "linear", "conv2d"):
                    conv = getattr(layer, cand, None)
                    if conv is not None and fits(wqm, getattr(conv, "weight", None)):
                        owners.append((wqm, conv.weight))
                        break
            if isinstance(layer, MultiheadAttentionQ):
                for wqm, w in ((layer.weight_fake_quantize_in, layer.mha.in_proj_weight), (layer.weight_fake_quantize_out, layer.mha.out_proj.weight)):
                    if fits(wqm, w):
                        owners.append((wqm, w))
            if isinstance(layer, LSTMQ):
                for name, wqm in layer.weight_quantizers_dict.items():
                    if fits(wqm, getattr(layer.lstm, name, None)):
                        owners.append((wqm, getattr(layer.lstm, name)))
        # layers declared as same-input pairs by their parent (`fqss_linear_pairs`) get adjacent storage: one
        # concatenated code image [Co1+Co2][Ci] (+ transposed, scales, row sums) and one dL/dW_q block, so that the
        # paired q-GEMMs (ops.LinearActQPair) see them as a single weight
        partner = {}
        for mod in model.modules():
            for n1, n2 in getattr(mod, "fqss_linear_pairs", ()):
                l1, l2 = getattr(mod, n1, None), getattr(mod, n2, None)
                w1 = getattr(getattr(l1, "conv1d", None), "weight", None)
                w2 = getattr(getattr(l2, "conv1d", None), "weight", None)
                own = {id(w) for _, w in owners}
                if w1 is None or w2 is None or id(w1) not in own or id(w2) not in own:
                    continue
                ok = w1.dim() == 3 and w1.shape[2] == 1 and w2.shape[1:] == w1.shape[1:] and K.q_eligible(w1.shape[1], w1.shape[0]) \
                    and K.q_eligible(w2.shape[1], w2.shape[0]) and (w1.numel() % 64 == 0) \
                    and (l1.conv1d.bias is None) == (l2.conv1d.bias is None)
                if ok:
                    partner[id(w1)] = w2
        second = {id(w2) for w2 in partner.values()}
        by_id = {id(w): (wqm, w) for wqm, w in owners}
        ordered = []
        for wqm, w in owners:
            if id(w) in second:
                continue
            ordered.append((wqm, w))
            if id(w) in partner:
                ordered.append(by_id[id(partner[id(w)])])
        owners = ordered
        n_gwq = sum((w.numel() + 63) // 64 * 64 for _, w in owners)
        self.gwq = torch.zeros(n_gwq, device=dev)                       # dL/dW_q arena (zeroed once per step)
        self.wq_store = torch.empty(n_gwq, device=dev)
        rows, off, blk = [], 0, 0
        self.weights = []
        self.pairs = []
        pending = None      # (pair codes, first wq, Co1) while the second member of a pair is being laid out
        for wqm, w in owners:
            shape, axis = tuple(w.shape), wqm.axis
            outer = 1
            for d in shape[:axis]:
                outer *= d
            inner = 1
            for d in shape[axis + 1:]:
                inner *= d
            C = shape[axis]
            wq = self.wq_store[off:off + w.numel()].view(shape)
            gwq = self.gwq[off:off + w.numel()].view(shape)
            pw = axis == 0 and w.dim() == 3 and shape[2] == 1 and K.q_eligible(shape[1], shape[0])
            # row-major linear weight [Co, Ci]: the same code image serves the int8 row GEMM (csrc/qrow.hip)
            pw = pw or (axis == 0 and w.dim() == 2 and ops_dp.QROW and K.qrow_eligible(shape[1]))
            # convolution weights [Co, Ci, k] / [Co, Ci, kh, kw] of the frame path (HTDemucs): the same code image over Ci * k "channels"
            # serves the DATA gradient alone (ops.LinearActQ: W_q^T gz on k_qgemm<1>); the forward reads float inputs
            fr = (not pw) and axis == 0 and w.dim() in (3, 4) and shape[0] % 16 == 0 and shape[0] <= 1024 and ops.FRAME_CODES_DGRAD
            wc = None
            ldT = C
            if pw and id(w) in partner:
                # first of a pair: allocate the concatenated images, this layer's are views of rows / columns [0, Co1)
                w2 = partner[id(w)]
                Co1, Co2, Ci = shape[0], w2.shape[0], shape[1]
                pc = K.WCodes()
                pc.Co, pc.Ci = Co1 + Co2, Ci
                pc.idx = torch.empty(Co1 + Co2, Ci, device=dev, dtype=torch.int8)
                pc.idxT = torch.empty(Ci, Co1 + Co2, device=dev, dtype=torch.int8)
                pc.dw = torch.empty(Co1 + Co2, device=dev)
                pc.rw = torch.empty(Co1 + Co2, device=dev)
                pair = _LinearPair()
                pair.wc, pair.Co1 = pc, Co1
                pair.gw = self.gwq[off:off + w.numel() + w2.numel()].view(Co1 + Co2, Ci)
                pending = (pair, wq, 0)
            if pw and pending is not None:
                pair, wq_first, _ = pending
                pc = pair.wc
                r0 = 0 if wq_first is wq else pair.Co1
                wc = K.WCodes()
                wc.Co, wc.Ci = shape[0], shape[1]
                wc.idx = pc.idx[r0:r0 + shape[0]]
                wc.idxT = None                         # column block of pc.idxT (row stride Co1+Co2): paired kernels only
                wc.dw, wc.rw = pc.dw[r0:r0 + shape[0]], pc.rw[r0:r0 + shape[0]]
                ldT = pc.Co
                idxT_ptr = pc.idxT.data_ptr() + r0
                if wq_first is not wq:
                    pair.partner = wq
                    wq_first._fqss_pair = pair
                    self.pairs.append(pair)
                    pending = None
            elif pw or fr:
                wc = K.WCodes()
                wc.Co, wc.Ci = shape[0], (inner if fr else shape[1])
                wc.idx = torch.empty(wc.Co, wc.Ci, device=dev, dtype=torch.int8)
                wc.idxT = torch.empty(wc.Ci, wc.Co, device=dev, dtype=torch.int8)
                wc.dw = torch.empty(shape[0], device=dev)
                wc.rw = torch.empty(shape[0], device=dev)
                idxT_ptr = wc.idxT.data_ptr()
            coded = pw or fr
            rows.append([w.data_ptr(), wq.data_ptr(), wc.idx.data_ptr() if coded else 0, idxT_ptr if coded else 0,
                         wc.dw.data_ptr() if coded else 0, wc.rw.data_ptr() if coded else 0, wqm.min_range.data_ptr(),
                         wqm.max_range.data_ptr(), gwq.data_ptr(), w.grad.data_ptr(), wqm.min_range.grad.data_ptr(),
                         wqm.max_range.grad.data_ptr(), outer, C, inner, blk, ldT])
            wq._fqss_gwq = gwq
            if pw:
                wq._fqss_wcodes = wc
            elif fr:
                wq._fqss_wcodes_dgrad = wc
            w._fqss_wq = wq
            self.weights.append((wqm, w, wq, wc))
            off += (w.numel() + 63) // 64 * 64
            blk += C
        assert pending is None
        self.wq_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.total_channels = blk
        # grouped, atomics-free weight gradients of the quantized 1x1 convolutions (csrc/qgemm.hip k_qwgrad_group); FQSS_GROUP_WGRAD=0:
        # one k_qwgrad2 launch per layer where autograd reaches it (rounds 2-4)
        self.det = None     # kernels.DetMode of the owning step (FQSS_DETERMINISTIC=1)
        self.wgrad_queue = K.WgradQueue() if os.environ.get("FQSS_GROUP_WGRAD", "1") != "0" else None
        # ... and of the row-major linears of the dual-path / transformer models (csrc/gemm_x3.hip k_gemm_x3_wq_multi)
        self.row_wgrad_queue = K.RowWgradQueue() if os.environ.get("FQSS_GROUP_WGRAD", "1") != "0" else None
        # ---- per-segment views of the two finish tables (descriptor 15 = first block is renumbered per table) -----------------
        self.seg_tables = None
        if segments is not None and len(segments) > 1:
            self.seg_tables = []
            flush_rows = self.flush_table.tolist()
            for ids in segments:
                wrows, b0, members = [], 0, []
                for r, (wqm, w, _, _) in zip(rows, self.weights):
                    if id(w) in ids:
                        r = list(r)
                        r[15] = b0
                        b0 += r[13]
                        wrows.append(r)
                        members.append((wqm, w))
                frows = [r for r, m in zip(flush_rows, aqs) if id(m.min_range) in ids]
                self.seg_tables.append((torch.tensor(wrows, dtype=torch.int64, device=dev).reshape(-1, len(rows[0])), b0,
                                        torch.tensor(frows, dtype=torch.int64, device=dev).reshape(-1, 4), members))
            assert sum(t[0].shape[0] for t in self.seg_tables) == len(rows), "a weight quantizer belongs to no backward segment"
            assert sum(t[2].shape[0] for t in self.seg_tables) == len(flush_rows), "an activation quantizer belongs to no backward segment"

    def weights_forward(self):
        K.wq_multi_fwd(self.wq_table, self.total_channels)
        if self.wgrad_queue is not None:
            self.wgrad_queue.jobs = []        # (a backward that never reached finish_backward leaves nothing behind)
        if self.row_wgrad_queue is not None:
            self.row_wgrad_queue.jobs = []

    def finish_backward(self, seg=None):
        """after autograd (of backward segment `seg`, or of the whole network): weight STE/range gradients from the dL/dW_q arena,
        then the range/slope flushes"""
        if self.wgrad_queue is not None:
            # the weight gradients of this segment's quantized 1x1 convolutions, queued by their autograd nodes (ops.LinearActQ /
            # LinearActQPair): ONE grouped launch per <= 25 layers, before the weight STE below reads the dL/dW_q arena
            self.wgrad_queue.flush()
        if self.row_wgrad_queue is not None:
            self.row_wgrad_queue.flush()
        if self.det is not None:
            self.det.finish(1)      # FQSS_DETERMINISTIC=1: the depthwise / frame-path weight gradients' integer sums -> the dL/dW_q arena
        if seg is None or self.seg_tables is None:
            K.wq_multi_bwd(self.wq_table, self.total_channels)
            K.gacc_flush_multi(self.flush_table)
            members = [(wqm, w) for wqm, w, _, _ in self.weights]
        else:
            wt, nch, ft, members = self.seg_tables[seg]
            if wt.shape[0]:
                K.wq_multi_bwd(wt, nch)
            if ft.shape[0]:
                K.gacc_flush_multi(ft)
        for wqm, w in members:
            w._fqss_touched = wqm.min_range._fqss_touched = wqm.max_range._fqss_touched = True


class KDTrainStep:
    """one QAT step = student fwd + teacher fwd + KD loss + bwd (+ grad all-reduce) + clip + Adam
    (reference: System.training_step / common_step, mysystem.py:124-157; Adam + clip 5.0,
    asteroid_librimix_trainer.py:94,132).

    `capture()` records the step into two hipGraphs (fwd+loss+bwd | clip+Adam) once the model has
    left the observer phase; the step has no host sync, so replay needs no Python at all.  At N>1 the
    RCCL all-reduce of the flat gradient buffer runs between the two graphs."""

    def __init__(self, model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=None, loss="sisdr_pit", source_weights=None,
                 batched_quantizers=True, fast=True, coded=True, buckets=None, sync_observer_ranges=True,
                 betas=(0.9, 0.999), loss_threshold=None, teacher_ahead=False):
        """loss: "sisdr_pit" = the asteroid KD loss (mysystem.py:124-151); "sisdr_pit_per_sample" = the speechbrain env's form of it
        (speechbrain_librimix_trainer.py:99-115, 141-149: log per sample, mean over the samples whose loss exceeds `loss_threshold`;
        per-GPU batch 1 or 2, see fqss_kd_loss_per_sample); "l1_sdr" = the htdemucs solver's
        (solver.py:333-366: L1 task + SDR-weighted L1 distillation, per-source weights).  clip <= 0: no clipping (htdemucs.yaml:84).
        batched_quantizers=False keeps every quantizer on its own launches (no QuantTables / codes-only dataflow);
        fast=False: quantizing layers also write their fp32 outputs (no carriers); coded=False: no layer output carries codes,
        every layer runs its un-fused fp32 kernels (the reference dataflow the tests pin the fused step against).
        buckets (world > 1, models with `fqss_segments`): the backward runs as that many separately launched segments and the
        gradient all-reduce of a finished segment overlaps the next segment's backward (default 4; 1 = one all-reduce after the
        whole backward).  sync_observer_ranges (world > 1): average the activation ranges over the ranks once, when the 50-call
        observer phase ends -- a documented deviation: the reference's DDP never re-synchronises the
        observer's .data writes (qat_quant.py:230-232), so its replicas quantize on different grids from then on.
        teacher_ahead: the frozen teacher's forward of the NEXT batch (`step(x, tgt, x_next=...)`) runs on the teacher stream beside
        this whole step -- forward AND backward -- instead of beside the student's forward only: it depends on nothing the step
        changes (mysystem.py:132-133 runs it under no_grad on the raw mixture), so a loader that knows the next batch can hide it
        behind the student's VALU / HBM-bound kernels.  Still one teacher forward per step; the step that finds no look-ahead
        result for its mixture (the first one, or a caller that did not announce it) runs the teacher beside its own forward as
        before.  Same values either way (tests/test_gpu_kdstep_path.py)."""
        self.model, self.fmodel = model, fmodel
        self.cpu = _lib.BACKEND == "cpu"
        if self.cpu:
            # cfg 1 (`--use_cpu`): the CPU backend serves the un-fused per-layer entry points only -- no codes-only dataflow, no batched
            # tables, no streams / graphs, the teacher as the plain module forward
            batched_quantizers = fast = coded = teacher_ahead = False
        self.teacher_ahead = bool(teacher_ahead)
        self.grad_at = None         # l1_sdr only, tests: evaluate dloss/dest at this output instead of the step's own (a parity gate against
                                    # reference gradients must not inherit the sign noise of |est - target| ~ 0 samples)
        self._ahead = None          # (mixture tensor, its version, teacher output) announced by the previous step
        self._tgraph = None         # captured teacher forward (teacher_ahead + hipGraph replay)
        self.loss_kind, self.source_weights, self.batched_quantizers = loss, source_weights, batched_quantizers
        if loss not in ("sisdr_pit", "sisdr_pit_per_sample", "l1_sdr"):
            raise ValueError(f"unknown loss {loss!r}")
        self.loss_threshold = loss_threshold       # per-sample objective only: samples at or below it leave the batch mean
        self.fast, self.coded = bool(fast and coded), bool(coded)
        self._graphs = None
        self.kd_lambda, self.lr, self.clip, self.betas = kd_lambda, lr, clip, tuple(betas)
        self.comm = comm
        # ---- backward segments = gradient buckets (world > 1 only: a single rank has nothing to overlap) -------------------
        self.segments = None        # [(arena lo, arena hi)] per segment in FORWARD order
        self._seg_ids = None
        self._ranges_synced = not (sync_observer_ranges and self._exchange())
        params = list(model.parameters())
        # FQSS_FORCE_BUCKETS=1 with a forced one-rank communicator (FQSS_FORCE_DIST=1): the bucketed schedule of world > 1 at world 1
        force_b = os.environ.get("FQSS_FORCE_BUCKETS", "0") == "1" and self._exchange()
        nb = (4 if buckets is None else int(buckets)) if (self._world() > 1 or force_b) else (int(buckets) if buckets else 1)
        if nb > 1 and hasattr(model, "fqss_segments"):
            segs = model.fqss_segments(nb)
            if len(segs) > 1:
                seen, ordered, self._seg_ids = set(), [], []
                for mods in segs:
                    ids = set()
                    for m in mods:
                        for p in m.parameters():
                            if id(p) not in seen:
                                seen.add(id(p))
                                ordered.append(p)
                            ids.add(id(p))
                    self._seg_ids.append(ids)
                rest = [p for p in params if id(p) not in seen]       # anything the model did not assign: with the first segment
                self._seg_ids[0] |= {id(p) for p in rest}
                params = rest + ordered
        self.arena = ParamArena(params)
        # FQSS_DETERMINISTIC=1: bit-reproducible gradients (kernels.DetMode; no effect on the values beyond fp32 summation order)
        self.det = None
        if os.environ.get("FQSS_DETERMINISTIC", "0") == "1" and not self.cpu:
            self.det = K.DetMode()
            self.det.attach(0, self.arena.flat_g)
        if self._seg_ids is not None:
            off = {id(p): (o, o + (p.numel() + 63) // 64 * 64) for p, o in zip(self.arena.params, self.arena.offsets)}
            self.segments = []
            for ids in self._seg_ids:
                spans = [off[i] for i in ids if i in off]
                self.segments.append((min(a for a, _ in spans), max(b for _, b in spans)))
            assert all(self.segments[k][1] == self.segments[k + 1][0] for k in range(len(self.segments) - 1)), self.segments
        self._cstream = None
        self.comm_events = None     # bench.py: (end of the last backward segment, all exchanges joined) as timing events of a replay
        for p in fmodel.parameters():
            p.requires_grad_(False)
        self.teacher = TeacherRunner(fmodel)     # fused inference chain for the frozen float teacher
        if self.cpu:
            self.teacher.ok = False
        self._tstream = None
        self.last = None
        self.tables = None          # QuantTables once the quantizing phase is reached
        self._eager_q = 0           # eager steps run in the quantizing phase (see maybe_capture)
        self.use_graph = not self.cpu
        self._sx = self._st = None

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        """the learning rate is an argument of the captured clip+Adam launch: a scheduler step drops the graphs (the trainers
        re-capture lazily, see maybe_capture)"""
        if getattr(self, "_lr", None) != value:
            self._graphs = None
        self._lr = value

    def can_capture(self):
        from .quantization.qat.qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize
        for m in self.model.modules():
            if isinstance(m, GradientActivationFakeQuantize) and m.observer_mode and m.n_iter < m.max_observations:
                return False
            if isinstance(m, GradientWeightFakeQuantize) and m.observer_mode:
                return False
        return True

    def maybe_capture(self, x, tgt):
        """training loops: once the observer phase is over (and after every learning-rate change) record the step into hipGraphs
        WITHOUT running it (warmup=0: no extra optimizer step); later calls with same-shaped batches replay.  At least one
        quantizing-phase step must have run eagerly first: it builds the QuantTables and starts the Adam clocks of the activation
        ranges (the captured clip+Adam launch does not activate parameters)"""
        if self.use_graph and self._graphs is None and self._eager_q >= 1 and self.can_capture():
            self.capture(x, tgt, warmup=0)
        elif self._graphs is not None and not self._fits(x, tgt):
            # a batch of another shape (a file-backed loader's short utterance) runs eagerly (__call__); only when the NEW shape stays
            # (three steps in a row) are the graphs dropped and recorded again -- a capture costs seconds, an eager step milliseconds
            self._odd = getattr(self, "_odd", 0) + 1
            if self._odd >= 3:
                self._graphs, self._odd = None, 0
        elif self._graphs is not None:
            self._odd = 0

    def _fits(self, x, tgt):
        return x.shape == self._sx.shape and tgt.shape == self._st.shape

    # ---- the two halves of a step -----------------------------------------------------------
    def _take_ahead(self, x):
        """the teacher output the previous step computed for exactly this mixture tensor, or None"""
        ah, self._ahead = self._ahead, None
        if ah is not None and ah[0] is x and ah[1] == x._version:
            return ah[2]
        return None

    def _forward_loss(self, x, tgt, x_next=None, fest_given=None, join_teacher=True):
        """zero the gradient arenas, weight fake-quants, student + teacher forward, loss and dloss/dest -> (res, est, gest, cuts).
        fest_given: the teacher's output for x, already computed (look-ahead); x_next: start the teacher on the next mixture now"""
        a = self.arena
        a.zero_grad()
        t = self._quant_tables()
        if t is not None:
            t.gwq.zero_()
            t.weights_forward()                # all 101 weight fake-quants (+ int8 codes): one launch
        # The frozen teacher depends on nothing but x: it runs on a second stream next to the student's forward (its
        # MFMA-bound GEMMs overlap the student's HBM-bound layers; inside a hipGraph capture this becomes a parallel
        # branch of the graph) and joins before the loss.
        cur = None if self.cpu else torch.cuda.current_stream()
        teacher_free = self.loss_kind != "l1_sdr" and not self.kd_lambda > 0        # kd_lambda = 0: plain PIT SI-SDR loss, no teacher
        ahead = fest_given is not None
        two_streams = TEACHER_STREAM and not self.cpu
        if teacher_free:
            fest = None
        elif ahead:
            fest = fest_given
            if self._tstream is not None and join_teacher:      # computed on the teacher stream during the previous step (joined BEFORE the next look-ahead is enqueued there)
                cur.wait_stream(self._tstream)
                fest.record_stream(cur)
        elif two_streams:
            if self._tstream is None:
                self._tstream = _teacher_stream()
            self._tstream.wait_stream(cur)
            with torch.cuda.stream(self._tstream):
                fest = self.teacher(x)
        if x_next is not None and not teacher_free:
            # look-ahead: the next step's teacher forward, enqueued behind this step's own (if any) on the teacher stream; nothing in
            # this step waits for it -- the step that consumes it does
            if self._tstream is None:
                self._tstream = _teacher_stream()
            self._tstream.wait_stream(cur)
            with torch.cuda.stream(self._tstream):
                self._ahead = (x_next, x_next._version, self.teacher(x_next))
        cuts = []
        with ops.fast_codes(self.fast), ops.coded_dataflow(self.coded), ops.deferred(t):   # student: codes-only dataflow between quantizing layers
            if self.segments is not None:
                with ops.cut_recorder() as cuts:
                    est = self.model(x)
            else:
                est = self.model(x)
        if teacher_free or ahead:
            pass
        elif two_streams:
            cur.wait_stream(self._tstream)
            fest.record_stream(cur)
        else:
            fest = self.teacher(x)
        if self.loss_kind == "l1_sdr":
            if self.source_weights is None:
                self.source_weights = torch.ones(tgt.shape[1], device=tgt.device)
            loss, task, kd, w, gest = K.hd_kd_loss(est.detach(), fest, tgt, self.source_weights, self.kd_lambda, want_grad=True)
            if self.grad_at is not None:       # tests: dloss/dest taken at a given output (the L1 loss has a sign gradient)
                gest = K.hd_kd_loss(self.grad_at, fest, tgt, self.source_weights, self.kd_lambda, want_grad=True)[4]
            res = dict(loss=loss, task=task, kd=kd, w=w, gnorm=a.gnorm, est=est.detach())
        elif teacher_free:
            # mysystem.py:153-156: PITLossWrapper(pairwise_neg_sisdr) on the student alone
            out, sisdr, gest = K.pit_sisdr_loss(est.detach(), tgt, want_grad=True)
            res = dict(loss=out[0], kd_loss=out[1], task=out[2], kd=out[3], w=None, sisdr=sisdr, gnorm=a.gnorm, est=est.detach())
        else:
            # "sisdr_pit_per_sample": the speechbrain env's objective (log per sample, thresholded mean; csrc/train_ops.hip)
            out, w, sisdr, gest = K.kd_loss(est.detach(), fest, tgt, self.kd_lambda, want_grad=True,
                                            per_sample=self.loss_kind == "sisdr_pit_per_sample", threshold=self.loss_threshold)
            res = dict(loss=out[0], kd_loss=out[1], task=out[2], kd=out[3], w=w, sisdr=sisdr, gnorm=a.gnorm, est=est.detach())
        return res, est, gest, cuts

    def _backward_segment(self, k, est, gest, cuts):
        """backward segment k (k = 0: the END of the network ... k = len(cuts): its start) + that segment's quantizer finish"""
        t = self.tables
        regular = [c for c in cuts if not c[2]]
        nseg = len(regular) + 1
        if self.det is not None:
            if not torch.cuda.is_current_stream_capturing():
                self.det.activate()     # (no-op unless another step's DetMode has taken the device-wide control block since)
            if k == 0:
                self.det.begin_backward()
        with ops.deferred(t):
            if k == 0:
                est.backward(gest)
            else:
                roots = [regular[nseg - 1 - k]] + ([c for c in cuts if c[2]] if k == nseg - 1 else [])
                pairs = [(o, l.grad) for orig, leaves, _ in roots for o, l in zip(orig, leaves)
                         if o is not None and o.requires_grad and l.grad is not None]
                torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        if t is not None:
            t.finish_backward(None if nseg == 1 else nseg - 1 - k)     # weight STE + range/slope gradients of this segment
        if self.det is not None:
            self.det.finish(0)      # the integer sums of this segment's bias / row-sum gradients -> the fp32 arena, before its exchange

    def _fwd_bwd(self, x, tgt, x_next=None, fest_given=None):
        """fwd + loss + the whole backward, no exchange (single rank; tests)"""
        res, est, gest, cuts = self._forward_loss(x, tgt, x_next, fest_given)
        for k in range(self._nseg(cuts)):
            self._backward_segment(k, est, gest, cuts)
        return res

    @staticmethod
    def _nseg(cuts):
        return sum(1 for c in cuts if not c[2]) + 1

    def _reduce_segment(self, k, nseg):
        """all-reduce(SUM) the gradient slice of backward segment k on the communication stream, behind everything enqueued on the
        current stream so far (RCCL over xGMI; the 1/world factor is folded into the clip+Adam kernel)"""
        if not self._exchange():
            return
        lo, hi = (0, self.arena.numel) if self.segments is None or nseg == 1 else self.segments[nseg - 1 - k]
        if self._cstream is None:
            self._cstream = torch.cuda.Stream()
        self._cstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._cstream):
            self.comm.all_reduce_sum(self.arena.flat_g[lo:hi])

    def _join_reduces(self):
        if self._exchange() and self._cstream is not None:
            torch.cuda.current_stream().wait_stream(self._cstream)

    def _quant_tables(self):
        """batched per-quantizer work becomes available once every observer has finished"""
        if not self.batched_quantizers:
            return None
        if self.tables is None:
            from .quantization.qat.qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize
            for m in self.model.modules():
                if isinstance(m, GradientActivationFakeQuantize) and m.observer_mode and m.n_iter < m.max_observations:
                    return None
                if isinstance(m, GradientWeightFakeQuantize) and m.observer_mode:
                    return None
            self.tables = QuantTables(self.model, self.arena, segments=self._seg_ids)
            if self.det is not None:
                self.det.attach(1, self.tables.gwq)
                self.tables.det = self.det
        return self.tables

    def _world(self):
        return self.comm.world if self.comm is not None else 1

    def _exchange(self):
        """collectives are issued: world > 1, or a forced one-rank communicator (parallel.Comm.force)"""
        return self.comm is not None and getattr(self.comm, "active", self.comm.world > 1)

    def _optimize(self, activate=True):
        if activate:
            self.arena._activate_touched()
        self.arena.clip_adam_step(self.lr, self.clip, 1.0 / self._world(), betas=self.betas, activate=False)

    def _maybe_sync_ranges(self):
        """world > 1: once, when the observer phase is over, every rank takes the mean of the observed activation ranges"""
        if not self._ranges_synced and self.can_capture():
            self.comm.sync_observer_ranges(self.model)
            self._ranges_synced = True

    def _step_eager(self, x, tgt, x_next=None):
        """fwd + loss, then per backward segment: backward -> all-reduce of its gradient slice on the communication stream, which
        overlaps the next segment's backward; the optimizer waits for the last exchange"""
        fest = self._take_ahead(x) if self.teacher_ahead else None
        res, est, gest, cuts = self._forward_loss(x, tgt, x_next if self.teacher_ahead else None, fest)
        nseg = self._nseg(cuts)
        for k in range(nseg):
            self._backward_segment(k, est, gest, cuts)
            self._reduce_segment(k, nseg)
        self._join_reduces()
        return res

    def __call__(self, x, tgt, x_next=None):
        """x_next (teacher_ahead=True only): the mixture tensor the NEXT call will be given -- the same tensor object, unmodified"""
        self._maybe_sync_ranges()
        if self._graphs is not None and self.use_graph and self._fits(x, tgt):
            return self.replay(x, tgt, x_next if (x_next is not None and x_next.shape == x.shape) else None)
        self.last = self._step_eager(x, tgt, x_next)
        self._optimize()
        if self.tables is not None or (not self.batched_quantizers and self.can_capture()):
            self._eager_q += 1
        return self.last

    # ---- hipGraph capture -------------------------------------------------------------------
    def capture(self, x, tgt, warmup=2):
        # a prefetching loader's reader thread allocates and launches on its own stream: not while this thread records a graph (the
        # default capture mode faults on another thread's hipMalloc / event calls)
        from .loader import DEVICE_WORK_LOCK
        with DEVICE_WORK_LOCK:
            return self._capture(x, tgt, warmup)

    def _capture(self, x, tgt, warmup=2):
        from .quantization.qat.qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize
        for m in self.model.modules():
            if isinstance(m, GradientActivationFakeQuantize):
                assert not (m.observer_mode and m.n_iter < m.max_observations), "capture() needs the observer phase to be over"
            if isinstance(m, GradientWeightFakeQuantize):
                assert not m.observer_mode, "capture() needs the weight observers to have run"
        self._maybe_sync_ranges()
        self._sx, self._st = x.clone(), tgt.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):              # eager steps: activates every live parameter's Adam clock (bench: warm caches)
                self.last = self._step_eager(self._sx, self._st)
                self._optimize()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # one hipGraph per backward segment (the first also holds fwd + loss) and one for clip + Adam: at world > 1 the exchange of
        # a segment's gradients is launched between two replays and overlaps the next one (no collective inside a graph)
        # (at world > 1 the process group's watchdog thread polls its events while this thread captures: only THIS thread's calls are
        # checked against the capture then -- torch's default mode lets a query from any thread invalidate it)
        mode = dict(capture_error_mode="thread_local") if self._exchange() else {}
        # teacher_ahead: the teacher's forward is a graph of its OWN (own memory pool: it replays on the teacher stream WHILE the step's
        # graphs replay) from the static look-ahead mixture _sxn into _fest_next; the step's graphs read the static copy _fest_cur
        fest_cur = None
        self._tgraph = None
        self._ahead_key = None
        if self.teacher_ahead and (self.loss_kind == "l1_sdr" or self.kd_lambda > 0):
            if self._tstream is None:
                self._tstream = _teacher_stream()
            self._sxn = x.clone()
            tg = torch.cuda.CUDAGraph()
            self._tstream.wait_stream(cur)
            with torch.cuda.graph(tg, stream=self._tstream, **mode):
                self._fest_next = self.teacher(self._sxn)
            cur.wait_stream(self._tstream)
            self._fest_cur = fest_cur = torch.empty_like(self._fest_next)
            self._tgraph = tg
        graphs = [torch.cuda.CUDAGraph()]
        with torch.cuda.graph(graphs[0], **mode):
            self.last, est, gest, cuts = self._forward_loss(self._sx, self._st, None, fest_cur, False)
            self._backward_segment(0, est, gest, cuts)
        for k in range(1, self._nseg(cuts)):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, pool=graphs[0].pool(), **mode):
                self._backward_segment(k, est, gest, cuts)
            graphs.append(gk)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=graphs[0].pool(), **mode):
            self._optimize(activate=False)
        self._graphs = (graphs, g2)
        del est, gest, cuts
        # single rank, one segment: nothing has to happen between the two halves, so the whole step is ALSO captured as one graph
        # (replay() then costs one launch; replay_fwd_bwd / replay_optimize keep serving the trainers that may skip an update)
        self._graph_all = None
        if not self._exchange() and len(graphs) == 1 and os.environ.get("FQSS_ONE_GRAPH", "1") != "0":
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, pool=graphs[0].pool()):
                self.last_all, est, gest, cuts = self._forward_loss(self._sx, self._st, None, fest_cur, False)
                self._backward_segment(0, est, gest, cuts)
                self._optimize(activate=False)
            self._graph_all = ga
            del est, gest, cuts
        return self

    def _stage(self, x, tgt, x_next):
        """inputs of a replay into the static buffers; teacher_ahead: hand the teacher output of this mixture (the previous replay's
        look-ahead, or a teacher replay right now if nobody announced the mixture) to the step's graphs, then start the teacher graph
        on the next mixture on the teacher stream"""
        if self._tgraph is None:
            if x is not None and x.data_ptr() != self._sx.data_ptr():
                self._sx.copy_(x)
                self._st.copy_(tgt)
            return
        cur, ts = torch.cuda.current_stream(), self._tstream
        src = self._sx if x is None else x
        key = self._ahead_key
        if not (key is not None and key[0] is src and key[1] == src._version):
            eager = self._take_ahead(src)          # an eager step (before the capture) may have looked ahead for this mixture
            ts.wait_stream(cur)
            with torch.cuda.stream(ts):
                if eager is not None:
                    self._fest_next.copy_(eager)
                else:
                    if src.data_ptr() != self._sxn.data_ptr():
                        self._sxn.copy_(src)
                    self._tgraph.replay()
        cur.wait_stream(ts)
        self._fest_cur.copy_(self._fest_next)
        if x is not None and x.data_ptr() != self._sx.data_ptr():
            self._sx.copy_(x)
            self._st.copy_(tgt)
        self._ahead_key = None
        if x_next is not None:
            self._sxn.copy_(x_next)
            ts.wait_stream(cur)
            with torch.cuda.stream(ts):
                self._tgraph.replay()
            self._ahead_key = (x_next, x_next._version)

    def _own_det(self):
        """FQSS_DETERMINISTIC=1: the device-wide control block must point at THIS step's shadows while its graphs replay -- another step's
        activate() since the capture would leave the replayed grad_adds on plain fp32 atomics without a word"""
        if self.det is not None and K.DetMode.owner is not self.det:
            self.det.activate()

    def replay_fwd_bwd(self, x=None, tgt=None, x_next=None):
        """first half of a captured step (fwd + loss + bwd, gradients exchanged); the caller then calls replay_optimize() or skips"""
        self._own_det()
        self._stage(x, tgt, x_next)
        graphs = self._graphs[0]
        ev = self.comm_events
        for k, gk in enumerate(graphs):
            gk.replay()
            if ev is not None and k == len(graphs) - 1:
                ev[0].record()                  # the backward is over: whatever the optimizer still waits for is exposed exchange time
            self._reduce_segment(k, len(graphs))
        self._join_reduces()
        if ev is not None:
            ev[1].record()
        return self.last

    def replay_optimize(self):
        self.arena._host_step += 1
        self._graphs[1].replay()

    def replay(self, x=None, tgt=None, x_next=None):
        if getattr(self, "_graph_all", None) is not None:
            self._own_det()
            self._stage(x, tgt, x_next)
            self.arena._host_step += 1
            self._graph_all.replay()
            return self.last_all
        self.replay_fwd_bwd(x, tgt, x_next)
        self.replay_optimize()
        return self.last


class InferRunner:
    """Quantized inference as it would be served: the eval-mode forward on the codes-only dataflow
    (activations travel between layers as u8 codes, the 1x1 convolutions run on the integer codes: bit-identical to the plain eval
    forward) captured into one hipGraph per input shape, so a request costs one graph launch."""

    def __init__(self, model, use_graph=True):
        from .quantization.qat.models.load_model import enable_observer
        self.model = model.eval()
        enable_observer(self.model, False)
        self.use_graph = use_graph
        self._graphs = {}
        if hasattr(model, "n_srcs"):
            self.n_srcs = model.n_srcs         # process.model_infer asks its `model` for it

    @torch.no_grad()
    def _forward(self, x):
        with ops.fast_codes(True):
            return self.model(x)

    @torch.no_grad()
    def __call__(self, x):
        if not self.use_graph:
            return self._forward(x)
        key = tuple(x.shape)
        ent = self._graphs.get(key)
        if ent is None:
            sx = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                      # warm-up outside the capture (allocator, lazy tables)
                    self._forward(sx)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._forward(sx)
            ent = self._graphs[key] = (g, sx, out)
        g, sx, out = ent
        if x.data_ptr() != sx.data_ptr():
            sx.copy_(x)
        g.replay()
        return out
