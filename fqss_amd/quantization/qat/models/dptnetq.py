"""DPTNet (dual-path transformer network) ready for W8A8 fake-quantization, MI355X edition (cfg 3).

Same module tree / attribute names / constructor arguments / `quantize_model` path table as the reference's
quantization/qat/models/dptnetq.py (TransformerEncoderLayer :60-97, Encoder :107-128, Decoder :130-141, SingleTransformer
:143-157, DPT :159-209, DPT_base :211-279, BF_module :281-309, DPTNetQ :311-478), so float and quantized `state_dict`s
interchange key for key.  Every op -- float or quantized -- executes as a HIP kernel (no ATen compute).

Layout (MI355X-first, differs from the reference's tensors but not from its numbers): the front and back ends are channel-first
[B, C, L] like ConvTasNet's; between `split_feature` and `merge_feature` the activations are sequence-first row matrices
[L', B', N] with the 64 features contiguous.  The intra-chunk view is [K, B*S, N], the inter-chunk view [S, B*K, N]; one
transposing copy (fqss_permute4) moves between them, where the reference does permute().contiguous() four times per block.
The decoder's Linear runs as a 1x1 conv on the channel-first tensor, so the largest activation of the network ([B, 2, 256, L])
is never transposed.
"""
import torch
import torch.nn as nn

from .... import kernels as K
from .... import ops, ops_dp
from ....process import postprocess, preprocess
from .. import qat_layers as QL
from ..float_exec import HipSequential, apply_module
from ..qat_layers import Add, Mul
from ..qat_utils import quantize_modules, replace_decoderq, replace_encoderq


def overlap_and_add(signal, frame_step):
    """[..., frames, 2] with hop 1 -> [..., frames + 1]  (the only geometry the model uses, dptnetq.py:140)"""
    if signal.shape[-1] != 2 or frame_step != 1:
        raise NotImplementedError("overlap_and_add: only 2-sample frames with hop 1 have a HIP kernel")
    lead = signal.shape[:-2]
    y = signal.reshape(-1, signal.shape[-2], 2).transpose(1, 2)
    return ops_dp.Ola2.apply(y).reshape(*lead, -1)


class CutTail(torch.autograd.Function):
    """x[..., :T] as a dense tensor (merge_feature drops the `rest` padding after its AddQ, dptnetq.py:273-274)"""

    @staticmethod
    def forward(ctx, x, T):
        ctx.full = x.shape[-1]
        h = x[..., :T]
        return K.axpby(h, h, 0.0)

    @staticmethod
    def backward(ctx, g):
        gx = K.empty_act(tuple(g.shape[:-1]) + (ctx.full,), g.device).zero_()
        gx[..., :g.shape[-1]].copy_(g)
        return gx, None


def _float_mha(mha, x):
    QL._mha_check(mha, x, x, x)
    return QL.run_mha(mha, x, mha.in_proj_weight, mha.out_proj.weight, None, None, None)


def _float_lstm(lstm, x):
    return QL.run_lstm(lstm, x, {n: getattr(lstm, n) for n in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse")},
                       None)


def run(m, x):
    """float or quantized member on a row-major tensor"""
    if isinstance(m, QL.LayerQ):
        y = m(x)
        return y[0] if isinstance(y, (list, tuple)) else y
    if isinstance(m, nn.Linear):
        return QL.run_linear(m, x, m.weight, None, None)
    if isinstance(m, nn.LayerNorm):
        return QL.run_layernorm(m, x, None)
    if isinstance(m, nn.LSTM):
        return _float_lstm(m, x)
    if isinstance(m, nn.PReLU):
        return QL.fq_node(None, x, m)
    return apply_module(m, x)


class TransformerEncoderLayer(nn.Module):
    """the "improved" transformer layer: self-attention, then LSTM -> ReLU -> Linear instead of the feed-forward pair"""

    def __init__(self, d_model, nhead, hidden_size, dim_feedforward, dropout, activation="relu"):
        super().__init__()
        if dropout != 0:
            raise NotImplementedError("dropout > 0 is not used by the FQSS DPTNet (DPT passes dropout=0, dptnetq.py:172)")
        if activation != "relu":
            raise NotImplementedError("only the relu activation has a HIP kernel")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.lstm = nn.LSTM(d_model, hidden_size, 1, bidirectional=True)
        self.dropout = nn.Dropout(dropout)
        self.linear = nn.Linear(hidden_size * 2, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.add_norm1 = Add()
        self.add_norm2 = Add()
        self._relu = nn.ReLU()

    def forward(self, src, then=None):
        """src [L, B', N] sequence-first (dptnetq.py:84-97).  then = ("cols" | "rows", B): the dual-path layout change DPT.forward applies to
        this layer's output -- folded into the last AddQ + LayerNormQ kernel where that runs fused, applied behind it otherwise"""
        s_att, s_res = ops.fork2(src)
        if isinstance(self.self_attn, nn.MultiheadAttention):
            src2 = _float_mha(self.self_attn, s_att)
        else:
            src2 = self.self_attn(s_att, s_att, s_att)[0]
        src = self._add_norm(self.add_norm1, self.norm1, s_res, src2)
        s_rnn, s_res = ops.fork2(src)
        if isinstance(self.lstm, QL.LSTMQ):
            h = self.lstm(s_rnn, post_relu=True)[0]                      # F.relu between LSTMQ and LinearQ: in the output quantizer's pass
        else:
            h = QL.fq_node(None, run(self.lstm, s_rnn), self._relu)     # (float model)
        src2 = run(self.linear, h)
        return self._add_norm(self.add_norm2, self.norm2, s_res, src2, then)

    @staticmethod
    def _add_norm(add, norm, a, b, then=None):
        if isinstance(add, QL.LayerQ) and isinstance(norm, QL.LayerQ):
            return QL.addq_layernorm(add, norm, a, b, then)    # quantizing phase: one kernel each way (fqss_addq_layernorm_*)
        if (isinstance(add, QL.Add) and isinstance(norm, nn.LayerNorm) and QL.FUSE_ADDLN and len(norm.normalized_shape) == 1
                and norm.elementwise_affine and a.shape == b.shape and a.dim() == 3):
            # float model: add + LayerNorm as one kernel; under no_grad (the frozen teacher) it also writes the next layer's layout
            a, b = ops.real(a), ops.real(b)
            rmap = ops_dp.layout_map(tuple(a.shape), then[0], then[1]) if (then is not None and QL.FUSE_LN_LAYOUT and not torch.is_grad_enabled()) else None
            y, _ = ops_dp.AddLayerNormRows.apply(a, b, norm.weight, norm.bias, norm.eps, None, None, None, False, None, None, None, rmap)
            return y if (then is None or rmap is not None) else ops_dp.change_layout(y, then[0], then[1])
        y = run(norm, add(a, b))
        return y if then is None else ops_dp.change_layout(y, then[0], then[1])


class Encoder(nn.Module):
    def __init__(self, W=2, N=64):
        super().__init__()
        self.W, self.N = W, N
        self.conv1d_U = nn.Conv1d(1, N, kernel_size=W, stride=W // 2, bias=False)
        self.relu = nn.ReLU()

    def forward(self, mixture):
        if isinstance(self.conv1d_U, nn.Conv1d):
            return QL.run_conv1d(self.conv1d_U, mixture, self.conv1d_U.weight, self.relu, None)
        return self.conv1d_U(mixture)            # Conv1dEncoderQ carries the ReLU; self.relu is nn.Identity then


class Decoder(nn.Module):
    def __init__(self, E, W):
        super().__init__()
        self.E, self.W = E, W
        self.basis_signals = nn.Linear(E, W, bias=False)

    def forward_cf(self, x):
        """x [B', E, L] channel-first -> list (n_combiner) of [B', L + W - 1]"""
        if self.W != 2:
            raise NotImplementedError("Decoder: only the 2-sample basis (kernel_size=2) has an overlap-add kernel")
        bs = self.basis_signals
        if isinstance(bs, nn.Linear):
            L = ops._Lin("pw", w_param=bs.weight)
            ops_dp.touch(bs.weight)
            frames = [ops.LinearActQ.apply(ops.real(x), bs.weight.view(self.W, self.E, 1), None, None, None, None, L, ops.ACT_NONE, ops.BYPASS)]
        else:
            frames = bs.forward_cf(x)
        return [ops_dp.Ola2.apply(f) for f in frames]

    def forward(self, mixture_w):
        """reference signature: [B, C, L, E] -> [(D,) B, C, T]"""
        return overlap_and_add(run(self.basis_signals, mixture_w), self.W // 2)


class SingleTransformer(nn.Module):
    def __init__(self, input_size, hidden_size, dropout):
        super().__init__()
        self.transformer = TransformerEncoderLayer(d_model=input_size, nhead=4, hidden_size=hidden_size,
                                                   dim_feedforward=hidden_size * 2, dropout=dropout)

    def forward(self, x, then=None):
        """x is ALREADY sequence-first [L, B', N] here (the reference permutes a batch-first tensor, :156)"""
        return self.transformer(x, then)


class DPT(nn.Module):
    def __init__(self, input_size, hidden_size, output_size, num_layers=1, dropout=0):
        super().__init__()
        self.input_size, self.output_size, self.hidden_size = input_size, output_size, hidden_size
        self.row_transformer = nn.ModuleList([SingleTransformer(input_size, hidden_size, dropout) for _ in range(num_layers)])
        self.col_transformer = nn.ModuleList([SingleTransformer(input_size, hidden_size, dropout) for _ in range(num_layers)])
        self.output = HipSequential(nn.PReLU(), nn.Conv2d(input_size, output_size, 1))

    fqss_cut_every = 0     # > 0: a backward cut point (ops.cut) in front of every that many (row, col) transformer pairs

    def forward(self, seg, B):
        """seg [K, B*S, N] intra-chunk rows -> [S, B*K, output_size] inter-chunk rows"""
        Kc, BS, N = seg.shape
        S = BS // B
        x = seg
        n = len(self.row_transformer)
        ce = self.fqss_cut_every
        for i in range(n):
            if ce and i and i % ce == 0:
                (x,) = ops.cut(x)                # the transformer pairs before this one are a backward segment of their own
            x = self.row_transformer[i](x, ("cols", B))                       # -> [S, B*K, N] (rows_to_cols)
            x = self.col_transformer[i](x, ("rows", B) if i + 1 < n else None)     # -> [K, B*S, N] (cols_to_rows)
        x = run(self.output[0], x)
        conv = self.output[1]
        if isinstance(conv, nn.Conv2d):
            return ops_dp.RowLinear.apply(x, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)
        return conv.forward_rows(x)


class DPT_base(nn.Module):
    def __init__(self, input_dim, feature_dim, hidden_dim, num_spk=2, layer=6, segment_size=250):
        super().__init__()
        self.input_dim, self.feature_dim, self.hidden_dim = input_dim, feature_dim, hidden_dim
        self.layer, self.segment_size, self.num_spk = layer, segment_size, num_spk
        self.eps = 1e-8
        self.BN = nn.Conv1d(self.input_dim, self.feature_dim, 1, bias=False)
        self.DPT = DPT(self.feature_dim, self.hidden_dim, self.feature_dim * self.num_spk, num_layers=layer)
        self.add = Add()


class BF_module(DPT_base):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.output = HipSequential(nn.Conv1d(self.feature_dim, self.feature_dim, 1), nn.Tanh())
        self.output_gate = HipSequential(nn.Conv1d(self.feature_dim, self.feature_dim, 1), nn.Sigmoid())
        self.mul = Mul()

    @staticmethod
    def _gated(seq, x):
        conv, nl = seq[0], seq[1]
        if isinstance(conv, nn.Conv1d):
            return QL.run_conv1d(conv, x, conv.weight, nl, None)
        return conv(x)

    def forward(self, x):
        """x [B, E, L] -> gated filters [B*nspk, N, L] channel-first (the reference returns its [B, nspk, L, N] transpose, which
        DPTNetQ.forward transposes straight back, dptnetq.py:307, 390)"""
        B, _, T = x.shape
        N, Kc, nspk = self.feature_dim, self.segment_size, self.num_spk
        f = apply_module(self.BN, x)
        seg = ops_dp.Segment.apply(ops.real(f), Kc)                      # [K, B*S, N]
        o = self.DPT(seg, B)                                             # [S, B*K, nspk*N]
        a, b = ops_dp.MergeStreams.apply(o, B, nspk, N, Kc)              # [B*nspk, N, Lm]
        m = self.add(a, b)
        if m.shape[-1] != T:
            m = CutTail.apply(ops.real(m), T)
        m1, m2 = ops.fork2(m)
        return self.mul(self._gated(self.output, m1), self._gated(self.output_gate, m2))


class DPTNetQ(nn.Module):
    def __init__(self, n_spks=2, kernel_size=2, enc_dim=256, feature_dim=64, hidden_dim=128, layer=6, segment_size=250):
        super().__init__()
        self.set_splitter_combiner(1, 1)
        self.window = kernel_size
        self.enc_dim, self.feature_dim, self.hidden_dim, self.segment_size = enc_dim, feature_dim, hidden_dim, segment_size
        self.layer = layer
        self.n_srcs = n_spks
        self.eps = 1e-8
        self.encoder = Encoder(kernel_size, enc_dim)
        self.enc_LN = nn.GroupNorm(1, self.enc_dim, eps=self.eps)
        self.separator = BF_module(self.enc_dim, self.feature_dim, self.hidden_dim, self.n_srcs, self.layer, self.segment_size)
        self.mask_conv1x1 = HipSequential(nn.Conv1d(self.feature_dim, self.enc_dim, 1, bias=False), nn.ReLU())
        self.decoder = Decoder(enc_dim, kernel_size)
        self.mul = Mul()

    def pre_process(self, x):
        return preprocess(x, n_splitter=self.n_splitter)

    def post_process(self, x):
        return postprocess(x, n_combiner=self.n_combiner)

    def forward(self, x):
        """x [B, 1, T] (or [B, T]) -> separated sources [B, S, T]"""
        with ops.fast_codes(False):       # the dual-path kernels take real fp32 activations (codes-only carriers are ConvTasNet's)
            x = self.pre_process(x)
            B = x.shape[0]
            w = self.encoder(x)                                                   # [B, E, L]
            w_mask, w_mul = ops.fork2(w)
            if self.separator.DPT.fqss_cut_every:
                (w_mul,) = ops.cut(w_mul, late=True)       # this edge jumps over every backward segment of the separator
            g = self.separator(apply_module(self.enc_LN, w_mask))                 # [B*S, N, L]
            m = self.mask_conv1x1(g)                                              # [B*S, E, L]
            E, L = self.enc_dim, m.shape[-1]
            sw = self.mul(ops.real(w_mul).reshape(B, 1, E, L), ops.real(m).reshape(B, self.n_srcs, E, L))
            est = self.decoder.forward_cf(ops.real(sw).reshape(B * self.n_srcs, E, L))   # n_combiner x [B*S, T]
            out = est[0] if len(est) == 1 else torch.stack(est)
            return self.post_process(out.reshape(self.n_combiner, B, self.n_srcs, 1, -1))

    def fqss_segments(self, n):
        """Backward segments = gradient buckets (runtime.KDTrainStep, see ConvTasNetQ.fqss_segments): groups of (row, col) transformer
        pairs; returns the module lists in forward order."""
        dpt, sep = self.separator.DPT, self.separator
        nl = len(dpt.row_transformer)
        n = max(1, min(int(n), nl))
        every = -(-nl // n)
        dpt.fqss_cut_every = every if n > 1 else 0
        if n == 1:
            return [[self]]
        segs = []
        for s0 in range(0, nl, every):
            mods = []
            for i in range(s0, min(nl, s0 + every)):
                mods += [dpt.row_transformer[i], dpt.col_transformer[i]]
            segs.append(mods)
        segs[0] = [self.encoder, self.enc_LN, sep.BN] + segs[0]
        segs[-1] = segs[-1] + [dpt.output, sep.add, sep.output, sep.output_gate, sep.mul, self.mask_conv1x1, self.mul, self.decoder]
        return segs

    def load_pretrain(self, weights_path):
        own = self.state_dict()
        loaded = torch.load(weights_path, map_location="cpu", weights_only=False)   # trusted local checkpoint
        loaded = loaded.get("state_dict", loaded)
        loaded = {k: v for k, v in loaded.items() if not k.startswith("fmodel.")}
        assert len(own) == len(loaded), ("Error: mismatch models weights. Please check if the model configurations "
                                         "match to model weights!")
        self.load_state_dict({nk: v for nk, v in zip(own.keys(), loaded.values())}, strict=True)

    def set_splitter_combiner(self, n_splitter, n_combiner):
        self.n_splitter = n_splitter
        self.n_combiner = n_combiner

    def quantize_model(self, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8,
                       inout_nl_quant=False, in_quant=False, in_act_n_bits=8, out_quant=True, out_act_n_bits=8):
        p = {"gradient_based": gradient_based, "act_quant": act_quant, "weight_quant": weight_quant,
             "weight_n_bits": weight_n_bits, "act_n_bits": act_n_bits}
        io = {"gradient_based": gradient_based, "act_quant": act_quant, "inout_nl_quant": inout_nl_quant,
              "weight_quant": weight_quant, "weight_n_bits": weight_n_bits}
        for _, m in list(self.named_modules()):
            if type(m) is DPTNetQ:
                replace_encoderq(m.encoder, ["conv1d_U", "relu"], dict(io, n_splitter=self.n_splitter, act_n_bits=act_n_bits,
                                                                      in_quant=in_quant, in_act_n_bits=in_act_n_bits))
                replace_decoderq(m.decoder, ["basis_signals"], dict(io, n_combiner=self.n_combiner, act_n_bits=out_act_n_bits,
                                                                   out_quant=out_quant, out_act_n_bits=out_act_n_bits))
                quantize_modules(m, ["enc_LN"], p)
                quantize_modules(m.mask_conv1x1, ["0", "1"], p)
                quantize_modules(m, ["mul"], p)
            elif type(m) is TransformerEncoderLayer:
                for name in ("lstm", "linear", "norm1", "norm2", "add_norm1", "add_norm2", "self_attn"):
                    quantize_modules(m, [name], p)
            elif type(m) is DPT:
                quantize_modules(m.output, ["0"], p)
                quantize_modules(m.output, ["1"], p)
            elif type(m) is BF_module:
                quantize_modules(m.output, ["0", "1"], p)
                quantize_modules(m.output_gate, ["0", "1"], p)
                for name in ("mul", "add", "BN"):
                    quantize_modules(m, [name], p)
