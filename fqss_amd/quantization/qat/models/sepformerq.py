"""Sepformer ready for W8A8 fake-quantization, MI355X edition (cfg 4).

Same module tree / attribute names / constructor arguments / `quantize_model` path table as the reference's
quantization/qat/models/sepformerq.py (PositionalEncoding :13-47, TransformerLayer :50-97, TransformerBlock :100-123,
DualPathBlock :126-177, MaskGenerator :180-345, SepformerQ :348-526), so float and quantized `state_dict`s interchange key for
key.  Every op -- float or quantized -- executes as a HIP kernel (no ATen compute).

Layout: as for DPTNet (models/dptnetq.py) the dual-path blocks hold sequence-first row matrices -- intra-chunk [K, B*S, F],
inter-chunk [S, B*K, F], F = 256 features contiguous -- so LayerNorm, the linears and the attention read rows, the block's
GroupNorm(1, F) runs on the row layout directly (fqss_gnrows_*: statistics over all rows of a sample) and ONE transposing copy
per direction replaces the reference's permute().contiguous() pairs (:151-157, 165-173).
"""
import math

import torch
import torch.nn as nn

from .... import ops, ops_dp
from ....process import postprocess, preprocess
from .. import qat_layers as QL
from ..float_exec import HipSequential, apply_module
from ..qat_layers import Add, Const, Mul
from ..qat_utils import quantize_modules, replace_decoderq, replace_encoderq
from .dptnetq import CutTail, _float_mha, run

EPS_T = 1e-6
EPS = 1e-8


def _fadd(a, b):
    """the reference's plain `+` (residual connections, over_add): a float add, no quantizer"""
    return ops.AddActQ.apply(ops.real(a), ops.real(b), None, None, 1.0, ops.BYPASS)


class PositionalEncoding(nn.Module):
    """absolute sinusoidal positional encoding; the table is a buffer (a state_dict key, like the reference's)"""

    def __init__(self, input_size, max_len=2500, device="cpu"):
        super().__init__()
        self.max_len = max_len
        pe = torch.zeros(self.max_len, input_size, requires_grad=False, device=device)
        positions = torch.arange(0, self.max_len).unsqueeze(1).float()
        denominator = torch.exp(torch.arange(0, input_size, 2).float() * -(math.log(10000.0) / input_size))
        pe[:, 0::2] = torch.sin(positions * denominator)
        pe[:, 1::2] = torch.cos(positions * denominator)
        self.register_buffer("pe", pe.unsqueeze(0))
        self.const = Const()

    def forward(self, x):
        """x: sequence-first rows [L, B', F] -> the (quantized) encoding [L, 1, F]"""
        L = x.size(0)
        if L > self.max_len:
            raise ValueError(f"sequence of {L} positions exceeds max_len={self.max_len}")
        return self.const(self.pe[0, :L].clone().detach()).reshape(L, 1, -1)


class TransformerLayer(nn.Module):
    def __init__(self, n_filters, n_ffn, n_heads, dropout=0.0):
        super().__init__()
        if dropout != 0:
            raise NotImplementedError("dropout > 0 is not used by the FQSS Sepformer")
        self.mha = nn.MultiheadAttention(n_filters, n_heads, dropout=dropout, batch_first=False)
        self.ffn = HipSequential(nn.Linear(n_filters, n_ffn), nn.ReLU(), nn.Dropout(dropout), nn.Linear(n_ffn, n_filters))
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(n_filters, eps=EPS_T)
        self.norm2 = nn.LayerNorm(n_filters, eps=EPS_T)

    def forward(self, x, add=None):
        """x [L, B', F] sequence-first (the reference holds it batch-first and permutes around the attention, :77-79).
        `add`: a tensor that is still to be ADDED to x (the previous layer's feed-forward output): the float residual adds of the
        reference (`x = x + attn`, `x = x + ffn`, :78-82) are folded into the LayerNorm that follows them (QL.add_layernorm: one kernel
        each way instead of add + norm, and no separate gradient sum at the fork).  Returns (residual stream, pending addend)."""
        if add is None:
            x_n, x_res = ops.fork2(x)
            q = run(self.norm1, x_n)
        else:
            q, x_res = QL.add_layernorm(self.norm1, x, add)
        a = _float_mha(self.mha, q) if isinstance(self.mha, nn.MultiheadAttention) else self.mha(q, q, q)[0]
        h, x_res = QL.add_layernorm(self.norm2, x_res, a)
        mods = [m for m in self.ffn if not isinstance(m, (nn.Dropout, nn.Identity))]
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, QL.LinearQ) and i + 1 < len(mods) and isinstance(mods[i + 1], QL.NlQ) and isinstance(mods[i + 1].nl, nn.ReLU):
                h = QL.linear_then_relu_q(m, mods[i + 1], h)          # LinearQ -> NlQ(ReLU): both quantizers in the GEMM's epilogue
                i += 2
                continue
            if isinstance(m, QL.LinearQ) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                h = m(h, post_relu=True)          # LinearQ -> nn.ReLU: the ReLU rides in the output quantizer's pass each way
                i += 2
                continue
            h = QL.fq_node(None, h, m) if isinstance(m, nn.ReLU) else run(m, h)
            i += 1
        return x_res, h


class TransformerBlock(nn.Module):
    def __init__(self, n_filters, n_heads, n_ffn, num_layers=8, dropout=0.0, device="cpu"):
        super().__init__()
        self.layers = nn.ModuleList([TransformerLayer(n_filters, n_heads=n_heads, n_ffn=n_ffn, dropout=dropout)
                                     for _ in range(num_layers)])
        self.norm = nn.LayerNorm(n_filters, eps=EPS_T)
        self.pos = PositionalEncoding(n_filters, device=device)
        self.pos_add = Add()

    def forward(self, x):
        x = self.pos_add(x, self.pos(x))
        add = None
        for layer in self.layers:
            x, add = layer(x, add)
        return QL.add_layernorm(self.norm, x, add)[0]        # the last layer's `x + ffn` and the block's final norm


def _gln_rows(gn, x, geom):
    if isinstance(gn, QL.GroupNormQ):
        return gn.forward_rows(x, geom)
    return QL.run_groupnorm_rows(gn, x, None, geom)


class DualPathBlock(nn.Module):
    def __init__(self, n_filters, n_heads, n_ffn, dropout=0.0, device="cpu"):
        super().__init__()
        self.intra_transformer_block = TransformerBlock(n_filters=n_filters, n_heads=n_heads, n_ffn=n_ffn, dropout=dropout, device=device)
        self.inter_transformer_block = TransformerBlock(n_filters=n_filters, n_heads=n_heads, n_ffn=n_ffn, dropout=dropout, device=device)
        self.intra_norm = nn.GroupNorm(num_groups=1, num_channels=n_filters, eps=EPS)
        self.inter_norm = nn.GroupNorm(num_groups=1, num_channels=n_filters, eps=EPS)
        self.intra_add = Add()
        self.inter_add = Add()

    fqss_cut_inside = False     # a backward cut point (ops.cut) between the intra and the inter half, set by SepformerQ.fqss_segments()

    def forward(self, x, B):
        """x [K, B*S, F] intra-chunk rows -> same layout"""
        Kc, BS, _ = x.shape
        S = BS // B
        x_in, x_res = ops.fork2(x)
        intra = self.intra_transformer_block(x_in)
        intra = self.intra_add(_gln_rows(self.intra_norm, intra, (B * S, S, B)), x_res)
        if self.fqss_cut_inside:
            (intra,) = ops.cut(intra)
        i_in, i_res = ops.fork2(intra)
        inter = self.inter_transformer_block(ops_dp.rows_to_cols(ops.real(i_in), B, S))          # [S, B*K, F]
        inter = _gln_rows(self.inter_norm, inter, (B * Kc, Kc, B))
        return self.inter_add(ops_dp.cols_to_rows(ops.real(inter), B, Kc), i_res)


class MaskGenerator(nn.Module):
    def __init__(self, n_srcs, n_filters, n_repeats=2, n_heads=8, chunk_size=250, n_ffn=1024, dropout=0.0, device="cpu"):
        super().__init__()
        self.n_srcs = n_srcs
        self.chunk_size = chunk_size
        self.norm = nn.GroupNorm(num_groups=1, num_channels=n_filters, eps=EPS)
        self.conv1d = nn.Conv1d(n_filters, n_filters, 1, bias=False)
        self.layers = nn.ModuleList([DualPathBlock(n_filters, n_heads=n_heads, n_ffn=n_ffn, dropout=dropout, device=device)
                                     for _ in range(n_repeats)])
        self.conv2d = nn.Conv2d(n_filters, n_srcs * n_filters, kernel_size=1, bias=True)
        self.end_conv = HipSequential(nn.Conv1d(n_filters, n_filters, 1, bias=False), nn.ReLU())
        self.prelu = nn.PReLU()
        self.net_out = HipSequential(nn.Conv1d(n_filters, n_filters, 1, bias=True), nn.Tanh())
        self.net_gate = HipSequential(nn.Conv1d(n_filters, n_filters, 1, bias=True), nn.Sigmoid())
        self.mul = Mul()

    fqss_cut_between = False    # backward cut points between the dual-path blocks, set by SepformerQ.fqss_segments()

    @staticmethod
    def _gated(seq, x):
        conv, nl = seq[0], seq[1]
        if isinstance(conv, nn.Conv1d):
            return QL.run_conv1d(conv, x, conv.weight, nl, None)
        return conv(x)

    def forward(self, x):
        """x [B, F, M] -> masks [B, n_srcs, F, M]"""
        B, F_, M = x.shape
        Kc = self.chunk_size
        xc = apply_module(self.conv1d, apply_module(self.norm, x))
        seg = ops_dp.Segment.apply(ops.real(xc), Kc)                           # [K, B*S, F]
        S = seg.shape[1] // B
        for i, layer in enumerate(self.layers):
            if i and self.fqss_cut_between:
                (seg,) = ops.cut(seg)           # the dual-path blocks before this one are a backward segment of their own
            seg = layer(seg, B)
        y = ops_dp.rows_to_cols(ops.real(run(self.prelu, seg)), B, S)          # [S, B*K, F]
        conv = self.conv2d
        if isinstance(conv, nn.Conv2d):
            o = ops_dp.RowLinear.apply(y, conv.weight.view(conv.out_channels, conv.in_channels), conv.bias)
        else:
            o = conv.forward_rows(y)
        a, b = ops_dp.MergeStreams.apply(o, B, self.n_srcs, F_, Kc)            # over_add (:297-327): a float add, then the gap cut
        m = _fadd(a, b)
        if m.shape[-1] != M:
            m = CutTail.apply(m, M)
        m1, m2 = ops.fork2(m)
        out = self.end_conv(self.mul(self._gated(self.net_out, m1), self._gated(self.net_gate, m2)))
        return ops.reshape_tagged(out, B, self.n_srcs, F_, -1)


class SepformerQ(nn.Module):
    def __init__(self, n_spks=1, kernel_size=16, stride=8, n_filters=256, n_repeats=2, n_heads=8, chunk_size=250, device="cpu"):
        super().__init__()
        self.n_srcs = n_spks
        self.enc_num_feats = n_filters
        self.set_splitter_combiner(1, 1)
        self.encoder = HipSequential(nn.Conv1d(1, n_filters, kernel_size, stride=stride, padding=0, bias=False), nn.ReLU())
        self.masker = MaskGenerator(n_spks, n_filters, n_repeats=n_repeats, n_heads=n_heads, chunk_size=chunk_size, device=device)
        self.decoder = nn.ConvTranspose1d(n_filters, 1, kernel_size, stride=stride, padding=0, bias=False)
        self.mul = Mul()

    def pre_process(self, x):
        return preprocess(x, n_splitter=self.n_splitter)

    def post_process(self, x):
        return postprocess(x, n_combiner=self.n_combiner)

    def forward(self, x):
        with ops.fast_codes(False):
            x = self.pre_process(x)
            batch = x.shape[0]
            feats = self.encoder(x)                                            # [B, F, M]
            f_mask, f_mul = ops.fork2(feats)
            if self.masker.fqss_cut_between:
                (f_mul,) = ops.cut(f_mul, late=True)       # this edge jumps over every backward segment of the masker
            masked = self.mul(self.masker(f_mask), ops.reshape_tagged(f_mul, batch, 1, self.enc_num_feats, -1))
            masked = ops.reshape_tagged(masked, batch * self.n_srcs, self.enc_num_feats, -1)
            out = apply_module(self.decoder, masked)
            return self.post_process(out.reshape((self.n_combiner, batch, self.n_srcs, 1, -1)))

    def fqss_segments(self, n):
        """Backward segments = gradient buckets (runtime.KDTrainStep, see ConvTasNetQ.fqss_segments): the dual-path blocks (n >= 2), each
        split again between its intra and inter transformer stacks (n >= 2 * blocks); returns the module lists in forward order."""
        mk = self.masker
        blocks = list(mk.layers)
        n = int(n)
        inside = n >= 2 * len(blocks)
        between = n >= 2 and len(blocks) > 1
        mk.fqss_cut_between = between or inside
        for blk in blocks:
            blk.fqss_cut_inside = inside
        if not (between or inside):
            return [[self]]
        segs = []
        for blk in blocks:
            first = [blk.intra_transformer_block, blk.intra_norm, blk.intra_add]
            second = [blk.inter_transformer_block, blk.inter_norm, blk.inter_add]
            if inside:
                segs += [first, second]
            else:
                segs.append(first + second)
        segs[0] = [self.encoder, mk.norm, mk.conv1d] + segs[0]
        segs[-1] = segs[-1] + [mk.prelu, mk.conv2d, mk.net_out, mk.net_gate, mk.mul, mk.end_conv, self.mul, self.decoder]
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
            if type(m) is SepformerQ:
                replace_encoderq(m.encoder, ["0", "1"], dict(io, n_splitter=self.n_splitter, act_n_bits=act_n_bits,
                                                             in_quant=in_quant, in_act_n_bits=in_act_n_bits))
                replace_decoderq(m, ["decoder"], dict(io, n_combiner=self.n_combiner, act_n_bits=act_n_bits, out_quant=out_quant,
                                                      out_act_n_bits=out_act_n_bits, train_res_dec=True))
                quantize_modules(m, ["mul"], p)
            elif type(m) is TransformerBlock:
                quantize_modules(m, ["norm"], p)
                quantize_modules(m, ["pos_add"], p)
                quantize_modules(m.pos, ["const"], p)
            elif type(m) is TransformerLayer:
                for name in ("norm1", "norm2", "mha"):
                    quantize_modules(m, [name], p)
                for name in ("0", "1", "3"):
                    quantize_modules(m.ffn, [name], p)
            elif type(m) is DualPathBlock:
                for name in ("inter_norm", "intra_norm", "inter_add", "intra_add"):
                    quantize_modules(m, [name], p)
            elif type(m) is MaskGenerator:
                quantize_modules(m.net_out, ["0", "1"], p)
                quantize_modules(m.net_gate, ["0", "1"], p)
                for name in ("norm", "conv1d", "conv2d"):
                    quantize_modules(m, [name], p)
                quantize_modules(m.end_conv, ["0", "1"], p)
                quantize_modules(m, ["prelu"], p)
                quantize_modules(m, ["mul"], p)
