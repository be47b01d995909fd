"""Hybrid Transformer Demucs ready for W8A8 fake-quantization, MI355X edition (cfg 5).

Same module tree / attribute names / constructor arguments / `quantize_model` path table as the reference's
quantization/qat/models/htdemucsq.py (HTDemucsQ :528-1245, CrossTransformerEncoder :330-525, MyTransformerEncoderLayer :138-222,
CrossTransformerEncoderLayer :224-328, MyGroupNorm :124-136), hdemucsq.py (ScaledEmbedding :43-69, HEncLayer :72-162, HDecLayer
:261-347) and demucsq.py (LayerScale :19-39, DConv :110-182), so float and quantized `state_dict`s interchange key for key.
Every op -- float or quantized -- executes as a HIP kernel; torch only moves memory (views, dense copies of strided views).

What is MI355X-specific here:
  * convolutions of any geometry = frame gather + the pointwise GEMM kernels (csrc/conv_frames.hip), transposed ones = GEMM +
    deterministic overlap-add; the crop after a transposed conv is a dense copy of the kept window;
  * the spectrogram pair is csrc/stft.hip (an LDS FFT per frame, `_spec` / `_ispec` framing folded into the indices);
  * the transformer holds batch-first rows [B, T, C] end to end: LayerNorm / linears / LayerScale read rows, the attention core
    (csrc/attn_long.hip) addresses heads as column blocks of the in-projection, MyGroupNorm runs on the row layout
    (fqss_gnrows_*), so the reference's transposes around attention and GroupNorm never materialise.
Defaults follow the reference; the un-used variants (Wiener filtering, MultiWrap, sparse attention, CAPE / scaled embeddings,
LSTM / attention inside DConv, branch merging when nfft is small) raise NotImplementedError.
"""
import math
from fractions import Fraction

import torch
import torch.nn as nn

from .... import kernels as K
from .... import ops, ops_dp
from ....process import postprocess, preprocess
from .. import qat_layers as QL
from ..qat_layers import Add, Const, Mul
from ..qat_utils import quantize_modules, replace_decoderq, replace_encoderq


# ----------------------------------------------------------------------------------------------------------------------
# float members on the HIP kernels (BYPASS mode), quantized members as they are
# ----------------------------------------------------------------------------------------------------------------------
def run(m, *xs, window=None):
    """window: the slice of a transposed convolution's output the caller keeps (QL.convtr_frames); modules that cannot take it return the
    whole output and the caller crops"""
    if isinstance(m, QL.LayerQ):
        y = m(*xs, window=window) if (window is not None and isinstance(m, (QL._ConvTrQ, QL.ConvTr1dDecoderQ, QL.ConvTr2dDecoderQ))) else m(*xs)
        return y[0] if isinstance(y, (list, tuple)) else y
    x = xs[0]
    if isinstance(m, (nn.Identity, nn.Dropout)):
        return x
    if isinstance(m, (nn.Conv1d, nn.Conv2d)):
        return QL.conv_frames(m, x, m.weight)
    if isinstance(m, (nn.ConvTranspose1d, nn.ConvTranspose2d)):
        return QL.convtr_frames(m, x, m.weight, window=window)
    if isinstance(m, MyGroupNorm):
        return m(x)
    if isinstance(m, nn.GroupNorm):
        return QL.run_groupnorm(m, x, None)
    if isinstance(m, (nn.GELU, nn.GLU, nn.ReLU)):
        return QL.run_nl(m, x, None)
    if isinstance(m, nn.Linear):
        return QL.run_linear(m, x, m.weight, None, None)
    if isinstance(m, nn.LayerNorm):
        return QL.run_layernorm(m, x, None)
    if isinstance(m, nn.Embedding):
        return ops_dp.EmbeddingRows.apply(m.weight, x)
    if isinstance(m, nn.MultiheadAttention):
        return QL.run_mha_x(m, xs[0], xs[1], xs[2], m.in_proj_weight, m.out_proj.weight, None, None, None)
    return m(*xs)


class _Window(torch.autograd.Function):
    """x.narrow(dim, start, length) as a dense tensor (the crop after a transposed conv, hdemucsq.py:340-345) / zero right padding
    (negative start is not needed: pad = True appends zeros).  Pure data movement."""

    @staticmethod
    def forward(ctx, x, dim, start, length, pad_to):
        ctx.cfg = (dim, start, length, x.shape[dim])
        if pad_to is None:
            v = x.narrow(dim, start, length)
            if CROP_VIEWS and dim == x.dim() - 2 and x.dim() == 4 and K.rowmat_collapsed_ok(v):
                # a crop along the frequency axis of [B, C, F, T]: planes of F' * T dense floats at the old plane stride -- the element-wise
                # kernels behind a decoder layer (the next layer's skip add, the activation quantizer) read such a view in place
                return v
            out = K.empty_act(tuple(v.shape), x.device) if dim == x.dim() - 1 else torch.empty(v.shape, device=x.device, dtype=x.dtype)
            out.copy_(v)
            return out
        shape = list(x.shape)
        shape[dim] = pad_to
        out = K.empty_act(tuple(shape), x.device)
        out.narrow(dim, 0, x.shape[dim]).copy_(x)
        out.narrow(dim, x.shape[dim], pad_to - x.shape[dim]).zero_()      # (only the appended tail is filled, not the whole tensor first)
        ctx.cfg = (dim, 0, x.shape[dim], None)
        return out

    @staticmethod
    def backward(ctx, g):
        dim, start, length, full = ctx.cfg
        if full is None:                      # padding: the gradient is the leading window
            v = g.narrow(dim, 0, length)
            out = K.empty_act(tuple(v.shape), g.device)
            out.copy_(v)
            return out, None, None, None, None
        shape = list(g.shape)
        shape[dim] = full
        out = torch.empty(shape, device=g.device, dtype=g.dtype)
        out.narrow(dim, start, length).copy_(g)
        if start > 0:                         # the two margins get their zeros, the window is written once
            out.narrow(dim, 0, start).zero_()
        if start + length < full:
            out.narrow(dim, start + length, full - start - length).zero_()
        return out, None, None, None, None


CROP_VIEWS = __import__("os").environ.get("FQSS_CROP_VIEWS", "1") != "0"      # (A/B knob: "0" = every crop is a dense copy)
FOLD_CROP = __import__("os").environ.get("FQSS_FOLD_CROP", "1") != "0"        # (A/B knob: "0" = transposed convolutions write their whole output)


def crop(x, dim, start, length):
    x = ops.real(x)
    dim = dim % x.dim()
    if start == 0 and length == x.shape[dim]:
        return x
    return _Window.apply(x, dim, start, length, None)


def pad_right(x, total):
    x = ops.real(x)
    return x if total == x.shape[-1] else _Window.apply(x, x.dim() - 1, 0, x.shape[-1], total)


class _Transpose(torch.autograd.Function):
    """swap the last two dims through the tiled transpose kernel (fqss_transpose2d)"""

    @staticmethod
    def forward(ctx, x):
        return K.transpose2d(x)

    @staticmethod
    def backward(ctx, g):
        return K.transpose2d(g)


SWAP_IN_PLACE = __import__("os").environ.get("FQSS_SWAP_IN_PLACE", "1") != "0"    # swap_mid reads a row-padded view in place (A/B)


SWAP_PAD_ROWS = __import__("os").environ.get("FQSS_SWAP_PAD_ROWS", "1") != "0"   # (A/B knob)


def swap_mid(x, rows_out=None):
    """[B, P, Q, T] -> [B, Q, P, T] (T-long contiguous chunks move: fqss_permute4).  rows_out True: the result is streamed as rows of T
    floats (DConv over [B F, C, T]): a row-padded activation, rows 16-B aligned, so the element-wise kernels behind it take their 16-B
    forms (T = 431); False: the INPUT was such rows (DConv's output), the result dense planes -- the gradient handed back is row-padded.
    None: dense both ways."""
    x = ops.real(x)
    if x.stride(-1) != 1 or not SWAP_IN_PLACE:
        x = x.contiguous()
    B, P, Q, T = x.shape
    sB, sP, sQ, _ = x.stride()          # (a view of a row-padded buffer -- DConv's output reshaped -- is read in place)
    pad = SWAP_PAD_ROWS and rows_out is not None
    return ops_dp.Permute4.apply(x, (B, Q, P), (sB, sQ, sP), (B, P, Q), (P * Q * T, T, P * T), False, (0, 2, 1) if pad else None,
                                 pad and rows_out, pad and not rows_out)


def _fadd(a, b):
    return ops.AddActQ.apply(ops.real(a), ops.real(b), None, None, 1.0, ops.BYPASS)


# ----------------------------------------------------------------------------------------------------------------------
# demucsq.py / hdemucsq.py blocks
# ----------------------------------------------------------------------------------------------------------------------
class LayerScale(nn.Module):
    """x * scale per channel (demucsq.py:19-39); channel_last: rows [.., C], else channel-first [B, C, T]"""

    def __init__(self, channels, init=0, channel_last=False):
        super().__init__()
        self.channel_last = channel_last
        self.scale = nn.Parameter(torch.zeros(channels, requires_grad=True))
        self.scale.data[:] = init
        self.mul = Mul()

    def forward(self, x):
        return self.mul(x, self.scale if self.channel_last else self.scale[:, None])


FUSE_TEACHER_DCONV = __import__("os").environ.get("FQSS_FUSE_TEACHER_DCONV", "1") != "0"


class DConv(nn.Module):
    """residual branches of dilated convs on [B', C, T] (demucsq.py:110-182; no LSTM / attention variants)"""

    def __init__(self, channels, compress=4, depth=2, init=1e-4, norm=True, attn=False, heads=4, ndecay=4, lstm=False, gelu=True,
                 kernel=3, dilate=True):
        super().__init__()
        assert kernel % 2 == 1
        if attn or lstm:
            raise NotImplementedError("DConv: the LocalState / BLSTM variants are not used by HTDemucsQ")
        self.channels, self.compress, self.depth = channels, compress, abs(depth)
        dilate = depth > 0
        norm_fn = (lambda d: nn.GroupNorm(1, d)) if norm else (lambda d: nn.Identity())
        hidden = int(channels / compress)
        act = nn.GELU if gelu else nn.ReLU
        self.layers = nn.ModuleList([])
        self.adds = nn.ModuleList([])
        for d in range(self.depth):
            dilation = 2 ** d if dilate else 1
            padding = dilation * (kernel // 2)
            self.layers.append(nn.Sequential(nn.Conv1d(channels, hidden, kernel, dilation=dilation, padding=padding), norm_fn(hidden), act(),
                                             nn.Conv1d(hidden, 2 * channels, 1), norm_fn(2 * channels), nn.GLU(1),
                                             LayerScale(channels, init)))
            self.adds.append(Add())

    def forward(self, x):
        for layer, add in zip(self.layers, self.adds):
            if FUSE_TEACHER_DCONV and not torch.is_grad_enabled():
                y = self._forward_nograd(layer, x)
                if y is not None:
                    x = y
                    continue
            x_in, x_res = ops.fork2(x)
            h = x_in
            for m in layer:
                h = run(m, h)
            x = add(x_res, h)
        return x

    @staticmethod
    def _forward_nograd(layer, x):
        """the float layer without autograd (the frozen teacher, mysystem.py:132-133 / solver.py:333-340 run it under no_grad): the two
        GroupNorm apply passes carry what follows them -- GELU; GLU, LayerScale and the residual add (fqss_gn_fwd_tail) -- four
        element-wise launches per layer instead of eight, value for value the module path"""
        mods = list(layer)
        if len(mods) != 7 or not (type(mods[0]) is nn.Conv1d and type(mods[1]) is nn.GroupNorm and type(mods[2]) is nn.GELU
                                  and type(mods[3]) is nn.Conv1d and type(mods[4]) is nn.GroupNorm and type(mods[5]) is nn.GLU
                                  and type(mods[6]) is LayerScale) or mods[1].num_groups != 1 or mods[4].num_groups != 1 \
                or getattr(mods[2], "approximate", "none") != "none" or mods[5].dim != 1 or mods[6].channel_last or x.dim() != 3:
            return None
        x = ops.real(x)
        h = ops.real(run(mods[0], x))
        h = K.gn_fwd_tail(h, mods[1].weight, mods[1].bias, mods[1].eps, 1)
        if h is None:
            return None
        h = ops.real(run(mods[3], h))
        return K.gn_fwd_tail(h, mods[4].weight, mods[4].bias, mods[4].eps, 2, ls=mods[6].scale, res=x)


class ScaledEmbedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, scale=10., smooth=False):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings, embedding_dim)
        if smooth:
            weight = torch.cumsum(self.embedding.weight.data, dim=0)
            weight = weight / torch.arange(1, num_embeddings + 1).to(weight).sqrt()[:, None]
            self.embedding.weight.data[:] = weight
        self.embedding.weight.data /= scale
        self.scale = scale
        self.mul = Mul()

    @property
    def weight(self):
        return self.embedding.weight * self.scale

    def forward(self, x):
        return self.mul(run(self.embedding, x), self.scale)


def _dconv_freq(dconv, y):
    """DConv over time with the frequency axis folded into the batch (hdemucsq.py:150-156)"""
    B, C, Fr, T = y.shape
    z = dconv(swap_mid(y, True).reshape(B * Fr, C, T))
    return swap_mid(ops.real(z).reshape(B, Fr, C, T), False)


def _takes_pad(m):
    """a convolution whose frame gather can stand in for the zero padding in front of it (QL.conv_frames pad_to): general geometry"""
    conv = m.conv1d if isinstance(m, QL.Conv1dNlQ) else m
    return isinstance(m, (QL.Conv1dNlQ, nn.Conv1d)) and isinstance(conv, nn.Conv1d) and QL.conv1d_geometry(conv).kind == "gather"


class HEncLayer(nn.Module):
    def __init__(self, chin, chout, kernel_size=8, stride=4, norm_groups=1, empty=False, freq=True, dconv=True, norm=True, context=0,
                 dconv_kw={}, pad=True, rewrite=True):
        super().__init__()
        if norm:
            raise NotImplementedError("HEncLayer: GroupNorm layers (norm_starts <= depth) are not used by HTDemucsQ")
        pad = kernel_size // 4 if pad else 0
        klass = nn.Conv1d
        self.freq, self.kernel_size, self.stride, self.empty, self.norm, self.pad = freq, kernel_size, stride, empty, norm, pad
        if freq:
            kernel_size, stride, pad, klass = [kernel_size, 1], [stride, 1], [pad, 0], nn.Conv2d
        self.conv = klass(chin, chout, kernel_size, stride, pad)
        if self.empty:
            return
        self.norm1 = nn.Identity()
        self.rewrite = None
        if rewrite:
            self.rewrite = klass(chout, 2 * chout, 1 + 2 * context, 1, context)
            self.norm2 = nn.Identity()
        self.dconv = DConv(chout, **dconv_kw) if dconv else None
        self.gelu = nn.GELU()
        self.glu = nn.GLU(dim=1)

    def forward(self, x, inject=None):
        x = ops.real(x)
        if not self.freq and x.dim() == 4:
            B, C, Fr, T = x.shape
            x = x.reshape(B, -1, T)
        pad_to = None
        if not self.freq:
            le = x.shape[-1]
            if le % self.stride:
                pad_to = le + self.stride - le % self.stride
                if not (FOLD_CROP and _takes_pad(self.conv)):
                    x, pad_to = pad_right(x, pad_to), None
        y = run(self.conv, x) if pad_to is None else (self.conv(x, pad_to=pad_to) if isinstance(self.conv, QL.LayerQ) else
                                                      QL.conv_frames(self.conv, x, self.conv.weight, pad_to))
        if self.empty:
            return y
        if inject is not None:
            raise NotImplementedError("HEncLayer: merging the time branch into the frequency branch needs nfft small enough "
                                      "for the branches to meet; not reached with the FQSS configuration")
        y = run(self.gelu, y)
        if self.dconv:
            y = _dconv_freq(self.dconv, y) if self.freq else self.dconv(y)
        if self.rewrite:
            return run(self.glu, run(self.rewrite, y))
        return y


class HDecLayer(nn.Module):
    def __init__(self, chin, chout, last=False, kernel_size=8, stride=4, norm_groups=1, empty=False, freq=True, dconv=True, norm=True,
                 context=1, dconv_kw={}, pad=True, context_freq=True, rewrite=True):
        super().__init__()
        if norm:
            raise NotImplementedError("HDecLayer: GroupNorm layers (norm_starts <= depth) are not used by HTDemucsQ")
        self.pad = kernel_size // 4 if pad else 0
        self.last, self.freq, self.chin, self.empty, self.stride, self.kernel_size = last, freq, chin, empty, stride, kernel_size
        self.norm, self.context_freq = norm, context_freq
        klass, klass_tr = nn.Conv1d, nn.ConvTranspose1d
        if freq:
            kernel_size, stride, klass, klass_tr = [kernel_size, 1], [stride, 1], nn.Conv2d, nn.ConvTranspose2d
        self.conv_tr = klass_tr(chin, chout, kernel_size, stride)
        self.norm2 = nn.Identity()
        if not self.last:
            self.gelu = nn.GELU()
        if self.empty:
            return
        self.rewrite = None
        if rewrite:
            if context_freq:
                self.rewrite = klass(chin, 2 * chin, 1 + 2 * context, 1, context)
            else:
                self.rewrite = klass(chin, 2 * chin, [1, 1 + 2 * context], 1, [0, context])
            self.norm1 = nn.Identity()
        self.dconv = DConv(chin, **dconv_kw) if dconv else None
        self.add = Add()
        self.glu = nn.GLU(dim=1)

    def forward(self, x, skip, length):
        x = ops.real(x)
        if self.freq and x.dim() == 3:
            B, C, T = x.shape
            x = x.reshape(B, self.chin, -1, T)
        if not self.empty:
            x = self.add(x, ops.real(skip))
            y = run(self.glu, run(self.rewrite, x)) if self.rewrite else x
            if self.dconv:
                y = _dconv_freq(self.dconv, y) if self.freq else self.dconv(y)
        else:
            y = x
            assert skip is None
        y_pre = y            # (`pre` only feeds an `empty` time decoder: not reached with the FQSS configuration)
        # the crop behind the transposed convolution (hdemucsq.py:340-345) is handed to it as a window: GELU and the fake-quant in
        # between are element-wise, so cropping first gives the same numbers without the margins ever being written
        window = None
        if FOLD_CROP:
            full = (y.shape[-2 if self.freq else -1] - 1) * self.stride + self.kernel_size
            if self.freq and self.pad:
                window = (-2, self.pad, full - 2 * self.pad)
            elif not self.freq:
                window = (-1, self.pad, length)
        z = run(self.conv_tr, y, window=window)
        if not self.last:
            z = run(self.gelu, z)
        if self.freq:
            if self.pad and not (window is not None and z.shape[-2] == window[2]):
                z = crop(z, -2, self.pad, z.shape[-2] - 2 * self.pad)
        elif not (window is not None and z.shape[-1] == length):
            z = crop(z, -1, self.pad, length)
        if not self.freq:
            assert z.shape[-1] == length, (z.shape[-1], length)
        return z, y_pre


# ----------------------------------------------------------------------------------------------------------------------
# cross-domain transformer
# ----------------------------------------------------------------------------------------------------------------------
def create_sin_embedding(length, dim, shift=0, device="cpu", max_period=10000):
    assert dim % 2 == 0
    pos = shift + torch.arange(length, device=device).view(-1, 1, 1)
    half_dim = dim // 2
    adim = torch.arange(dim // 2, device=device).view(1, 1, -1)
    phase = pos / (max_period ** (adim / (half_dim - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)


def create_2d_sin_embedding(d_model, height, width, device="cpu", max_period=10000):
    if d_model % 4 != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dimension (got dim={:d})".format(d_model))
    pe = torch.zeros(d_model, height, width)
    d_model = int(d_model / 2)
    div_term = torch.exp(torch.arange(0.0, d_model, 2) * -(math.log(max_period) / d_model))
    pos_w = torch.arange(0.0, width).unsqueeze(1)
    pos_h = torch.arange(0.0, height).unsqueeze(1)
    pe[0:d_model:2, :, :] = torch.sin(pos_w * div_term).transpose(0, 1).unsqueeze(1).repeat(1, height, 1)
    pe[1:d_model:2, :, :] = torch.cos(pos_w * div_term).transpose(0, 1).unsqueeze(1).repeat(1, height, 1)
    pe[d_model::2, :, :] = torch.sin(pos_h * div_term).transpose(0, 1).unsqueeze(2).repeat(1, 1, width)
    pe[d_model + 1::2, :, :] = torch.cos(pos_h * div_term).transpose(0, 1).unsqueeze(2).repeat(1, 1, width)
    return pe[None, :].to(device)


class MyGroupNorm(nn.GroupNorm):
    """GroupNorm(1, C) of batch-first rows x [B, T, C]: statistics over all of (T, C) per sample (htdemucsq.py:124-136), then Const"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.const = Const()

    def forward(self, x):
        if self.num_groups != 1 or not self.affine:
            raise NotImplementedError("MyGroupNorm: only num_groups = 1 with affine parameters has a HIP kernel")
        x = ops.real(x)
        B, T, C = x.shape
        y = ops_dp.GroupNormRows.apply(x.contiguous(), self.weight, self.bias, self.eps, (B * T, T, B))
        return self.const(y)


class _EncoderLayerBase(nn.Module):
    def _ff_block(self, x):
        return run(self.linear2, run(self.activation, run(self.linear1, x)))

    def _post(self, x):
        return self.norm_out(x) if self.norm_out is not None else x


def _layer_members(self, d_model, nhead, dim_feedforward, dropout, activation_is_gelu, layer_norm_eps, layer_scale, init_values, norm_first,
                   group_norm, norm_out, names):
    if dropout != 0:
        raise NotImplementedError("dropout > 0 is not used by the FQSS HTDemucs")
    if not norm_first:
        raise NotImplementedError("norm_first=False is not used by the FQSS HTDemucs")
    self.linear1 = nn.Linear(d_model, dim_feedforward)
    self.dropout = nn.Dropout(dropout)
    self.linear2 = nn.Linear(dim_feedforward, d_model)
    self.norm_first = norm_first
    for n in names:
        setattr(self, n, MyGroupNorm(int(group_norm), d_model, eps=layer_norm_eps) if group_norm else nn.LayerNorm(d_model, eps=layer_norm_eps))
    self.norm_out = MyGroupNorm(num_groups=int(norm_out), num_channels=d_model) if (norm_first and norm_out) else None
    self.gamma_1 = LayerScale(d_model, init_values, True) if layer_scale else nn.Identity()
    self.gamma_2 = LayerScale(d_model, init_values, True) if layer_scale else nn.Identity()
    self.dropout1 = nn.Dropout(dropout)
    self.dropout2 = nn.Dropout(dropout)
    self.activation = nn.GELU() if activation_is_gelu else nn.ReLU()
    self.add_norm1 = Add()
    self.add_norm2 = Add()


class MyTransformerEncoderLayer(_EncoderLayerBase):
    """pre-norm self-attention layer on batch-first rows (htdemucsq.py:138-222)"""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation=None, group_norm=0, norm_first=False, norm_out=False,
                 layer_norm_eps=1e-5, layer_scale=False, init_values=1e-4, sparse=False, batch_first=False, gelu=True, **unused):
        super().__init__()
        if sparse:
            raise NotImplementedError("sparse attention is not used by the FQSS HTDemucs")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=batch_first)
        _layer_members(self, d_model, nhead, dim_feedforward, dropout, gelu, layer_norm_eps, layer_scale, init_values, norm_first, group_norm,
                       norm_out, ("norm1", "norm2"))

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        x_res, x_n = ops.fork2(ops.real(src))
        q = run(self.norm1, x_n)
        x = self.add_norm1(x_res, run(self.gamma_1, run(self.self_attn, q, q, q)))
        x_res, x_n = ops.fork2(ops.real(x))
        x = self.add_norm2(x_res, run(self.gamma_2, self._ff_block(run(self.norm2, x_n))))
        return self._post(x)


class CrossTransformerEncoderLayer(_EncoderLayerBase):
    """pre-norm cross-attention layer: queries from one branch, keys / values from the other (htdemucsq.py:224-328)"""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation=None, layer_norm_eps=1e-5, layer_scale=False,
                 init_values=1e-4, norm_first=False, group_norm=False, norm_out=False, sparse=False, batch_first=False, gelu=True, **unused):
        super().__init__()
        if sparse:
            raise NotImplementedError("sparse attention is not used by the FQSS HTDemucs")
        self.cross_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=batch_first)
        _layer_members(self, d_model, nhead, dim_feedforward, dropout, gelu, layer_norm_eps, layer_scale, init_values, norm_first, group_norm,
                       norm_out, ("norm1", "norm2", "norm3"))

    def forward(self, q, k, mask=None):
        q_res, q_n = ops.fork2(ops.real(q))
        kn = run(self.norm2, ops.real(k))
        x = self.add_norm1(q_res, run(self.gamma_1, run(self.cross_attn, run(self.norm1, q_n), kn, kn)))
        x_res, x_n = ops.fork2(ops.real(x))
        x = self.add_norm2(x_res, run(self.gamma_2, self._ff_block(run(self.norm3, x_n))))
        return self._post(x)


class CrossTransformerEncoder(nn.Module):
    def __init__(self, dim, emb="sin", hidden_scale=4.0, num_heads=8, num_layers=6, cross_first=False, dropout=0.0, max_positions=1000,
                 norm_in=True, norm_in_group=False, group_norm=False, norm_first=False, norm_out=False, max_period=10000.0,
                 weight_decay=0.0, lr=None, layer_scale=False, gelu=True, sin_random_shift=0, weight_pos_embed=1.0, sparse_self_attn=False,
                 sparse_cross_attn=False, **unused):
        super().__init__()
        assert dim % num_heads == 0
        if emb != "sin" or sin_random_shift != 0:
            raise NotImplementedError("only the deterministic sinusoidal embedding (t_emb='sin', no random shift) is built")
        hidden_dim = int(dim * hidden_scale)
        self.num_layers = num_layers
        self.classic_parity = 1 if cross_first else 0
        self.emb, self.max_period, self.weight_decay, self.weight_pos_embed, self.sin_random_shift, self.lr = \
            emb, max_period, weight_decay, weight_pos_embed, sin_random_shift, lr
        if norm_in:
            self.norm_in, self.norm_in_t = nn.LayerNorm(dim), nn.LayerNorm(dim)
        elif norm_in_group:
            self.norm_in, self.norm_in_t = MyGroupNorm(int(norm_in_group), dim), MyGroupNorm(int(norm_in_group), dim)
        else:
            self.norm_in, self.norm_in_t = nn.Identity(), nn.Identity()
        kw = dict(d_model=dim, nhead=num_heads, dim_feedforward=hidden_dim, dropout=dropout, gelu=gelu, group_norm=group_norm,
                  norm_first=norm_first, norm_out=norm_out, layer_scale=layer_scale, batch_first=True)
        self.layers = nn.ModuleList()
        self.layers_t = nn.ModuleList()
        for idx in range(num_layers):
            if idx % 2 == self.classic_parity:
                self.layers.append(MyTransformerEncoderLayer(sparse=sparse_self_attn, **kw))
                self.layers_t.append(MyTransformerEncoderLayer(sparse=sparse_self_attn, **kw))
            else:
                self.layers.append(CrossTransformerEncoderLayer(sparse=sparse_cross_attn, **kw))
                self.layers_t.append(CrossTransformerEncoderLayer(sparse=sparse_cross_attn, **kw))
        self.add_x = Add()
        self.add_xt = Add()
        self.const_pos_emb_2d = Const()
        self.const_pos_emb = Const()
        self._tables = {}

    def _table(self, key, make, device):
        t = self._tables.get((key, str(device)))
        if t is None:
            t = self._tables[(key, str(device))] = make().to(device).contiguous()
        return t

    def _add_pos(self, add, x, pos):
        """x [B, L, C] + pos [1, L, C] (batch broadcast)"""
        B, L, C = x.shape
        if self.weight_pos_embed != 1.0:
            pos = K.axpby(pos, pos, 0.0, sa=float(self.weight_pos_embed))
        if B == 1:
            return add(x, pos.reshape(1, L, C))
        return ops.real(add(x.reshape(1, B, L * C), ops.real(pos).reshape(1, 1, L * C))).reshape(B, L, C)

    def forward(self, x, xt):
        x, xt = ops.real(x), ops.real(xt)
        B, C, Fr, T1 = x.shape
        # token order of the reference: "b c fr t1 -> b (t1 fr) c"
        pe2 = self._table(("2d", C, Fr, T1), lambda: create_2d_sin_embedding(C, Fr, T1, "cpu", self.max_period)
                          .permute(0, 3, 2, 1).reshape(T1 * Fr, C), x.device)
        pos_emb_2d = run(self.const_pos_emb_2d, pe2)
        xr = _Transpose.apply(x.reshape(B, C, Fr * T1))                          # [B, (fr t1), C]
        xr = ops_dp.Permute4.apply(xr, (B, T1, Fr), (Fr * T1 * C, C, T1 * C), (B, Fr, T1), (Fr * T1 * C, C, Fr * C)).reshape(B, T1 * Fr, C)
        xr = run(self.norm_in, xr)
        xr = self._add_pos(self.add_x, ops.real(xr), pos_emb_2d)
        B, C, T2 = xt.shape
        xtr = _Transpose.apply(xt.contiguous())                                  # [B, T2, C]
        pe1 = self._table(("1d", C, T2), lambda: create_sin_embedding(T2, C, 0, "cpu", self.max_period).reshape(T2, C).float(), x.device)
        pos_emb = run(self.const_pos_emb, pe1)
        xtr = run(self.norm_in_t, xtr)
        xtr = self._add_pos(self.add_xt, ops.real(xtr), pos_emb)
        for idx in range(self.num_layers):
            if idx % 2 == self.classic_parity:
                xr = self.layers[idx](xr)
                xtr = self.layers_t[idx](xtr)
            else:
                xa, xb = ops.fork2(ops.real(xr))
                ta, tb = ops.fork2(ops.real(xtr))
                xr = self.layers[idx](xa, ta)
                xtr = self.layers_t[idx](tb, xb)
        xr = ops.real(xr).reshape(B, T1, Fr, C)
        xo = ops_dp.Permute4.apply(xr.contiguous(), (B, Fr, T1), (T1 * Fr * C, C, Fr * C), (B, T1, Fr), (T1 * Fr * C, C, T1 * C)).reshape(B, Fr * T1, C)
        xo = _Transpose.apply(xo).reshape(B, C, Fr, T1)
        xto = _Transpose.apply(ops.real(xtr).contiguous())
        return xo, xto


# ----------------------------------------------------------------------------------------------------------------------
# spectrogram pair and the per-sample normalisation as autograd nodes
# ----------------------------------------------------------------------------------------------------------------------
class _ISpec(torch.autograd.Function):
    """x [B, S, 2*C, Fr, T] ("complex as channels": channel = 2*c + (re, im)) -> waveform [B, S, C, length] (`_mask` + `_ispec`)"""

    @staticmethod
    def forward(ctx, x, nfft, hop, length):
        B, S, C2, Fr, T = x.shape
        assert Fr == nfft // 2 and C2 % 2 == 0
        ctx.cfg = (nfft, hop, length, tuple(x.shape))
        z = K.transpose2d(x.reshape(B * S * (C2 // 2), 2, Fr, T))           # [rows, 2, T, Fr]
        y = K.istft(z, nfft, hop, hop // 2 * 3, length)
        return y.reshape(B, S, C2 // 2, length)

    @staticmethod
    def backward(ctx, g):
        nfft, hop, length, shape = ctx.cfg
        B, S, C2, Fr, T = shape
        gz = K.istft_bwd(g.reshape(B * S * (C2 // 2), length), nfft, hop, hop // 2 * 3, T)
        return K.transpose2d(gz).reshape(shape), None, None, None


class _Denorm(torch.autograd.Function):
    """x * std_b + mean_b (htdemucsq.py:1034-1035); mean / std come from the input mixture: constants of the graph"""

    @staticmethod
    def forward(ctx, x, ms):
        ctx.save_for_backward(ms)
        return K.sample_norm(x, ms, True)

    @staticmethod
    def backward(ctx, g):
        (ms,) = ctx.saved_tensors
        ms0 = ms.clone()
        ms0[:, 0] = 0.0
        return K.sample_norm(g, ms0, True), None


def rescale_conv(conv, reference):
    std = conv.weight.std().detach()
    scale = (std / reference) ** 0.5
    conv.weight.data /= scale
    if conv.bias is not None:
        conv.bias.data /= scale


def rescale_module(module, reference):
    """weight rescaling trick at construction (demucsq.py:95-107)"""
    for sub in module.modules():
        if isinstance(sub, (nn.Conv1d, nn.ConvTranspose1d, nn.Conv2d, nn.ConvTranspose2d)):
            rescale_conv(sub, reference)


class HTDemucsQ(nn.Module):
    def __init__(self, sources, audio_channels=2, channels=48, channels_time=None, growth=2, nfft=4096, wiener_iters=0, end_iters=0,
                 wiener_residual=False, cac=True, depth=4, rewrite=True, multi_freqs=None, multi_freqs_depth=3, freq_emb=0.2, emb_scale=10,
                 emb_smooth=True, kernel_size=8, time_stride=2, stride=4, context=1, context_enc=0, norm_starts=4, norm_groups=4,
                 dconv_mode=1, dconv_depth=2, dconv_comp=8, dconv_init=1e-3, bottom_channels=0, t_layers=5, t_emb="sin",
                 t_hidden_scale=4.0, t_heads=8, t_dropout=0.0, t_max_positions=10000, t_norm_in=True, t_norm_in_group=False,
                 t_group_norm=False, t_norm_first=True, t_norm_out=True, t_max_period=10000.0, t_weight_decay=0.0, t_lr=None,
                 t_layer_scale=True, t_gelu=True, t_weight_pos_embed=1.0, t_sin_random_shift=0, t_cape_mean_normalize=True,
                 t_cape_augment=True, t_cape_glob_loc_scale=[5000.0, 1.0, 1.4], t_sparse_self_attn=False, t_sparse_cross_attn=False,
                 t_mask_type="diag", t_mask_random_seed=42, t_sparse_attn_window=500, t_global_window=100, t_sparsity=0.95,
                 t_auto_sparsity=False, t_cross_first=False, rescale=0.1, samplerate=44100, segment=10, use_train_segment=True):
        super().__init__()
        if not cac or wiener_iters or end_iters or multi_freqs:
            raise NotImplementedError("HTDemucsQ: only complex-as-channels without Wiener filtering / MultiWrap is built")
        self._init_args_kwargs = ((), dict(sources=sources, audio_channels=audio_channels, channels=channels, nfft=nfft, depth=depth,
                                           bottom_channels=bottom_channels, t_layers=t_layers, t_heads=t_heads))
        self.cac = cac
        self.set_splitter_combiner(1, 1)
        self.wiener_residual, self.audio_channels, self.sources, self.n_srcs = wiener_residual, audio_channels, sources, len(sources)
        self.kernel_size, self.context, self.stride, self.depth = kernel_size, context, stride, depth
        self.bottom_channels, self.channels, self.samplerate, self.segment = bottom_channels, channels, samplerate, segment
        self.use_train_segment, self.nfft, self.hop_length = use_train_segment, nfft, nfft // 4
        self.wiener_iters, self.end_iters = wiener_iters, end_iters
        self.freq_emb = None
        self.encoder, self.decoder, self.tencoder, self.tdecoder = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        chin = audio_channels
        chin_z = chin * 2
        chout = channels_time or channels
        chout_z = channels
        freqs = nfft // 2
        for index in range(depth):
            norm = index >= norm_starts
            freq = freqs > 1
            stri, ker = stride, kernel_size
            if not freq:
                assert freqs == 1
                ker, stri = time_stride * 2, time_stride
            pad, last_freq = True, False
            if freq and freqs <= kernel_size:
                ker, pad, last_freq = freqs, False, True
            kw = dict(kernel_size=ker, stride=stri, freq=freq, pad=pad, norm=norm, rewrite=rewrite, norm_groups=norm_groups,
                      dconv_kw=dict(depth=dconv_depth, compress=dconv_comp, init=dconv_init, gelu=True))
            kwt = dict(kw, freq=0, kernel_size=kernel_size, stride=stride, pad=True)
            if last_freq:
                chout_z = max(chout, chout_z)
                chout = chout_z
            self.encoder.append(HEncLayer(chin_z, chout_z, dconv=dconv_mode & 1, context=context_enc, **kw))
            if freq:
                self.tencoder.append(HEncLayer(chin, chout, dconv=dconv_mode & 1, context=context_enc, empty=last_freq, **kwt))
            if index == 0:
                chin = self.audio_channels * len(self.sources)
                chin_z = chin * 2
            self.decoder.insert(0, HDecLayer(chout_z, chin_z, dconv=dconv_mode & 2, last=index == 0, context=context, **kw))
            if freq:
                self.tdecoder.insert(0, HDecLayer(chout, chin, dconv=dconv_mode & 2, empty=last_freq, last=index == 0, context=context, **kwt))
            chin, chin_z = chout, chout_z
            chout, chout_z = int(growth * chout), int(growth * chout_z)
            if freq:
                freqs = 1 if freqs <= kernel_size else freqs // stride
            if index == 0 and freq_emb:
                self.freq_emb = ScaledEmbedding(freqs, chin_z, smooth=emb_smooth, scale=emb_scale)
                self.freq_emb_scale = freq_emb
                self.add_freq = Add()
                self.mul_freq = Mul()
        if rescale:
            rescale_module(self, reference=rescale)
        transformer_channels = channels * growth ** (depth - 1)
        if bottom_channels:
            self.channel_upsampler = nn.Conv1d(transformer_channels, bottom_channels, 1)
            self.channel_downsampler = nn.Conv1d(bottom_channels, transformer_channels, 1)
            self.channel_upsampler_t = nn.Conv1d(transformer_channels, bottom_channels, 1)
            self.channel_downsampler_t = nn.Conv1d(bottom_channels, transformer_channels, 1)
            transformer_channels = bottom_channels
        if t_layers > 0:
            self.crosstransformer = CrossTransformerEncoder(
                dim=transformer_channels, emb=t_emb, hidden_scale=t_hidden_scale, num_heads=t_heads, num_layers=t_layers,
                cross_first=t_cross_first, dropout=t_dropout, max_positions=t_max_positions, norm_in=t_norm_in,
                norm_in_group=t_norm_in_group, group_norm=t_group_norm, norm_first=t_norm_first, norm_out=t_norm_out,
                max_period=t_max_period, weight_decay=t_weight_decay, lr=t_lr, layer_scale=t_layer_scale, gelu=t_gelu,
                sin_random_shift=t_sin_random_shift, weight_pos_embed=t_weight_pos_embed, sparse_self_attn=t_sparse_self_attn,
                sparse_cross_attn=t_sparse_cross_attn)
        else:
            self.crosstransformer = None

    # ------------------------------------------------------------------------------------------------------------------
    def set_splitter_combiner(self, n_splitter, n_combiner):
        self.n_splitter = n_splitter
        self.n_combiner = n_combiner

    def valid_length(self, length):
        if not self.use_train_segment:
            return length
        training_length = int(self.segment * self.samplerate)
        if training_length < length:
            raise ValueError(f"Given length {length} is longer than training length {training_length}")
        return training_length

    def _spec(self, mix):
        """mix [B, C, L] -> "complex as channels" magnitude [B, 2*C, Fr, le]  (`_spec` + `_magnitude`, htdemucsq.py:931-971)"""
        B, C, L = mix.shape
        hl = self.hop_length
        le = int(math.ceil(L / hl))
        z = K.stft(mix.reshape(B * C, L), self.nfft, hl, le, hl // 2 * 3)       # [B*C, 2, le, Fr]
        return K.transpose2d(z).reshape(B, C * 2, self.nfft // 2, le)

    def pre_process(self, mix):
        self.length = mix.shape[-1]
        self.length_pre_pad = None
        if self.use_train_segment:
            if self.training:
                self.segment = Fraction(mix.shape[-1], self.samplerate)
            else:
                self.training_length = int(self.segment * self.samplerate)
                if mix.shape[-1] < self.training_length:
                    self.length_pre_pad = mix.shape[-1]
                    mix = pad_right(mix, self.training_length)
        with torch.no_grad():
            mix = mix.contiguous()
            mag = self._spec(mix)
            self.ms = K.sample_meanstd(mag)
            x = K.sample_norm(mag, self.ms, False)
            self.ms_t = K.sample_meanstd(mix)
            xt = K.sample_norm(mix, self.ms_t, False)
            x = preprocess(x, n_splitter=self.n_splitter)
            xt = preprocess(xt, n_splitter=self.n_splitter, normalize=False)
        return x, xt

    def post_process(self, x, xt):
        x = postprocess(x, n_combiner=self.n_combiner)              # [B, S, 2*C, Fq, T]
        xt = postprocess(xt, n_combiner=self.n_combiner)            # [B, S, C, L]
        x = _Denorm.apply(ops.real(x).contiguous(), self.ms)
        xt = _Denorm.apply(ops.real(xt).contiguous(), self.ms_t)
        length = self.length if (not self.use_train_segment or self.training) else self.training_length
        x = _ISpec.apply(x, self.nfft, self.hop_length, length)
        xt = xt.reshape(self.B, self.n_srcs, -1, length)
        y = _fadd(xt, x)
        if self.length_pre_pad:
            y = crop(y, -1, 0, self.length_pre_pad)
        return y

    def forward(self, mix):
        x, xt = self.pre_process(mix)
        self.B, C, Fq, T = x.shape
        saved, saved_t, lengths, lengths_t = [], [], [], []
        for idx, encode in enumerate(self.encoder):
            lengths.append(x.shape[-1])
            inject = None
            if idx < len(self.tencoder):
                lengths_t.append(xt.shape[-1])
                tenc = self.tencoder[idx]
                xt = tenc(xt)
                if not tenc.empty:
                    xt, keep = ops.fork2(ops.real(xt))
                    saved_t.append(keep)
                else:
                    inject = xt
            x = encode(x, inject)
            if idx == 0 and self.freq_emb is not None:
                frs = torch.arange(x.shape[-2], device=x.device)
                emb = self.freq_emb(frs)                                             # [Fr, C] (quantized table x scale)
                emb = run(self.mul_freq, emb, self.freq_emb_scale)
                x = self._add_freq_emb(x, emb)
            x, keep = ops.fork2(ops.real(x))
            saved.append(keep)
        if self.fqss_cuts[0]:
            # backward cut: everything above is the LAST backward segment; the U-Net skips jump over the segments in between and
            # are pushed with it (late cuts)
            x, xt = ops.cut(x, xt)
            saved = list(ops.cut(*saved, late=True))
            saved_t = list(ops.cut(*saved_t, late=True))
        if self.crosstransformer:
            if self.bottom_channels:
                b, c, f, t = x.shape
                x = ops.real(run(self.channel_upsampler, ops.real(x).reshape(b, c, f * t))).reshape(b, -1, f, t)
                xt = run(self.channel_upsampler_t, xt)
            x, xt = self.crosstransformer(x, xt)
            if self.bottom_channels:
                x = ops.real(run(self.channel_downsampler, ops.real(x).reshape(b, -1, f * t))).reshape(b, c, f, t)
                xt = run(self.channel_downsampler_t, xt)
        if self.fqss_cuts[1]:
            x, xt = ops.cut(x, xt)
            if not self.fqss_cuts[0]:
                saved = list(ops.cut(*saved, late=True))
                saved_t = list(ops.cut(*saved_t, late=True))
        for idx, decode in enumerate(self.decoder):
            skip = saved.pop(-1)
            x, pre = decode(x, skip, lengths.pop(-1))
            offset = self.depth - len(self.tdecoder)
            if idx >= offset:
                tdec = self.tdecoder[idx - offset]
                length_t = lengths_t.pop(-1)
                if tdec.empty:
                    assert pre.shape[2] == 1, pre.shape
                    xt, _ = tdec(pre[:, :, 0], None, length_t)
                else:
                    xt, _ = tdec(xt, saved_t.pop(-1), length_t)
        assert len(saved) == 0 and len(lengths_t) == 0 and len(saved_t) == 0
        x = ops.real(x).reshape(self.n_combiner, self.B, self.n_srcs, -1, Fq, T)
        xt = ops.real(xt).reshape(self.n_combiner, self.B, self.n_srcs, -1, xt.shape[-1])
        return self.post_process(x, xt)

    fqss_cuts = (False, False)      # backward cut points (encoders | cross-transformer | decoders), set by fqss_segments()

    def fqss_segments(self, n):
        """Backward segments = gradient buckets (runtime.KDTrainStep, see ConvTasNetQ.fqss_segments): both encoder stacks, the
        cross-transformer with its channel up / down samplers, both decoder stacks (n >= 3); encoders + transformer | decoders (n = 2).
        Parameters outside these modules stay with the first segment (runtime: the last one whose gradients become final)."""
        n = int(n)
        xf = [m for m in (getattr(self, "channel_upsampler", None), getattr(self, "channel_upsampler_t", None), self.crosstransformer,
                          getattr(self, "channel_downsampler", None), getattr(self, "channel_downsampler_t", None))
              if isinstance(m, nn.Module)]
        enc = [self.encoder, self.tencoder] + [m for m in (self.freq_emb, getattr(self, "mul_freq", None), getattr(self, "add_freq", None))
                                               if isinstance(m, nn.Module)]
        dec = [self.decoder, self.tdecoder]
        if n >= 3 and xf:
            self.fqss_cuts = (True, True)
            return [enc, xf, dec]
        if n >= 2:
            self.fqss_cuts = (False, True)
            return [enc + xf, dec]
        self.fqss_cuts = (False, False)
        return [[self]]

    def _add_freq_emb(self, x, emb):
        """add_freq(x, emb.t()[None, :, :, None].expand_as(x))  (htdemucsq.py:1063-1068): the [Fr, C] table is added along batch and
        time by the per-channel add kernel (channel = c*Fr + fr); a quantized `add_freq` applies its quantizer to the sum"""
        x = ops.real(x)
        B, C, Fr, T = x.shape
        table = _Transpose.apply(ops.real(emb).contiguous()).reshape(C * Fr)       # [C, Fr]
        y = ops_dp.ChanAdd.apply(x.reshape(B, C * Fr, T), table).reshape(B, C, Fr, T)
        if isinstance(self.add_freq, QL.AddQ):
            return QL.fq_node(self.add_freq.activation_fake_quantize, y)
        return y

    # ------------------------------------------------------------------------------------------------------------------
    def quantize_model(self, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, inout_nl_quant=False,
                       in_quant=False, in_act_n_bits=8, out_quant=True, out_act_n_bits=8):
        """the reference's rewrite table (htdemucsq.py:1157-1245), path for path"""
        p = dict(gradient_based=gradient_based, act_quant=act_quant, weight_quant=weight_quant, weight_n_bits=weight_n_bits,
                 act_n_bits=act_n_bits)
        enc_p = dict(n_splitter=self.n_splitter, gradient_based=gradient_based, act_quant=act_quant, inout_nl_quant=inout_nl_quant,
                     weight_quant=weight_quant, in_quant=in_quant, weight_n_bits=weight_n_bits, act_n_bits=act_n_bits,
                     in_act_n_bits=in_act_n_bits)
        for n, m in list(self.named_modules()):
            if type(m) is HEncLayer:
                if n in ("encoder.0", "tencoder.0"):
                    replace_encoderq(m, ["conv", "gelu"], enc_p)
                else:
                    quantize_modules(m, ["conv", "gelu"], p)
                quantize_modules(m, ["rewrite", "glu"], p)
            elif type(m) is HDecLayer:
                quantize_modules(m, ["rewrite", "glu"], p)
                quantize_modules(m, ["add"], p)
                if m.last:
                    replace_decoderq(m, ["conv_tr"], dict(n_combiner=self.n_combiner, gradient_based=gradient_based, act_quant=act_quant,
                                                          inout_nl_quant=inout_nl_quant, act_n_bits=out_act_n_bits, out_quant=out_quant,
                                                          out_act_n_bits=out_act_n_bits, weight_quant=weight_quant,
                                                          weight_n_bits=weight_n_bits, train_res_dec=n in ["decoder.3"]))
                else:
                    quantize_modules(m, ["conv_tr", "gelu"], p)
            elif type(m) is HTDemucsQ:
                if self.bottom_channels:
                    for name in ("channel_downsampler", "channel_downsampler_t", "channel_upsampler", "channel_upsampler_t"):
                        quantize_modules(m, [name], p)
                quantize_modules(m, ["add_freq"], p)
                quantize_modules(m, ["mul_freq"], p)
                quantize_modules(m.freq_emb, ["embedding"], p)
                quantize_modules(m.freq_emb, ["mul"], p)
            elif type(m) is DConv:
                for layer in m.layers:
                    quantize_modules(layer, ["0", "1", "2"], p)
                    quantize_modules(layer, ["3", "4", "5"], p)
                    quantize_modules(layer[6], ["mul"], p)
                for i in range(len(m.adds)):
                    quantize_modules(m.adds, [str(i)], p)
            elif type(m) is CrossTransformerEncoder:
                for name in ("norm_in", "norm_in_t", "add_x", "add_xt", "const_pos_emb", "const_pos_emb_2d"):
                    quantize_modules(m, [name], p)
            elif type(m) in (CrossTransformerEncoderLayer, MyTransformerEncoderLayer):
                cross = type(m) is CrossTransformerEncoderLayer
                for name in ("add_norm1", "add_norm2", "norm1", "norm2") + (("norm3",) if cross else ()):
                    quantize_modules(m, [name], p)
                quantize_modules(m.norm_out, ["const"], p)
                quantize_modules(m.gamma_1, ["mul"], p)
                quantize_modules(m.gamma_2, ["mul"], p)
                quantize_modules(m, ["linear1", "activation"], p)
                quantize_modules(m, ["linear2"], p)
                quantize_modules(m, ["cross_attn" if cross else "self_attn"], p)
