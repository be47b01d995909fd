"""oracle/dptnet_oracle.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (torch fp32 ATen ops + autograd) of the DPTNet W8A8 QAT step (cfg 3), written
functionally over a flat ``{state_dict key: tensor}`` table like oracle/fqss_oracle.py.  Only tests/,
__graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import it.

Reference followed (ssi-research/FQSS @ 2024_10_08):
  layer_norm_q        quantization/qat/qat_layers.py:455-465   LayerNormQ
  linear_q            quantization/qat/qat_layers.py:521-536   LinearQ
  lstm_q              quantization/qat/qat_layers.py:571-600   LSTMQ: torch's fused LSTM on fake-quantized weights, zero
                      initial state, output fake-quant only; the cell (i, f, g, o order, c = f*c + i*g, h = o*tanh(c)) is
                      restated step by step instead of calling _VF.lstm
  mha_q               quantization/qat/qat_layers.py:865-950   MultiheadAttentionQ.forward incl. its quirks: q/k/v are three
                      separately quantized copies of the SAME full in-projection, `attn - fq(attn)` statements are no-ops
                      (their quantizers only observe), no dropout
  conv2d_q            quantization/qat/qat_layers.py  Conv2dQ (1x1)
  linear_decoder      quantization/qat/qat_layers.py:1256-1296 + the nn.Linear branch of ResidualErrorBlock :1178-1187
  overlap_and_add     quantization/qat/models/dptnetq.py:17-58
  split/merge_feature quantization/qat/models/dptnetq.py:232-276
  transformer layer   quantization/qat/models/dptnetq.py:84-97 (attention -> add -> norm -> LSTM -> ReLU -> linear -> add -> norm)
  DPT / BF_module     quantization/qat/models/dptnetq.py:189-209, 290-309
  DPTNetQ.forward     quantization/qat/models/dptnetq.py:368-409; quantize_model :430-478 decides which ops carry quantizers
  teacher             the same graph with float ops (train_utils.py:25), n_splitter = n_combiner = 1

Pinned by tests/golden/dpt_layers.npz, dpt_tiny_step.npz, cfg3_step.npz (tools/make_goldens_dptnet.py, produced by the
imported reference); the loss / PIT / trainer are those of oracle/fqss_oracle.py (same asteroid env).
"""
import math

import torch
import torch.nn.functional as F

from .fqss_oracle import ActRange, QTable, WeightRange, combine, split

EPS = 1e-8


# ------------------------------------------------------------------------------------------------------------------
# data movement
# ------------------------------------------------------------------------------------------------------------------
def overlap_and_add(sig, step):
    """[..., frames, L] -> [..., (frames-1)*step + L]   (L a multiple of step, as everywhere in the model)"""
    frames, L = sig.shape[-2:]
    assert L % step == 0
    n = L // step
    lead = sig.shape[:-2]
    out = sig.new_zeros(*lead, frames - 1 + n, step)
    parts = sig.reshape(*lead, frames, n, step)
    for j in range(n):
        out[..., j:j + frames, :] = out[..., j:j + frames, :] + parts[..., j, :]
    return out.reshape(*lead, -1)


def pad_segment(x, K):
    B, D, T = x.shape
    P = K // 2
    rest = K - (P + T % K) % K
    if rest > 0:
        x = torch.cat([x, x.new_zeros(B, D, rest)], 2)
    z = x.new_zeros(B, D, P)
    return torch.cat([z, x, z], 2), rest


def split_feature(x, K):
    """[B, N, T] -> [B, N, K, S] 50 %-overlapped chunks"""
    x, rest = pad_segment(x, K)
    B, D, _ = x.shape
    P = K // 2
    s1 = x[:, :, :-P].contiguous().view(B, D, -1, K)
    s2 = x[:, :, P:].contiguous().view(B, D, -1, K)
    seg = torch.cat([s1, s2], 3).view(B, D, -1, K).transpose(2, 3)
    return seg.contiguous(), rest


def merge_halves(x):
    """[B, N, K, S] -> the two streams whose sum is the merged signal (before the rest is cut)"""
    B, D, K, _ = x.shape
    P = K // 2
    x = x.transpose(2, 3).contiguous().view(B, D, -1, K * 2)
    a = x[:, :, :, :K].contiguous().view(B, D, -1)[:, :, P:]
    b = x[:, :, :, K:].contiguous().view(B, D, -1)[:, :, :-P]
    return a, b


# ------------------------------------------------------------------------------------------------------------------
# LSTM cell, restated
# ------------------------------------------------------------------------------------------------------------------
def lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [S, B, I] -> h [S, B, H], zero initial state"""
    S, B, _ = x.shape
    H = w_hh.shape[1]
    pre = F.linear(x, w_ih, b_ih)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = [None] * S
    for t in (range(S - 1, -1, -1) if reverse else range(S)):
        g = pre[t] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs)


def lstm_bidir(x, W):
    """W: dict name -> tensor with torch's flat-weight names"""
    fw = lstm_dir(x, W["weight_ih_l0"], W["weight_hh_l0"], W["bias_ih_l0"], W["bias_hh_l0"], False)
    bw = lstm_dir(x, W["weight_ih_l0_reverse"], W["weight_hh_l0_reverse"], W["bias_ih_l0_reverse"], W["bias_hh_l0_reverse"], True)
    return torch.cat([fw, bw], -1)


def mha_core(q, k, v, nhead):
    """q,k,v: [L, B, E] (already projected); returns heads [L, B, E] (before out_proj); plain float attention"""
    L, B, E = q.shape
    hd = E // nhead
    qh = q.reshape(L, B * nhead, hd).permute(1, 0, 2) / math.sqrt(hd)
    kh = k.reshape(L, B * nhead, hd).permute(1, 0, 2)
    vh = v.reshape(L, B * nhead, hd).permute(1, 0, 2)
    a = torch.softmax(torch.bmm(qh, kh.transpose(-2, -1)), dim=-1)
    return torch.bmm(a, vh).transpose(1, 0).reshape(L, B, E)


# ------------------------------------------------------------------------------------------------------------------
NL_MAPS = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": F.relu, "gelu": F.gelu, "glu": (lambda t: F.glu(t, 1)),
           None: (lambda t: t)}


class DQTable(QTable):
    """QTable whose quantizer discovery also knows the dual-path layers' extra quantizers: any range parameter of rank >= 2
    is a per-channel weight range (axis 0, or 1 for transposed convs), rank-1 ranges are activation ranges."""

    def __init__(self, state_dict):
        self.p = {k: v.detach().clone().float().requires_grad_(True) for k, v in state_dict.items()}
        self.aq, self.wq = {}, {}
        for k in self.p:
            if not k.endswith(".min_range"):
                continue
            pre = k[: -len(".min_range")]
            if self.p[k].dim() >= 2:
                owner = pre.rsplit(".", 1)[0] + "."
                # ch_out_idx = 1 for transposed convs: the decoder (qat_layers.py:1317) and, with train_res_dec, the residual
                # decoder's own quantizer (qat_layers.py:1141-1145)
                axis = 1 if (((owner + "convTr1d.weight") in self.p or (owner + "convTr2d.weight") in self.p) and pre.endswith(".weight_fake_quantize")) or \
                    (pre.endswith(".weight_fake_quantize_dec") and self.p.get(owner + "residual_decoder.weight", torch.zeros(1)).dim() >= 3) else 0
                self.wq[pre] = WeightRange(self.p, pre, axis)
            else:
                self.aq[pre] = ActRange(self.p, pre)

    def A(self, full, x):
        return self.aq[full](x)

    def Wq(self, full, w):
        return self.wq[full](w)

    # -- the LayerQ forwards the dual-path models add --------------------------------------------------------------
    def layer_norm_q(self, name, x, eps=1e-5):
        y = F.layer_norm(x, x.shape[-1:], self.p[name + ".layernorm.weight"], self.p[name + ".layernorm.bias"], eps)
        return self._A(name, y)

    def linear_q(self, name, x):
        w = self._W(name, self.p[name + ".linear.weight"])
        return self._A(name, F.linear(x, w, self.p.get(name + ".linear.bias")))

    def lstm_q(self, name, x):
        W = {}
        for n in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse"):
            W[n] = self.Wq(f"{name}.weight_quantizers_dict.{n}", self.p[f"{name}.lstm.{n}"])
        for n in ("bias_ih_l0", "bias_hh_l0", "bias_ih_l0_reverse", "bias_hh_l0_reverse"):
            W[n] = self.p[f"{name}.lstm.{n}"]
        return self._A(name, lstm_bidir(x, W))

    def mha_q(self, name, x, nhead=4, key=None, batch_first=False):
        """MultiheadAttentionQ.forward (qat_layers.py:878-946); key (= value) None: self-attention.  Each of the q / k / v quantizers
        runs on the FULL [.., 3E] projection of its input and only its own third is used"""
        p = self.p
        Wi = self.Wq(name + ".weight_fake_quantize_in", p[name + ".mha.in_proj_weight"])
        Wo = self.Wq(name + ".weight_fake_quantize_out", p[name + ".mha.out_proj.weight"])
        key = x if key is None else key
        if batch_first:
            x, key = x.transpose(1, 0), key.transpose(1, 0)
        L, B, E = x.shape
        Lk = key.shape[0]
        hd = E // nhead
        Xq = F.linear(x, Wi, p[name + ".mha.in_proj_bias"])
        Xk = Xq if key is x else F.linear(key, Wi, p[name + ".mha.in_proj_bias"])
        Q = self.A(name + ".activation_fake_quantize_q", Xq)[..., :E]
        K = self.A(name + ".activation_fake_quantize_k", Xk)[..., E:2 * E]
        V = self.A(name + ".activation_fake_quantize_v", Xk)[..., 2 * E:]
        q = Q.reshape(L, B * nhead, hd).permute(1, 0, 2)
        k = K.reshape(Lk, B * nhead, hd).permute(1, 0, 2)
        v = V.reshape(Lk, B * nhead, hd).permute(1, 0, 2)
        q = self.A(name + ".activation_fake_quantize_div", q / math.sqrt(hd))
        attn = torch.bmm(q, k.transpose(-2, -1))
        self.A(name + ".activation_fake_quantize_attn", attn.detach())       # result discarded in the reference (:907)
        attn = torch.softmax(attn, dim=-1)
        self.A(name + ".activation_fake_quantize_softmax", attn.detach())    # result discarded (:909)
        heads = self.A(name + ".activation_fake_quantize_head", torch.bmm(attn, v))
        y = F.linear(heads.transpose(1, 0).reshape(L * B, E), Wo, p[name + ".mha.out_proj.bias"]).reshape(L, B, E)
        if batch_first:
            y = y.transpose(1, 0)
        return self._A(name, y)

    def conv2d_q(self, name, x):
        w = self._W(name, self.p[name + ".conv2d.weight"])
        return self._A(name, F.conv2d(x, w, self.p.get(name + ".conv2d.bias")))

    def conv1d_nl_q(self, name, x, nl, **geom):
        """Conv1dQ / Conv1dNlQ (qat_layers.py:124-153); geom = stride / padding / dilation of the wrapped nn.Conv1d"""
        w = self._W(name, self.p[name + ".conv1d.weight"])
        y = F.conv1d(x, w, self.p.get(name + ".conv1d.bias"), **geom)
        return self._A(name, NL_MAPS[nl](y))

    def conv1d_gn_nl_q(self, name, x, nl, groups=1, eps=1e-5, **geom):
        """Conv1dGnNlQ (qat_layers.py:222-259): fq(nl(GroupNorm(conv1d(x, fq_w(W))))), one activation quantizer at the end"""
        w = self._W(name, self.p[name + ".conv1d.weight"])
        y = F.conv1d(x, w, self.p.get(name + ".conv1d.bias"), **geom)
        y = F.group_norm(y, groups, self.p[name + ".gn.weight"], self.p[name + ".gn.bias"], eps)
        return self._A(name, NL_MAPS[nl](y))

    def conv2d_nl_q(self, name, x, nl, **geom):
        """Conv2dQ / Conv2dNlQ (qat_layers.py:156-186, 261-293)"""
        w = self._W(name, self.p[name + ".conv2d.weight"])
        return self._A(name, NL_MAPS[nl](F.conv2d(x, w, self.p.get(name + ".conv2d.bias"), **geom)))

    def convtr_nl_q(self, name, x, nl, **geom):
        """ConvTranspose1d/2d(Nl)Q (qat_layers.py:296-435): weight ranges per OUTPUT channel = dim 1 of the [Ci, Co, k..] weight"""
        key = name + (".convTr1d" if x.dim() == 3 else ".convTr2d")
        w = self._W(name, self.p[key + ".weight"])
        fn = F.conv_transpose1d if x.dim() == 3 else F.conv_transpose2d
        return self._A(name, NL_MAPS[nl](fn(x, w, self.p.get(key + ".bias"), **geom)))

    def convtr_decoder_q(self, name, x, n_combiner=2, train_res_dec=False, **geom):
        """ConvTr1dDecoderQ / ConvTr2dDecoderQ with a general geometry (qat_layers.py:1305-1418) and the transposed-conv branches of
        ResidualErrorBlock (:1189-1216).  Quirks kept: the residual encoder is called with the stride only; the 1-D decode of
        the residual drops the bias, the 2-D decode uses `residual_decoder.bias`"""
        p, two_d = self.p, x.dim() == 4
        key = name + (".convTr2d" if two_d else ".convTr1d")
        convT, conv = (F.conv_transpose2d, F.conv2d) if two_d else (F.conv_transpose1d, F.conv1d)
        w = self._W(name, p[key + ".weight"])
        y = self._A(name, convT(x, w, p.get(key + ".bias"), **geom))
        if n_combiner == 1:
            return y
        rb = name + ".residual_error_block"
        w_res = self._W(rb, p[rb + ".residual_encoder.weight"])
        Y1 = self._A(rb, x - conv(y, w_res, p.get(rb + ".residual_encoder.bias"), stride=geom.get("stride", 1)))
        wd = self.Wq(rb + ".weight_fake_quantize_dec", p[rb + ".residual_decoder.weight"]) if train_res_dec else w
        y1 = convT(Y1, wd, p[rb + ".residual_decoder.bias"] if two_d else None, **geom)
        return torch.stack([y, self.aq[name + ".activation_fake_quantize_residual"](y1)])

    # -- first layers of cfg 5 (HTDemucs) -----------------------------------------------------------
    def linear_nl_q(self, name, x, nl):
        """LinearNlQ (qat_layers.py:539-561)"""
        w = self._W(name, self.p[name + ".linear.weight"])
        y = F.linear(x, w, self.p.get(name + ".linear.bias"))
        return self._A(name, {"relu": F.relu, "gelu": F.gelu}[nl](y))

    def div_q(self, name, a, b):
        """DivQ (qat_layers.py:104-113)"""
        return self._A(name, torch.div(a, b))

    def embedding_q(self, name, idx):
        """EmbeddingQ (qat_layers.py:490-508): lookup in the per-row fake-quantized table"""
        return self._A(name, F.embedding(idx, self._W(name, self.p[name + ".embedding.weight"])))

    def linear_decoder_q(self, name, x, n_combiner=2):
        w = self._W(name, self.p[name + ".linear.weight"])
        y0 = self._A(name, F.linear(x, w, None))
        if n_combiner == 1:
            return y0
        rb = name + ".residual_error_block"
        w_res = self._W(rb, self.p[rb + ".residual_encoder.weight"])
        Y1 = self._A(rb, x - F.linear(y0, w_res, None))
        y1 = self.aq[name + ".activation_fake_quantize_residual"](F.linear(Y1, w, None))
        return torch.stack([y0, y1])


class StudentDPTNetQ(DQTable):
    """W8A8 fake-quantized DPTNet over a flat parameter table with the reference's state_dict key names."""

    def __init__(self, state_dict, n_src=2, kernel_size=2, segment_size=250, n_splitter=2, n_combiner=2, nhead=4):
        super().__init__(state_dict)
        self.n_src, self.W, self.K = n_src, kernel_size, segment_size
        self.n_splitter, self.n_combiner, self.nhead = n_splitter, n_combiner, nhead
        self.layers = 1 + max(int(k.split(".")[3]) for k in self.p if k.startswith("separator.DPT.row_transformer."))
        self.N = self.p["separator.BN.conv1d.weight"].shape[0]
        self.E = self.p["separator.BN.conv1d.weight"].shape[1]

    def transformer(self, name, src):
        """src [L, B', N] seq-first (dptnetq.py:84-97)"""
        t = name + ".transformer"
        src2 = self.mha_q(t + ".self_attn", src, self.nhead)
        src = self._A(t + ".add_norm1", src + src2)
        src = self.layer_norm_q(t + ".norm1", src)
        src2 = self.linear_q(t + ".linear", F.relu(self.lstm_q(t + ".lstm", src)))
        src = self._A(t + ".add_norm2", src + src2)
        return self.layer_norm_q(t + ".norm2", src)

    def dpt(self, x):
        B, N, d1, d2 = x.shape
        out = x
        for i in range(self.layers):
            r = out.permute(0, 3, 2, 1).contiguous().view(B * d2, d1, N)
            r = self.transformer(f"separator.DPT.row_transformer.{i}", r.permute(1, 0, 2).contiguous()).permute(1, 0, 2).contiguous()
            out = r.view(B, d2, d1, N).permute(0, 3, 2, 1).contiguous()
            c = out.permute(0, 2, 3, 1).contiguous().view(B * d1, d2, N)
            c = self.transformer(f"separator.DPT.col_transformer.{i}", c.permute(1, 0, 2).contiguous()).permute(1, 0, 2).contiguous()
            out = c.view(B, d1, d2, N).permute(0, 3, 1, 2).contiguous()
        out = self._A("separator.DPT.output.0", F.prelu(out, self.p["separator.DPT.output.0.nl.weight"]))
        return self.conv2d_q("separator.DPT.output.1", out)

    def separator(self, x):
        B = x.shape[0]
        f = self._conv("separator.BN", x)
        seg, rest = split_feature(f, self.K)
        o = self.dpt(seg).view(B * self.n_src, self.N, self.K, -1)
        a, b = merge_halves(o)
        m = self._A("separator.add", a + b)
        if rest > 0:
            m = m[:, :, :-rest]
        m = m.contiguous()
        g = self._A("separator.mul", self.conv1d_nl_q("separator.output.0", m, "tanh") * self.conv1d_nl_q("separator.output_gate.0", m, "sigmoid"))
        return g.transpose(1, 2).contiguous().view(B, self.n_src, -1, self.N)

    def forward(self, x):
        x = split(x, self.n_splitter)
        B = x.shape[0]
        w = self._conv("encoder.conv1d_U", x, nl="relu")                       # [B, E, L]
        s = self._gn("enc_LN", w)
        s = self.separator(s)                                                  # [B, S, L, N]
        s = s.view(B * self.n_src, -1, self.N).transpose(1, 2).contiguous()
        m = self.conv1d_nl_q("mask_conv1x1.0", s, "relu").view(B, self.n_src, self.E, -1)
        sw = self._A("mul", w.unsqueeze(1) * m).transpose(2, 3)                # [B, S, L, E]
        dec = self.linear_decoder_q("decoder.basis_signals", sw, self.n_combiner)
        est = overlap_and_add(dec, self.W // 2)
        return combine(est.reshape(self.n_combiner, B, self.n_src, 1, -1), self.n_combiner)

    __call__ = forward


class TeacherDPTNet:
    """float copy of the same network (no splitter, no quantizers); keys are those of the un-quantized DPTNetQ"""

    def __init__(self, state_dict, n_src=2, kernel_size=2, segment_size=250, nhead=4):
        self.p = {k: v.detach().clone().float() for k, v in state_dict.items()}
        self.n_src, self.W, self.K, self.nhead = n_src, kernel_size, segment_size, nhead
        self.layers = 1 + max(int(k.split(".")[3]) for k in self.p if k.startswith("separator.DPT.row_transformer."))
        self.N, self.E = self.p["separator.BN.weight"].shape[:2]

    def transformer(self, t, src):
        p = self.p
        t = t + ".transformer"
        E = src.shape[-1]
        X = F.linear(src, p[t + ".self_attn.in_proj_weight"], p[t + ".self_attn.in_proj_bias"])
        heads = mha_core(X[..., :E], X[..., E:2 * E], X[..., 2 * E:], self.nhead)
        src = src + F.linear(heads, p[t + ".self_attn.out_proj.weight"], p[t + ".self_attn.out_proj.bias"])
        src = F.layer_norm(src, (E,), p[t + ".norm1.weight"], p[t + ".norm1.bias"], 1e-5)
        W = {n: p[f"{t}.lstm.{n}"] for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
                                             "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")}
        src = src + F.linear(F.relu(lstm_bidir(src, W)), p[t + ".linear.weight"], p[t + ".linear.bias"])
        return F.layer_norm(src, (E,), p[t + ".norm2.weight"], p[t + ".norm2.bias"], 1e-5)

    def forward(self, x):
        p = self.p
        if x.dim() == 2:
            x = x.unsqueeze(1)
        B = x.shape[0]
        w = F.relu(F.conv1d(x, p["encoder.conv1d_U.weight"], None, stride=self.W // 2))
        s = F.group_norm(w, 1, p["enc_LN.weight"], p["enc_LN.bias"], EPS)
        f = F.conv1d(s, p["separator.BN.weight"], None)
        seg, rest = split_feature(f, self.K)
        Bq, N, d1, d2 = seg.shape
        out = seg
        for i in range(self.layers):
            r = out.permute(0, 3, 2, 1).contiguous().view(B * d2, d1, N).permute(1, 0, 2).contiguous()
            r = self.transformer(f"separator.DPT.row_transformer.{i}", r).permute(1, 0, 2).contiguous()
            out = r.view(B, d2, d1, N).permute(0, 3, 2, 1).contiguous()
            c = out.permute(0, 2, 3, 1).contiguous().view(B * d1, d2, N).permute(1, 0, 2).contiguous()
            c = self.transformer(f"separator.DPT.col_transformer.{i}", c).permute(1, 0, 2).contiguous()
            out = c.view(B, d1, d2, N).permute(0, 3, 1, 2).contiguous()
        out = F.conv2d(F.prelu(out, p["separator.DPT.output.0.weight"]), p["separator.DPT.output.1.weight"], p["separator.DPT.output.1.bias"])
        a, b = merge_halves(out.view(B * self.n_src, N, self.K, -1))
        m = a + b
        if rest > 0:
            m = m[:, :, :-rest]
        m = m.contiguous()
        g = torch.tanh(F.conv1d(m, p["separator.output.0.weight"], p["separator.output.0.bias"])) * \
            torch.sigmoid(F.conv1d(m, p["separator.output_gate.0.weight"], p["separator.output_gate.0.bias"]))
        s = g.transpose(1, 2).contiguous().view(B, self.n_src, -1, N).view(B * self.n_src, -1, N).transpose(1, 2).contiguous()
        m = F.relu(F.conv1d(s, p["mask_conv1x1.0.weight"], None)).view(B, self.n_src, self.E, -1)
        sw = (w.unsqueeze(1) * m).transpose(2, 3)
        est = overlap_and_add(F.linear(sw, p["decoder.basis_signals.weight"], None), self.W // 2)
        return combine(est.reshape(1, B, self.n_src, 1, -1), 1)

    __call__ = forward
