"""oracle/sepformer_oracle.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (torch fp32 ATen ops + autograd) of the Sepformer W8A8 QAT forward (cfg 4), functional
over a flat ``{state_dict key: tensor}`` table like the other oracles.  Only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import it.

Reference followed (ssi-research/FQSS @ 2024_10_08), quantization/qat/models/sepformerq.py:
  positional encoding + ConstQ + broadcasting AddQ   :13-47, 117-118
  TransformerLayer     :70-97   (pre-norm attention and feed-forward; the residual `+` are NOT quantized)
  TransformerBlock     :116-123 ;  DualPathBlock :145-177 (GroupNormQ over [F, K, S] per sample, AddQ residuals)
  MaskGenerator        :329-345 with padding / segmentation / over_add :239-327 (over_add's `+` is NOT quantized)
  SepformerQ.forward   :398-439 ;  quantize_model :474-526 decides which ops carry quantizers
  decoder              qat_layers.py:1305-1361 with train_res_dec=True (:1137-1146, 1194-1202): the LSB channel has its own
                       trainable transposed conv and weight quantizer
  teacher              the same graph with float ops (train_utils.py:25), n_splitter = n_combiner = 1
The step (loss, PIT, clip 5, Adam) is oracle/fqss_oracle.py's: at B = 1 -- the shipped per-GPU batch -- the speechbrain env's KD
objective (speechbrain_librimix_trainer.py:99-115) equals the asteroid env's term for term (tools/make_goldens_sepformer.py).

Pinned by tests/golden/sep_layers.npz, sep_tiny_step.npz, cfg4_step.npz (produced by the imported reference).
"""
import math

import torch
import torch.nn.functional as F

from .dptnet_oracle import DQTable, mha_core, merge_halves, split_feature
from .fqss_oracle import combine, split

EPS, EPS_T = 1e-8, 1e-6


class StudentSepformerQ(DQTable):
    def __init__(self, state_dict, n_src=2, kernel_size=16, stride=8, chunk_size=250, n_heads=8, n_splitter=2, n_combiner=2):
        super().__init__(state_dict)
        self.n_src, self.Kw, self.stride, self.K, self.nh = n_src, kernel_size, stride, chunk_size, n_heads
        self.n_splitter, self.n_combiner = n_splitter, n_combiner
        self.n_rep = 1 + max(int(k.split(".")[2]) for k in self.p if k.startswith("masker.layers."))
        self.n_lay = 1 + max(int(k.split(".")[5]) for k in self.p if k.startswith("masker.layers.0.intra_transformer_block.layers."))

    def ln(self, name, x):
        y = F.layer_norm(x, x.shape[-1:], self.p[name + ".layernorm.weight"], self.p[name + ".layernorm.bias"], EPS_T)
        return self._A(name, y)

    def layer(self, name, x):
        """x [B', L, F] batch-first"""
        q = self.ln(name + ".norm1", x).permute(1, 0, 2)
        x = x + self.mha_q(name + ".mha", q, self.nh).permute(1, 0, 2)
        h = self.ln(name + ".norm2", x)
        h = self.linear_q(name + ".ffn.0", h)
        h = self._A(name + ".ffn.1", F.relu(h))
        return x + self.linear_q(name + ".ffn.3", h)

    def block(self, name, x):
        pe = self.A(name + ".pos.const.activation_fake_quantize", self.p[name + ".pos.pe"][:, : x.shape[1]].detach())
        x = self._A(name + ".pos_add", x + pe)
        for i in range(self.n_lay):
            x = self.layer(f"{name}.layers.{i}", x)
        return self.ln(name + ".norm", x)

    def dual(self, name, x):
        B, F_, K, S = x.shape
        intra = self.block(name + ".intra_transformer_block", x.permute(0, 3, 2, 1).contiguous().reshape(B * S, K, F_))
        intra = intra.reshape(B, S, K, F_).permute(0, 3, 2, 1).contiguous()
        intra = self._A(name + ".intra_add", self._gn(name + ".intra_norm", intra) + x)
        inter = self.block(name + ".inter_transformer_block", intra.permute(0, 2, 3, 1).contiguous().reshape(B * K, S, F_))
        inter = inter.reshape(B, K, S, F_).permute(0, 3, 1, 2).contiguous()
        return self._A(name + ".inter_add", self._gn(name + ".inter_norm", inter) + intra)

    def masker(self, x):
        B, F_, M = x.shape
        y = self._conv("masker.conv1d", self._gn("masker.norm", x))
        seg, gap = split_feature(y, self.K)
        for i in range(self.n_rep):
            seg = self.dual(f"masker.layers.{i}", seg)
        o = self.conv2d_q("masker.conv2d", self._nl("masker.prelu", seg))
        a, b = merge_halves(o.reshape(B * self.n_src, F_, self.K, -1))
        m = a + b
        if gap > 0:
            m = m[:, :, :-gap]
        g = self._A("masker.mul", self.conv1d_nl_q("masker.net_out.0", m, "tanh") * self.conv1d_nl_q("masker.net_gate.0", m, "sigmoid"))
        return self.conv1d_nl_q("masker.end_conv.0", g, "relu").reshape(B, self.n_src, F_, -1)

    def decoder(self, x):
        name = "decoder"
        w = self._W(name, self.p[name + ".convTr1d.weight"])
        y0 = self._A(name, F.conv_transpose1d(x, w, None, stride=self.stride))
        if self.n_combiner == 1:
            return y0
        rb = name + ".residual_error_block"
        Yq = F.conv1d(y0, self._W(rb, self.p[rb + ".residual_encoder.weight"]), None, stride=self.stride)
        Y1 = self._A(rb, x - Yq)
        w_dec = self.Wq(rb + ".weight_fake_quantize_dec", self.p[rb + ".residual_decoder.weight"])
        y1 = self.aq[name + ".activation_fake_quantize_residual"](F.conv_transpose1d(Y1, w_dec, None, stride=self.stride))
        return torch.stack([y0, y1])

    def forward(self, x):
        x = split(x, self.n_splitter)
        B = x.shape[0]
        feats = self._conv("encoder.0", x, nl="relu", stride=self.stride)
        masked = self._A("mul", self.masker(feats) * feats.unsqueeze(1))
        dec = self.decoder(masked.reshape(B * self.n_src, feats.shape[1], -1))
        return combine(dec.reshape(self.n_combiner, B, self.n_src, 1, -1), self.n_combiner)

    __call__ = forward


class TeacherSepformer:
    def __init__(self, state_dict, n_src=2, stride=8, chunk_size=250, n_heads=8):
        self.p = {k: v.detach().clone().float() for k, v in state_dict.items()}
        self.n_src, self.stride, self.K, self.nh = n_src, stride, chunk_size, n_heads
        self.n_rep = 1 + max(int(k.split(".")[2]) for k in self.p if k.startswith("masker.layers."))
        self.n_lay = 1 + max(int(k.split(".")[5]) for k in self.p if k.startswith("masker.layers.0.intra_transformer_block.layers."))

    def layer(self, n, x):
        p = self.p
        E = x.shape[-1]
        q = F.layer_norm(x, (E,), p[n + ".norm1.weight"], p[n + ".norm1.bias"], EPS_T).permute(1, 0, 2)
        X = F.linear(q, p[n + ".mha.in_proj_weight"], p[n + ".mha.in_proj_bias"])
        heads = mha_core(X[..., :E], X[..., E:2 * E], X[..., 2 * E:], self.nh)
        x = x + F.linear(heads, p[n + ".mha.out_proj.weight"], p[n + ".mha.out_proj.bias"]).permute(1, 0, 2)
        h = F.layer_norm(x, (E,), p[n + ".norm2.weight"], p[n + ".norm2.bias"], EPS_T)
        h = F.linear(F.relu(F.linear(h, p[n + ".ffn.0.weight"], p[n + ".ffn.0.bias"])), p[n + ".ffn.3.weight"], p[n + ".ffn.3.bias"])
        return x + h

    def block(self, n, x):
        p = self.p
        x = x + p[n + ".pos.pe"][:, : x.shape[1]]
        for i in range(self.n_lay):
            x = self.layer(f"{n}.layers.{i}", x)
        return F.layer_norm(x, x.shape[-1:], p[n + ".norm.weight"], p[n + ".norm.bias"], EPS_T)

    def forward(self, x):
        p = self.p
        if x.dim() == 2:
            x = x.unsqueeze(1)
        B = x.shape[0]
        feats = F.relu(F.conv1d(x, p["encoder.0.weight"], None, stride=self.stride))
        F_ = feats.shape[1]
        y = F.conv1d(F.group_norm(feats, 1, p["masker.norm.weight"], p["masker.norm.bias"], EPS), p["masker.conv1d.weight"], None)
        seg, gap = split_feature(y, self.K)
        for i in range(self.n_rep):
            n = f"masker.layers.{i}"
            _, _, K, S = seg.shape
            intra = self.block(n + ".intra_transformer_block", seg.permute(0, 3, 2, 1).contiguous().reshape(B * S, K, F_))
            intra = intra.reshape(B, S, K, F_).permute(0, 3, 2, 1).contiguous()
            intra = F.group_norm(intra, 1, p[n + ".intra_norm.weight"], p[n + ".intra_norm.bias"], EPS) + seg
            inter = self.block(n + ".inter_transformer_block", intra.permute(0, 2, 3, 1).contiguous().reshape(B * K, S, F_))
            inter = inter.reshape(B, K, S, F_).permute(0, 3, 1, 2).contiguous()
            seg = F.group_norm(inter, 1, p[n + ".inter_norm.weight"], p[n + ".inter_norm.bias"], EPS) + intra
        o = F.conv2d(F.prelu(seg, p["masker.prelu.weight"]), p["masker.conv2d.weight"], p["masker.conv2d.bias"])
        a, b = merge_halves(o.reshape(B * self.n_src, F_, self.K, -1))
        m = a + b
        if gap > 0:
            m = m[:, :, :-gap]
        g = torch.tanh(F.conv1d(m, p["masker.net_out.0.weight"], p["masker.net_out.0.bias"])) * \
            torch.sigmoid(F.conv1d(m, p["masker.net_gate.0.weight"], p["masker.net_gate.0.bias"]))
        mask = F.relu(F.conv1d(g, p["masker.end_conv.0.weight"], None)).reshape(B, self.n_src, F_, -1)
        masked = (mask * feats.unsqueeze(1)).reshape(B * self.n_src, F_, -1)
        dec = F.conv_transpose1d(masked, p["decoder.weight"], None, stride=self.stride)
        return combine(dec.reshape(1, B, self.n_src, 1, -1), 1)

    __call__ = forward
