"""Conv-TasNet (non-causal) ready for W8A8 fake-quantization, MI355X edition.

Same module tree / attribute names / constructor arguments / `quantize_model` path table as the
reference's quantization/qat/models/convtasnetq.py (ConvBlock :11-42, MaskGenerator :45-115,
ConvTasNetQ :118-288), so float and quantized `state_dict`s interchange key for key (948 keys at
the shipped size).  Every op -- float or quantized -- executes as a HIP kernel (no ATen compute).
"""
import torch
import torch.nn as nn

from .... import ops
from ....process import postprocess, preprocess
from ..float_exec import HipSequential, apply_module
from ..qat_layers import Add, Mul, run_conv1d_pair
from ..qat_utils import quantize_modules, replace_decoderq, replace_encoderq

EPS = 1e-8


class ConvBlock(nn.Module):
    """1x1 conv + PReLU + gLN + depthwise dilated conv + PReLU + gLN, then residual and skip 1x1 convs."""

    def __init__(self, io_channels, hidden_channels, kernel_size, padding, dilation=1):
        super().__init__()
        self.shared_block = HipSequential(
            nn.Conv1d(io_channels, hidden_channels, 1),
            nn.PReLU(),
            nn.GroupNorm(1, hidden_channels, eps=EPS),
            nn.Conv1d(hidden_channels, hidden_channels, kernel_size, padding=padding, dilation=dilation,
                      groups=hidden_channels),
            nn.PReLU(),
            nn.GroupNorm(1, hidden_channels, eps=EPS),
        )
        self.res_conv = nn.Conv1d(hidden_channels, io_channels, 1)
        self.skip_conv = nn.Conv1d(hidden_channels, io_channels, 1)
        self.add = Add()

    fqss_linear_pairs = (("res_conv", "skip_conv"),)   # same-input 1x1 convs: runtime.QuantTables concatenates their codes

    def forward(self, x, skip_sum=None, skip_add=None):
        """-> (x + res_conv(f), skip_conv(f)); skip_sum / skip_add: the running skip sum and the Add layer the caller applies to
        (skip_sum, <second output>) next -- known here so that both adds can run in the res | skip GEMM's epilogue"""
        x_blk, x_res = ops.fork2(x)
        f = self.shared_block(x_blk)
        # one GEMM for both (quantizing phase, graph mode); `residual` is consumed only by self.add below and `skip_out`
        # only by the skip sum of MaskGenerator.forward (or dropped): their AddQ backward also runs these convs'
        # output-quantizer backward, and their AddQ forward runs in the GEMM's epilogue
        fused = run_conv1d_pair(self.res_conv, self.skip_conv, f, sole_ew_consumers=True,
                                adds=((x_res, self.add), (skip_sum, skip_add) if skip_add is not None else None))
        if fused is not None:
            residual, skip_out = fused
        else:
            f_res, f_skip = ops.fork2(f)
            residual = apply_module(self.res_conv, f_res)
            skip_out = apply_module(self.skip_conv, f_skip)
        return self.add(x_res, residual), skip_out


class MaskGenerator(nn.Module):
    """TCN separation module: bottleneck, num_stacks x num_layers ConvBlocks with dilation 2**layer,
    skip-sum, PReLU + 1x1 conv + mask activation."""

    def __init__(self, input_dim, n_srcs, kernel_size, num_feats, num_hidden, num_layers, num_stacks, msk_activate):
        super().__init__()
        self.input_dim = input_dim
        self.n_srcs = n_srcs
        self.bottleneck = HipSequential(nn.GroupNorm(1, input_dim, eps=EPS), nn.Conv1d(input_dim, num_feats, 1))
        self.receptive_field = 0
        self.TCN = nn.ModuleList([])
        for s in range(num_stacks):
            for layer in range(num_layers):
                d = 2 ** layer
                self.TCN.append(ConvBlock(num_feats, num_hidden, kernel_size, dilation=d, padding=d))
                self.receptive_field += kernel_size if s == 0 and layer == 0 else (kernel_size - 1) * d
        self.adds = nn.ModuleList([Add() for _ in range(len(self.TCN) - 1)])
        for a in self.adds:
            a.fqss_chain = True     # forward below: the running skip sum is consumed by the next add alone -> ONE backward launch for the chain
        if msk_activate == "sigmoid":
            act = nn.Sigmoid()
        elif msk_activate == "relu":
            act = nn.ReLU()
        else:
            raise ValueError(f"Unsupported activation {msk_activate}")
        self.mask_net = HipSequential(nn.PReLU(), nn.Conv1d(num_feats, input_dim * n_srcs, 1), act)
        self.mask_net.fqss_sole_consumer = True     # its output only feeds ConvTasNetQ.mul (see HipSequential.forward)

    fqss_cut_every = 0     # > 0: a backward cut point (ops.cut) after every that many TCN blocks, set by fqss_segments()

    def forward(self, x):
        batch = x.shape[0]
        feats = self.bottleneck(x)
        feats, output = self.TCN[0](feats)
        ce = self.fqss_cut_every
        for i, layer in enumerate(self.TCN[1:]):
            if ce and (i + 1) % ce == 0:
                feats, output = ops.cut(feats, output)       # blocks 0 .. i are one backward segment
            feats, skip = layer(feats, skip_sum=output, skip_add=self.adds[i])
            output = self.adds[i](output, skip)
        output = self.mask_net(output)
        return ops.reshape_tagged(output, batch, self.n_srcs, self.input_dim, -1)


class ConvTasNetQ(nn.Module):
    def __init__(self, n_spks=1, kernel_size=32, stride=16, n_filters=512, mask_kernel_size=3, bn_chan=128,
                 hid_chan=512, n_blocks=8, n_repeats=3, mask_act="relu"):
        super().__init__()
        self.n_srcs = n_spks
        self.enc_num_feats = n_filters
        self.set_splitter_combiner(1, 1)
        self.encoder = nn.Conv1d(1, n_filters, kernel_size, stride=stride, padding=0, bias=False)
        self.masker = MaskGenerator(input_dim=n_filters, n_srcs=n_spks, kernel_size=mask_kernel_size, num_feats=bn_chan,
                                    num_hidden=hid_chan, num_layers=n_blocks, num_stacks=n_repeats, msk_activate=mask_act)
        self.decoder = nn.ConvTranspose1d(n_filters, 1, kernel_size, stride=stride, padding=0, bias=False)
        self.mul = Mul()

    def pre_process(self, x):
        return preprocess(x, n_splitter=self.n_splitter)

    def post_process(self, x):
        return postprocess(x, n_combiner=self.n_combiner)

    def forward(self, x):
        """x [B, 1, L] (or [B, L]) -> separated sources [B, S, L']"""
        x = self.pre_process(x)
        batch = x.shape[0]
        feats = apply_module(self.encoder, x)                                  # [B, F, M]
        f_mask, f_mul = ops.fork2(feats)
        if self.masker.fqss_cut_every:
            (f_mul,) = ops.cut(f_mul, late=True)       # this edge jumps over every backward segment of the TCN stack
        masked = self.mul(self.masker(f_mask), ops.reshape_tagged(f_mul, batch, 1, self.enc_num_feats, -1))  # [B, S, F, M]
        masked = ops.reshape_tagged(masked, batch * self.n_srcs, self.enc_num_feats, -1)
        out = apply_module(self.decoder, masked)                               # [D, B*S, 1, L] or [B*S, 1, L]
        out = out.reshape((self.n_combiner, batch, self.n_srcs, 1, -1))
        return self.post_process(out)

    def fqss_segments(self, n):
        """Split the network into <= n backward segments (forward order) and arm the cut points between them: returns the module
        lists whose parameters become final when the corresponding segment's backward has run.  runtime.KDTrainStep lays the
        gradient arena out in this order and all-reduces a segment's slice while the next segment's backward runs."""
        tcn, adds = self.masker.TCN, self.masker.adds
        nb = len(tcn)
        n = max(1, min(int(n), nb))
        every = -(-nb // n)
        self.masker.fqss_cut_every = every if n > 1 else 0
        segs = []
        for s0 in range(0, nb, every):
            s1 = min(nb, s0 + every)
            mods = [tcn[i] for i in range(s0, s1)] + [adds[i - 1] for i in range(max(s0, 1), s1)]
            if s0 == 0:
                mods = [self.encoder, self.masker.bottleneck] + mods
            if s1 == nb:
                mods = mods + [self.masker.mask_net, self.mul, self.decoder]
            segs.append(mods)
        return segs

    def load_pretrain(self, weights_path):
        """order-based key mapping of a checkpoint with the same number of entries (reference :225-237)"""
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
            if type(m) is ConvTasNetQ:
                replace_encoderq(m, ["encoder"], dict(io, n_splitter=self.n_splitter, act_n_bits=act_n_bits,
                                                      in_quant=in_quant, in_act_n_bits=in_act_n_bits))
                replace_decoderq(m, ["decoder"], dict(io, n_combiner=self.n_combiner, act_n_bits=out_act_n_bits,
                                                      out_quant=out_quant, out_act_n_bits=out_act_n_bits))
                quantize_modules(m, ["mul"], p)
            elif type(m) is ConvBlock:
                for group in (["0", "1"], ["2"], ["3", "4"], ["5"]):
                    quantize_modules(m.shared_block, group, p)
                for name in ("res_conv", "skip_conv", "add"):
                    quantize_modules(m, [name], p)
            elif type(m) is MaskGenerator:
                for group in (["0"], ["1"]):
                    quantize_modules(m.bottleneck, group, p)
                quantize_modules(m.mask_net, ["0"], p)
                quantize_modules(m.mask_net, ["1", "2"], p)
                for i in range(len(m.adds)):
                    quantize_modules(m.adds, [str(i)], p)
