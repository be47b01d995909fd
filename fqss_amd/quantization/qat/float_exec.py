"""Runs plain float nn modules (the un-quantized student before `quantize_model`, and the frozen
teacher of the KD step) on the SAME HIP kernels as their LayerQ counterparts, in BYPASS mode.
The reference relies on ATen for these (`nn.Conv1d.forward` ...); here no ATen compute op is on
the hot path, so the module containers dispatch through this file instead."""
import torch.nn as nn

from . import qat_layers as QL


def apply_module(m, x):
    if isinstance(m, (QL.LayerQ, HipSequential)):
        return m(x)
    if isinstance(m, nn.Identity):
        return x
    if isinstance(m, nn.Conv1d):
        return QL.run_conv1d(m, x, m.weight, None, None)
    if isinstance(m, nn.ConvTranspose1d):
        return QL.run_convtr1d(m, x, m.weight, None)
    if isinstance(m, nn.GroupNorm):
        return QL.run_groupnorm(m, x, None)
    if isinstance(m, (nn.PReLU, nn.ReLU)):
        return QL.run_nl(m, x, None)
    raise NotImplementedError(f"{type(m).__name__} has no HIP kernel on the float path")


class HipSequential(nn.Sequential):
    """nn.Sequential (same state_dict keys) whose float members execute on the HIP kernels;
    a Conv1d followed by PReLU/ReLU runs as one fused launch pair."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv1d) and i + 1 < len(mods) and isinstance(mods[i + 1], (nn.PReLU, nn.ReLU)):
                x = QL.run_conv1d(m, x, m.weight, mods[i + 1], None)
                i += 2
                continue
            # a Sequential edge has exactly one consumer: when it is a GroupNormQ, the producer hands over what that
            # layer's backward needs to run the producer's epilogue backward on the fly (ops._Producer)
            nxt = next((n for n in mods[i + 1:] if not isinstance(n, nn.Identity)), None)
            QL.ops.NEXT_IS_GROUPNORM = isinstance(nxt, QL.GroupNormQ)
            # a GroupNormQ whose one consumer is a 3-tap depthwise Conv1dNlQ (the TCN block's gLN -> depthwise conv): that layer's kernel
            # runs the GroupNorm too (ops.GroupNormActQ leaves its launch record on the tensor; run_conv1d launches it if it cannot fuse)
            QL.ops.NEXT_IS_DW3 = isinstance(m, QL.GroupNormQ) and QL.is_depthwise3(nxt)
            # the owner of the sequence may vouch for its OUTPUT instead (fqss_sole_consumer: the mask network, whose output only
            # feeds the masking MulQ): the last layer then hands its producer record to that consumer
            QL.ops.NEXT_TAKES_PRODUCER = nxt is None and getattr(self, "fqss_sole_consumer", False)
            try:
                x = apply_module(m, x)
            finally:
                QL.ops.NEXT_IS_GROUPNORM = False
                QL.ops.NEXT_IS_DW3 = False
                QL.ops.NEXT_TAKES_PRODUCER = False
            i += 1
        return x
