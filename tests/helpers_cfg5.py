"""Name-keyed deterministic parameters and inputs of the full-architecture HTDemucs fixture (cfg5_step.npz): the SAME generator runs
next to the real reference (tools/make_goldens_htdemucs_full.py) and next to the GPU build, so both start from identical weights
without shipping 166 MB of state."""
import zlib

import numpy as np
import torch

KW = dict(sources=["drums", "bass", "other", "vocals"], bottom_channels=512, segment=1)
T_SAMPLES = 44100


def keyed_randn(key, shape, scale=1.0):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return torch.randn(*shape, generator=g) * scale


def cfg5_fill(model, prefix):
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("min_range") or k.endswith("max_range"):
                continue
            if k.endswith(".scale"):                                   # LayerScale: away from its 1e-4 init so the branches matter
                v = 0.5 + keyed_randn(prefix + k, tuple(p.shape), 0.1)
            elif p.dim() == 1 and ("norm" in k or ".gn." in k or k.split(".")[-2] in ("1", "4")) and k.endswith("weight"):
                v = 1.0 + keyed_randn(prefix + k, tuple(p.shape), 0.1)
            elif p.dim() == 1:
                v = keyed_randn(prefix + k, tuple(p.shape), 0.05)
            else:
                fan = p[0].numel() if "convTr" not in k and "conv_tr" not in k else p.shape[0] * p[0, 0].numel()
                v = keyed_randn(prefix + k, tuple(p.shape), 1.0 / np.sqrt(max(fan, 1)))
            p.copy_(v.to(p.device))


def cfg5_kw(seconds=1):
    """constructor arguments of the fixture model; `segment` = the excerpt length so the eval-mode teacher is not zero-padded"""
    return dict(KW, segment=seconds)


def cfg5_batch(seconds=1):
    """1 x `seconds` s of stereo 44.1 kHz stems (seconds = 10: the BASELINE workload's segment length, cfg5_full_step.npz)"""
    T = T_SAMPLES * seconds
    src = keyed_randn("cfg5.src" if seconds == 1 else f"cfg5.src.{seconds}s", (1, 4, 2, T), 0.3)
    src = torch.nn.functional.avg_pool1d(src.reshape(8, 1, T), 5, 1, 2).reshape(1, 4, 2, T) * 2.0
    return src.sum(1), src
