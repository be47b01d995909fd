"""Name-keyed deterministic parameter fill for the full-size cfg-1 fixture -- the restatement of
tools/make_goldens.py::cfg1_fill (the generator that ran the REAL reference).  Same keys, same CPU generator,
so the GPU build and the reference start from identical weights without shipping 20 MB of state."""
import zlib

import numpy as np
import torch


def keyed_randn(key, shape, scale=1.0):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    return torch.randn(*shape, generator=g) * scale


def cfg1_fill(model, prefix):
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("min_range") or k.endswith("max_range"):
                continue
            if p.numel() == 1:
                p.fill_(0.25)
            elif p.dim() == 1 and ("norm" in k.lower() or k.split(".")[-2].isdigit()) and k.endswith("weight"):
                p.copy_((1.0 + keyed_randn(prefix + k, tuple(p.shape), 0.1)).to(p.device))
            elif p.dim() == 1:
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 0.02).to(p.device))
            else:
                fan = max(1, int(np.prod(p.shape[1:])))
                p.copy_(keyed_randn(prefix + k, tuple(p.shape), 1.0 / np.sqrt(fan)).to(p.device))
