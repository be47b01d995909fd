"""Synthetic 2-speaker mixtures of the measurement contract: sources
s = 0.05*randn band-limited by a fixed 5-tap FIR, mixture x = s.sum(1).  Built on the host with a
seeded torch.Generator (identical to oracle/fqss_oracle.synth_batch), then moved to the device."""
import torch
import torch.nn.functional as F


def synth_batch(B, T, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    s = 0.05 * torch.randn(B, 2, T + 4, generator=g)
    fir = torch.tensor([0.1, 0.25, 0.3, 0.25, 0.1]).view(1, 1, 5)
    s = F.conv1d(s.view(B * 2, 1, T + 4), fir).view(B, 2, T)
    return s.sum(1, keepdim=True).to(device), s.to(device)
