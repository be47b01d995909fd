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


def synth_batch_2band(B, T, seed=0, device="cpu"):
    """A STREAM of separable synthetic mixtures for the convergence gate (tools/make_goldens_long.py and
    tests/test_gpu_converge.py call this same function): speaker 1 is low-passed noise, speaker 2 the same noise process
    modulated to the upper half band (9-tap triangular FIR, (-1)^k for the high-pass), so a separator can actually learn
    something from never-repeating batches -- two sources with the SAME spectrum (synth_batch) cannot be told apart.  Filtered
    in fp64 and rounded once, so the fp32 values do not depend on the host's convolution backend."""
    g = torch.Generator().manual_seed(seed)
    n = 0.05 * torch.randn(B, 2, T + 8, generator=g)
    lp = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 4.0, 3.0, 2.0, 1.0], dtype=torch.float64) / 25.0
    hp = lp * torch.tensor([1.0, -1.0] * 4 + [1.0], dtype=torch.float64)
    w = torch.stack([lp, hp]).view(2, 1, 9)
    s = F.conv1d(n.double(), w, groups=2).float()          # [B, 2, T]
    return s.sum(1, keepdim=True).to(device), s.to(device)


def synth_stems(B, S, C, T, seed=0, device="cpu"):
    """A STREAM of stereo multi-stem mixtures for the HTDemucs convergence gate (tools/make_goldens_long.py htdemucs and
    tests/test_gpu_converge.py call this same function): stem s is the 9-tap triangular low-pass noise of synth_batch_2band
    modulated by cos(pi s k / (S - 1 or 1)) -- stem 0 low band, the last stem high band -- with a per-channel gain so that the two
    audio channels differ.  Returns (mix [B, C, T], stems [B, S, C, T]); filtered in fp64 and rounded once."""
    g = torch.Generator().manual_seed(seed)
    n = 0.3 * torch.randn(B, S * C, T + 8, generator=g)
    lp = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 4.0, 3.0, 2.0, 1.0], dtype=torch.float64) / 25.0
    k = torch.arange(9, dtype=torch.float64)
    w = torch.stack([lp * torch.cos(torch.pi * s * k / max(S - 1, 1)) * (1.0 if c == 0 else 0.8) for s in range(S) for c in range(C)])
    src = F.conv1d(n.double(), w.view(S * C, 1, 9), groups=S * C).float().view(B, S, C, T)
    return src.sum(1).to(device), src.to(device)
