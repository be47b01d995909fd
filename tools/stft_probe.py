#!/usr/bin/env python3
"""The spectrogram pair of HTDemucs at the cfg 5 shapes (32 rows x 441000 samples, n_fft 4096, hop 1024) in isolation:
python tools/stft_probe.py   (knobs: FQSS_FFT_THREADS=256|512|1024, FQSS_FFT_TW_LDS=0|1)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from fqss_amd import kernels as K  # noqa: E402


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


n_fft, hop, L = 4096, 1024, 441000
T = 431
pad = hop // 2 * 3
for rows in (8, 32):
    x = torch.randn(rows, L, device="cuda")
    z = K.stft(x, n_fft, hop, T, pad)
    g = torch.randn(rows, L, device="cuda")
    print(f"rows {rows:3d}: stft {t(lambda: K.stft(x, n_fft, hop, T, pad)):7.1f} us   istft {t(lambda: K.istft(z, n_fft, hop, pad, L)):7.1f} us   "
          f"istft_bwd {t(lambda: K.istft_bwd(g, n_fft, hop, pad, T)):7.1f} us", flush=True)
