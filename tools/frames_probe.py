"""fqss_frames_gather / fqss_frames_ola at the geometries of one HTDemucs (cfg 5) step: us per launch against the bytes they move
(input read once + frames written once for the gather; frames read once + signal written once for the overlap-add).
    python tools/frames_probe.py            (GPU box; FQSS_LIB selects a library variant)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K  # noqa: E402

dev = "cuda"
CASES = [  # (label, signal shape [B, C, H, W], kernel, stride, pad)
    ("freq enc 0  [4,4,2048,431] k8 s4", (4, 4, 2048, 431), (8, 1), (4, 1), (2, 0)),
    ("freq enc 1  [4,48,512,431] k8 s4", (4, 48, 512, 431), (8, 1), (4, 1), (2, 0)),
    ("freq enc 2  [4,96,128,431] k8 s4", (4, 96, 128, 431), (8, 1), (4, 1), (2, 0)),
    ("freq enc 3  [4,192,32,431] k8 s4", (4, 192, 32, 431), (8, 1), (4, 1), (2, 0)),
    ("time enc 0  [4,2,1,441000] k8 s4", (4, 2, 1, 441000), (1, 8), (1, 4), (0, 2)),
    ("time enc 1  [4,48,1,110250] k8 s4", (4, 48, 1, 110252), (1, 8), (1, 4), (0, 2)),
    ("time enc 2  [4,96,1,27563] k8 s4", (4, 96, 1, 27564), (1, 8), (1, 4), (0, 2)),
]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for label, shp, k, s, p in CASES:
    g = K.ConvGeom(k, s, p)
    x = K.empty_sig(shp, dev).normal_()
    f, Ho, Wo = K.frames_gather(x, g)
    nb = 4.0 * (x.numel() + f.numel())
    us = timed(lambda: K.frames_gather(x, g))
    # the transposed convolution of the decoder: frames [B, C k, Ho Wo] -> signal of the encoder's input shape
    us2 = timed(lambda: K.frames_ola(f, None, shp, g))
    print(f"{label:36s} frames {tuple(f.shape)}  gather {us:7.1f} us = {nb / us * 1e-6:5.2f} TB/s   ola {us2:7.1f} us = {nb / us2 * 1e-6:5.2f} TB/s", flush=True)
