"""Forward LSTM recurrence at the DPTNet bench shape (S = 250 x 194 and S = 97 x 500 sequences, H = 128): the
round-2 form (FQSS_LSTM_V1=1, read per call) against the staggered form, agreement and time.  python tools/lstm_probe.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
dev = "cuda"
H = 128
for S, B in ((250, 194), (97, 500), (50, 3)):
    pre = torch.randn(S, B, 8 * H, device=dev) * 0.5
    whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
    bhh = torch.randn(2, 4 * H, device=dev) * 0.1
    outs = {}
    for w16 in ("1", "0"):
        os.environ["FQSS_LSTM_V1"] = w16
        for save in (True, False):
            h, g, c = K.lstm_fwd(pre, whh, bhh, S, B, H, save=save)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                K.lstm_fwd(pre, whh, bhh, S, B, H, save=save)
            e1.record()
            torch.cuda.synchronize()
            print(f"S {S} B {B} v1 {w16} save {save}: {e0.elapsed_time(e1) / reps * 1e3:9.1f} us", flush=True)
            if save:
                outs[w16] = (h, g, c)
    for a, b, n in zip(outs["0"], outs["1"], ("h", "gates", "cell")):
        print(f"   {n}: max |d| {float((a - b).abs().max()):.3e}  (max |ref| {float(a.abs().max()):.3f})", flush=True)
os.environ.pop("FQSS_LSTM_V1", None)
