"""Error of the LSTM recurrence's gate functions (csrc/lstm.hip gate_fn) against float64, in ulp of the exact value, by range of the argument.
python tools/lstm_gate_err.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import _lib  # noqa: E402

n = 1 << 22
edges = [0, 0.36, 1, 2, 4, 8, 16, 32, 64, 87]
for lo, hi in zip(edges[:-1], edges[1:]):
    for sign in (1, -1):
        x = (torch.rand(n, dtype=torch.float64) * (hi - lo) + lo).float() * sign
        xd = x.cuda()
        sg, th = torch.empty_like(xd), torch.empty_like(xd)
        _lib.call("fqss_lstm_gate_fn", xd.data_ptr(), sg.data_ptr(), th.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        line = f"x in {sign * lo:7.2f} .. {sign * hi:7.2f}:"
        for got, want, name in ((sg.cpu(), torch.sigmoid(x.double()), "sigmoid"), (th.cpu(), torch.tanh(x.double()), "tanh")):
            ulp = torch.clamp(2.0 ** torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -126))), min=2.0 ** -126) * 2.0 ** -23
            err = (got.double() - want).abs() / ulp
            line += f"  {name} max {float(err.max()):5.2f} ulp (abs {float((got.double() - want).abs().max()):.2e})"
        print(line, flush=True)
