#!/usr/bin/env python3
"""A handful of hot kernels of cfg 3 / 4 / 5 launched in isolation at their real shapes, for counter passes
(tools/kprobe_pmc.sh) and HIP-event timing: python tools/kprobe.py [time]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from fqss_amd import kernels as K  # noqa: E402


def cases():
    dev = "cuda"
    out = []
    B, nh, hd, L = 4, 8, 64, 3448
    E = nh * hd
    q, k, v, go = (torch.randn(B, L, E, device=dev) * 0.3 for _ in range(4))
    o, st = K.attn_long_fwd(q, k, v, nh, True)
    out.append(("attn_long fwd 3448^2", lambda: K.attn_long_fwd(q, k, v, nh, True)))
    out.append(("attn_long bwd 3448^2", lambda: K.attn_long_bwd(q, k, v, o, go, st, nh, True)))
    for name, R, Ci, Co in (("sepformer ffn0", 8500, 256, 1024), ("sepformer out_proj", 8500, 256, 256), ("sepformer ffn3", 8500, 1024, 256),
                            ("sepformer in_proj", 8500, 256, 768), ("dptnet lstm proj", 48500, 64, 1024), ("dptnet ffn", 48500, 256, 64),
                            ("htdemucs 512", 13792, 512, 512), ("htdemucs ffn", 13792, 512, 2048)):
        x, w, b = torch.randn(R, Ci, device=dev), torch.randn(Co, Ci, device=dev), torch.randn(Co, device=dev)
        g, gw = torch.randn(R, Co, device=dev), torch.zeros(Co, Ci, device=dev)
        xc = torch.randint(0, 256, (R, Ci), device=dev, dtype=torch.uint8)
        lo, hi = torch.tensor([-1.0], device=dev), torch.tensor([1.0], device=dev)
        out.append((f"{name} fwd", lambda x=x, w=w, b=b: K.rowlin_fwd(x, w, b)))
        out.append((f"{name} bwd_x", lambda g=g, w=w: K.rowlin_bwd_x(g, w)))
        out.append((f"{name} bwd_w", lambda g=g, x=x, gw=gw: K.rowlin_bwd_w(g, x, gw)))
        if K.qrow_bwd_ok(Ci, Co):
            out.append((f"{name} bwd_w coded", lambda g=g, xc=xc, gw=gw: K.qrow_bwd_w(g, xc, lo, hi, gw)))
            wc = K.WCodes()
            wc.Ci, wc.Co = Ci, Co
            wc.idx = torch.randint(-127, 128, (Co, Ci), device=dev, dtype=torch.int8)
            wc.dw = torch.rand(Co, device=dev) * 0.01
            out.append((f"{name} bwd_x coded", lambda g=g, wc=wc: K.qrow_bwd_x(g, wc)))
    return out


def main():
    cs = cases()
    timing = len(sys.argv) > 1
    for name, fn in cs:
        for _ in range(2):
            fn()
        if timing:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"{name:32s} {e0.elapsed_time(e1) / 20 * 1e3:9.1f} us", flush=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
