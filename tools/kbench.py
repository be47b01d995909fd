#!/usr/bin/env python3
"""Kernel micro-bench (GPU box): every student kernel of a TCN block at its cfg-2 launch shape (B = 8, M = 3999), timed with HIP events
over a ROTATING set of operand buffers larger than the 256 MiB Infinity Cache (in the step nothing is cache-resident: a block moves ~1 GB).

    python tools/kbench.py [name-substring ...]      -> one line per case: us per launch, algorithmic MB, TB/s
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqss_amd import kernels as K   # noqa: E402

B, M, NB, NH = 8, 3999, 128, 512
dev = "cuda"
n = B * M


def act(C):
    return K.empty_act((B, C, M), dev).normal_()


def codes(C):
    return K.empty_codes((B, C, M), dev).random_(0, 256)


def ring(make, nbytes):
    """enough copies of an operand set to exceed ~600 MB"""
    k = max(2, int(600e6 // max(nbytes, 1)) + 1)
    return [make() for _ in range(min(k, 48))]


lo, hi = torch.tensor([-2.0], device=dev), torch.tensor([2.5], device=dev)
slope = torch.tensor([0.25], device=dev)
gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64, device=dev)
pga, pgb = torch.zeros_like(gacc), torch.zeros_like(gacc)
ones = lambda c: torch.ones(c, 1, 1, device=dev) * 0.2
wc_up = K.wq_codes(torch.randn(NH, NB, 1, device=dev) * 0.05, -ones(NH), ones(NH))
wr_, ws_ = (K.wq_codes(torch.randn(NB, NH, 1, device=dev) * 0.05, -ones(NB), ones(NB)) for _ in range(2))
pc = K.WCodes()
pc.Co, pc.Ci = 2 * NB, NH
pc.idx, pc.idxT = torch.cat([wr_.idx, ws_.idx], 0).contiguous(), torch.cat([wr_.idxT, ws_.idxT], 1).contiguous()
pc.dw, pc.rw = torch.cat([wr_.dw, ws_.dw]), torch.cat([wr_.rw, ws_.rw])
bu, bd, bd2 = torch.randn(NH, device=dev), torch.randn(NB, device=dev), torch.randn(NB, device=dev)
gw_up, gw_pair = torch.zeros(NH, NB, device=dev), torch.zeros(2 * NB, NH, device=dev)
w_dw, b_dw = torch.randn(NH, 1, 3, device=dev), torch.randn(NH, device=dev) * 0.1
gm_, bt_ = torch.rand(NH, device=dev) + 0.5, torch.randn(NH, device=dev) * 0.1
gg, gb2, gbb, gw_dw = torch.zeros(NH, device=dev), torch.zeros(NH, device=dev), torch.zeros(NH, device=dev), torch.zeros(NH, 1, 3, device=dev)
pb_a, pb_b = torch.zeros(NB, device=dev), torch.zeros(NB, device=dev)
_, _, mr = K.gnq_fwd(codes(NH), lo, hi, gm_, bt_, 1e-8, lo, hi, False)

CASES = []


def case(name, mb, make, fn):
    CASES.append((name, mb, make, fn))


MBh, MBb = 1e-6 * NH * n, 1e-6 * NB * n     # MB per byte-per-element of a hidden / bottleneck tensor
case("ewq_fwd add (codes+codes->codes) C=128", 3 * MBb, lambda: (codes(NB), codes(NB)),
     lambda s: K.ewq_fwd(s[0], lo, hi, s[1], lo, hi, None, 1.0, 0, None, lo, hi, False))
case("ewq_bwd_p both producers C=128", (2 + 12 + 8) * MBb, lambda: (codes(NB), codes(NB), act(NB), act(NB), act(NB)),
     lambda s: K.ewq_bwd_p(s[0], lo, hi, s[1], lo, hi, 1.0, s[2], 0, None, lo, hi, gacc, NB, prod_a=(s[3], 0, None, pga, pb_a),
                           prod_b=(s[4], 0, None, pgb, pb_b)))
case("ewq_bwd_p one producer C=128", (2 + 8 + 8) * MBb, lambda: (codes(NB), codes(NB), act(NB), act(NB)),
     lambda s: K.ewq_bwd_p(s[0], lo, hi, s[1], lo, hi, 1.0, s[2], 0, None, lo, hi, gacc, NB, prod_b=(s[3], 0, None, pgb, pb_b)))
case("gnq_fwd (stats+apply) C=512", 3 * MBh, lambda: (codes(NH),),
     lambda s: K.gnq_fwd(s[0], lo, hi, gm_, bt_, 1e-8, lo, hi, False))
case("gnq_bwd plain C=512", (1 + 4 + 1 + 4 + 4) * MBh, lambda: (codes(NH), act(NH)),
     lambda s: K.gnq_bwd(s[0], lo, hi, s[1], gm_, bt_, mr, lo, hi, gacc, gg, gb2))
case("gnq_bwd_p (conv1 producer) C=512", (1 + 4 + 1 + 4 + 4 + 4) * MBh, lambda: (codes(NH), act(NH), act(NH)),
     lambda s: K.gnq_bwd(s[0], lo, hi, s[1], gm_, bt_, mr, lo, hi, gacc, gg, gb2, producer=(s[2], 1, slope, pga, gbb)))
for _d in (4, 128):
    case(f"dwq_fwd C=512 dil {_d} (+gLN statistics)", 2 * MBh, lambda: (codes(NH),),
         lambda s, _d=_d: K.dwq_fwd(s[0], lo, hi, w_dw, b_dw, _d, _d, 1, slope, lo, hi, False, stats=K.new_stats("dwq", B, NH, M, dev)))
case("dwq_bwd C=512 dil 4", 9 * MBh, lambda: (codes(NH), act(NH)),
     lambda s: K.dwq_bwd(s[0], lo, hi, w_dw, b_dw, s[1], 4, 4, 1, slope, lo, hi, gacc, gbb, gw_dw))
case("qpw_fwdq conv1 128->512 (+PReLU+fq)", MBb + 5 * MBh, lambda: (codes(NB),),
     lambda s: K.qpw_fwdq(s[0], wc_up, bu, None, lo, hi, NH, 1, slope, (lo, hi)))
case("qpw_fwdq pair 512->128+128", MBh + 10 * MBb, lambda: (codes(NH),),
     lambda s: K.qpw_fwdq(s[0], pc, bd, bd2, lo, hi, NB, 0, None, (lo, hi), (lo, hi)))
case("qpw_bwd_x conv1 (512->128)", 4 * MBh + 4 * MBb, lambda: (act(NH),), lambda s: K.qpw_bwd_x(s[0], wc_up))
case("qpw_bwd_x2 pair (256->512)", 8 * MBb + 4 * MBh, lambda: (act(NB), act(NB)), lambda s: K.qpw_bwd_x2(s[0], s[1], pc))
case("qpw_bwd_w conv1", 4 * MBh + MBb, lambda: (act(NH), codes(NB)), lambda s: K.qpw_bwd_w(s[0], s[1], lo, hi, gw_up))
case("qpw_bwd_w2 pair", 8 * MBb + MBh, lambda: (act(NB), act(NB), codes(NH)), lambda s: K.qpw_bwd_w2(s[0], s[1], s[2], lo, hi, gw_pair))
# the waveform side (decoder [16, 512, 3999] -> [16, 1, 32000]; masking product)
w_dec = torch.randn(NH, 1, 16, device=dev) * 0.1
act2 = lambda: K.empty_act((2 * B, NH, M), dev).normal_()
codes2 = lambda: K.empty_codes((2 * B, NH, M), dev).random_(0, 256)
case("ola_convtr_fwd fp32 (decoder)", 8 * MBh, lambda: (act2(),), lambda s: K.ola_convtr_fwd(s[0], w_dec, 8))
case("ola_convtr_fwd_q codes (decoder)", 2 * MBh, lambda: (codes2(),), lambda s: K.ola_convtr_fwd_q(s[0], lo, hi, w_dec, 8))
case("ola_convtr_mul_fwd (teacher mask x feats)", 12 * MBh, lambda: (act2().view(B, 2, NH, M), act(NH)),
     lambda s: K.ola_convtr_mul_fwd(s[0], s[1], w_dec, 8))
case("mulq_fwd S=2", 5 * MBh, lambda: (codes2().view(B, 2, NH, M), codes(NH)),
     lambda s: K.mulq_fwd(s[0], lo, hi, s[1], lo, hi, lo, hi, False))
case("mulq_bwd S=2 (+producer)", (2 + 1 + 8 + 8 + 8 + 4) * MBh, lambda: (codes2().view(B, 2, NH, M), codes(NH), act2().view(B, 2, NH, M), act2()),
     lambda s: K.mulq_bwd(s[0], lo, hi, s[1], lo, hi, s[2], lo, hi, gacc, prod=(s[3], K.ACT_RELU, None, pga, None)))
case("frames_conv_fwd (decoder dgrad) 1->512", 8 * MBh, lambda: (torch.randn(2 * B, 1, 32000, device=dev),),
     lambda s: K.frames_conv_fwd(s[0], w_dec.view(NH, 1, 16), 8))
case("frames_conv_fwd + addend (decoder dgrad at the fork)", 12 * MBh, lambda: (torch.randn(2 * B, 1, 32000, device=dev), act2()),
     lambda s: K.frames_conv_fwd(s[0], w_dec.view(NH, 1, 16), 8, add=s[1]))
w_enc2 = torch.randn(NH, 2, 16, device=dev) * 0.1
case("frames_conv_fwd encoder 2->512", 4 * MBh, lambda: (torch.randn(B, 2, 32000, device=dev),),
     lambda s: K.frames_conv_fwd(s[0], w_enc2, 8))
gw_dec = torch.zeros(NH, 1, 16, device=dev)
case("frames_wgrad (decoder) a fp32", 8 * MBh, lambda: (act2(), torch.randn(2 * B, 1, 32000, device=dev)),
     lambda s: K.frames_wgrad(s[0], s[1], gw_dec, 8))
case("frames_wgrad1_q (decoder) a codes", 2 * MBh, lambda: (codes2(), torch.randn(2 * B, 1, 32000, device=dev)),
     lambda s: K.frames_wgrad1_q(s[0], lo, hi, s[1], gw_dec, 8))
case("axpby C=128", 12 * MBb, lambda: (act(NB), act(NB)), lambda s: K.axpby(s[0], s[1], 1.0))


def main():
    sel = sys.argv[1:]
    for name, mb, make, fn in CASES:
        if sel and not any(s in name for s in sel):
            continue
        per_set = sum(t.numel() * t.element_size() for t in make())
        sets = ring(make, per_set)
        for s in sets[:3]:
            fn(s)
        torch.cuda.synchronize()
        iters = max(20, 2 * len(sets))
        # the launches are recorded into one hipGraph (python launch overhead ~15 us would hide the short kernels) and replayed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(2):
                fn(sets[i % len(sets)])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(iters):
                fn(sets[i % len(sets)])
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * iters) * 1e3
        del g
        print(f"{name:44s} {us:7.1f} us  {mb:7.1f} MB  {mb / us:6.2f} TB/s", flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
