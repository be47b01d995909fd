#!/usr/bin/env python3
"""cfg 3 / cfg 4 (DPTNet, Sepformer 2spk 8 kHz W8A8 QAT): time one QAT step of the full-size
network on one MI355X.  Not the round's bench line (bench.py stays on cfg 2, the configuration BASELINE.json quotes the metric on)
-- a probe for DESIGN.md and for `rocprofv3 --kernel-trace --stats -- python3 tools/bench_dualpath.py`.

  python tools/bench_dualpath.py [--model dptnet|sepformer] [--B 1] [--T 24000|32000] [--steps 10] [--no-graph] [--cpu-baseline]
"""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from fqss_amd.data import synth_batch  # noqa: E402
from fqss_amd.quantization.qat import qat_quant as QQ  # noqa: E402
from fqss_amd.quantization.qat.models.dptnetq import DPTNetQ  # noqa: E402
from fqss_amd.quantization.qat.models.sepformerq import SepformerQ  # noqa: E402
from fqss_amd.quantization.qat.models.load_model import quantize_model  # noqa: E402
from fqss_amd.runtime import KDTrainStep  # noqa: E402
from fqss_amd.smoke import QCFG  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dptnet", choices=("dptnet", "sepformer"))
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--T", type=int, default=0, help="samples per mixture (default: 24000 = 3 s for DPTNet, 32000 = 4 s for Sepformer)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (Sepformer is host-launch-bound then)")
    ap.add_argument("--cpu-baseline", action="store_true", help="also time ONE step of the oracle (CPU port of the reference path) on the host cores")
    a = ap.parse_args()
    torch.manual_seed(0)
    a.T = a.T or (24000 if a.model == "dptnet" else 32000)
    lr = 4e-4 if a.model == "dptnet" else 1.5e-4
    model = DPTNetQ(n_spks=2, kernel_size=2) if a.model == "dptnet" else SepformerQ(n_spks=2, kernel_size=16, stride=8)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, dict(QCFG)).cuda().train()
    fmodel = fmodel.cuda().eval()
    x, tgt = synth_batch(a.B, a.T, seed=0, device="cuda")
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=lr, clip=5.0)
    t0 = time.time()
    for _ in range(3):                       # observer steps (weights observed at 1, quantized from 2)
        r = step(x, tgt)
    torch.cuda.synchronize()
    obs_ms = (time.time() - t0) / 3 * 1e3
    for m in model.modules():                # jump to the quantizing phase
        if isinstance(m, QQ.GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
    for _ in range(a.warmup):
        r = step(x, tgt)
    if not a.no_graph:
        step.capture(x, tgt, warmup=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        r = step(x, tgt)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    # roofline of the row GEMM that carries most of the MFMA work, at this workload's shape (HIP events on torch's stream)
    from fqss_amd import kernels as K
    from fqss_amd.kernels import dp_chunks
    if a.model == "dptnet":
        L = a.T - 1
        R, Ci, Co, what = a.B * 250 * dp_chunks(L, 250)[1], 64, 1024, "LSTM input projection (both directions)"
    else:
        L = (a.T - 16) // 8 + 1
        R, Ci, Co, what = a.B * 250 * dp_chunks(L, 250)[1], 256, 1024, "feed-forward 256 -> 1024"
    xs, ws, bs = torch.randn(R, Ci, device="cuda"), torch.randn(Co, Ci, device="cuda"), torch.randn(Co, device="cuda")
    for _ in range(3):
        K.rowlin_fwd(xs, ws, bs)
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(20):
        K.rowlin_fwd(xs, ws, bs)
    g1.record()
    torch.cuda.synchronize()
    us = g0.elapsed_time(g1) / 20 * 1e3
    tf = 2.0 * R * Ci * Co / us * 1e-6
    roofline = {"kernel": "k_gemm_x3 (fqss_rowlin_fwd)", "what": what, "shape": [R, Ci, Co], "bound": "mfma", "launch_us": round(us, 1),
                "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 3), "traffic": None}
    cpu = None
    if a.cpu_baseline:
        import oracle.fqss_oracle as O
        if a.model == "dptnet":
            import oracle.dptnet_oracle as D
            s_o = D.StudentDPTNetQ({k: v.detach().cpu() for k, v in model.state_dict().items()})
            t_o = D.TeacherDPTNet({k: v.detach().cpu() for k, v in fmodel.state_dict().items()})
        else:
            import oracle.sepformer_oracle as S
            s_o = S.StudentSepformerQ({k: v.detach().cpu() for k, v in model.state_dict().items()})
            t_o = S.TeacherSepformer({k: v.detach().cpu() for k, v in fmodel.state_dict().items()})
        s_o.leave_observer_phase()
        cores = min(16, len(os.sched_getaffinity(0)))      # the GPU box's CPU share; more threads oversubscribe the tiny ops
        torch.set_num_threads(cores)
        tr = O.Trainer(s_o, t_o, lr=lr)
        # bounded sample: ONE step on a 0.5 s excerpt of the same batch (the oracle's LSTM / attention run as Python-level torch
        # loops: a full 3-4 s step takes minutes), after a 0.1 s warm-up call; the rate is reported in the metric's unit
        Tc = min(a.T, 4000)
        xw, tw = synth_batch(a.B, 800, seed=1)
        print("cpu_baseline: warm-up ...", file=sys.stderr, flush=True)
        tr.step(xw, tw)
        xc, tc = x.cpu()[..., :Tc].contiguous(), tgt.cpu()[..., :Tc].contiguous()
        print("cpu_baseline: timed step ...", file=sys.stderr, flush=True)
        c0 = time.time()
        tr.step(xc, tc)
        sec = time.time() - c0
        cpu = {"value": round(a.B * (Tc / a.T) / sec, 4), "unit": "samples/s", "cores": cores, "kind": "port",
               "sample": f"one full QAT step of the oracle on B={a.B} x {Tc} samples ({sec:.1f} s), scaled by {Tc}/{a.T} to the workload's "
                         f"segment length, torch CPU fp32"}
    print(json.dumps({"roofline": roofline, "cpu_baseline": cpu, "workload": f"{'DPTNet' if a.model == 'dptnet' else 'Sepformer'} 2spk 8 kHz W8A8 QAT step, B={a.B}, T={a.T}", "ms_per_step": round(ms, 3),
                      "samples_per_s": round(a.B / ms * 1e3, 2), "observer_phase_ms_per_step": round(obs_ms, 1),
                      "launch": "eager" if a.no_graph else "hipGraph replay", "loss_db": round(float(r["loss"]), 4),
                      "params": sum(p.numel() for p in model.parameters())}))


if __name__ == "__main__":
    main()
