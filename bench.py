#!/usr/bin/env python3
"""bench.py -- QAT-step throughput of ConvTasNet-2spk 8 kHz W8A8 on MI355X (BASELINE.json metric).

One "step" = student quantized fwd + float-teacher fwd + SDR-weighted KD loss (PIT) + backward +
(gradient all-reduce at N>1) + global-norm clip 5.0 + Adam, on one batch of 8 x 4 s synthetic
2-speaker mixtures per GPU (cfg 2 of BASELINE.json), in the QUANTIZING phase (the 50-call observer
phase is passed in an untimed calibration before the warm-up).  Inputs are resident in HBM.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed live) and `cpu_baseline` (oracle/ = CPU port of the reference path, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

B_PER_GPU, T_SAMPLES = 8, 32000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def dominant_kernel_roofline(dev):
    """Times the dominant kernel of the step live with HIP events on the launch stream.

    Dominant kernel (profiles/r01_step_stats.csv): the fp32-MFMA pointwise-conv GEMM k_gemm_f32 --
    24 blocks x (1x 128->512 + 2x 512->128) fwd for student and teacher, the same again as dgrad, plus
    wgrad.  Timed here on its most frequent launch: z[8][512][3999] = W[512x128] * x[8][128][3999].
    Algorithmic work per launch: 2*Co*Ci*B*M flop; algorithmic bytes 4*(Ci+Co)*B*M (SURVEY §8(d)
    convention: each LayerQ boundary tensor moves once)."""
    from fqss_amd import kernels as K
    B, Ci, Co, M = B_PER_GPU, 128, 512, (T_SAMPLES - 16) // 8 + 1
    x = K.empty_act((B, Ci, M), dev).normal_()
    w = torch.randn(Co, Ci, 1, device=dev) * 0.1
    bias = torch.randn(Co, device=dev)
    for _ in range(3):
        K.pwconv_fwd(x, w, bias)
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.pwconv_fwd(x, w, bias)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * Co * Ci * B * M
    abytes = 4.0 * (Ci + Co) * B * M
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel": "k_gemm_f32 (pwconv_fwd 128->512, B=8, M=3999)", "bound": "mfma", "achieved": round(tf, 2),
            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
            "traffic": None, "launch_ms": round(ms, 4), "algorithmic_GBps": round(abytes / (ms * 1e-3) / 1e9, 1)}


def cpu_baseline(threads):
    """oracle/ (CPU port of the reference path, fqss_oracle.Trainer) on the host cores of THIS box:
    bounded sample = full ConvTasNet, quantizing phase, batch 2 x 4 s, 1 warm + 2 timed steps."""
    import oracle.fqss_oracle as O
    from fqss_amd.smoke import build_pair
    n = threads or min(32, os.cpu_count() or 1)
    torch.set_num_threads(n)
    model, fmodel = build_pair("cpu", 0, n_spks=2, kernel_size=16, stride=8)
    s = O.StudentConvTasNetQ(model.state_dict())
    t = O.TeacherConvTasNet(fmodel.state_dict())
    tr = O.Trainer(s, t)
    x, tgt = O.synth_batch(2, T_SAMPLES, seed=0)
    tr.step(x, tgt)                 # observer step: sets every range from data
    s.leave_observer_phase()
    tr.step(x, tgt)                 # warm
    t0 = time.perf_counter()
    k = 2
    for _ in range(k):
        tr.step(x, tgt)
    dt = (time.perf_counter() - t0) / k
    return {"value": round(2 / dt, 4), "unit": "samples/s", "cores": n, "kind": "port",
            "sample": f"full ConvTasNetQ QAT step, quantizing phase, batch 2 x 4 s, {k} timed steps ({dt:.2f} s/step), torch CPU fp32"}


def main():
    a = parse()
    assert torch.cuda.is_available(), "bench.py needs ROCm GPUs"
    from fqss_amd.data import synth_batch
    from fqss_amd.parallel import Comm
    from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
    from fqss_amd.runtime import KDTrainStep
    from fqss_amd.smoke import build_pair

    comm = Comm.from_env("cuda")
    assert comm.world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={comm.world}"
    ldev = comm.local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(ldev)
    dev = torch.device("cuda", ldev)

    model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)   # same init on every rank
    x, tgt = synth_batch(B_PER_GPU, T_SAMPLES, seed=100 + comm.rank, device=dev)   # per-rank shard (weak scaling)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm)

    # untimed calibration: leave the 50-call observer phase (1 full step + 49 observer forwards), then
    # every timed step runs the quantizers
    step(x, tgt)
    with torch.no_grad():
        for _ in range(49):
            model(x)
    assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))

    step(x, tgt)                       # first quantizing step, eager (starts the activation ranges' Adam clocks)
    launch = "eager"
    if not a.no_graph:
        step.capture(x, tgt)           # whole step -> hipGraphs; every later call is a replay
        launch = "hipGraph replay"
        if comm.world > 1:
            # RCCL between two graph replays cannot be exercised on the 1-GPU dev box: self-calibrate (untimed)
            # and keep whichever launch mode is faster on THIS node; all ranks take the same decision.
            t = []
            for mode in (True, False):
                step.use_graph = mode
                step(x, tgt)
                comm.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(2):
                    step(x, tgt)
                torch.cuda.synchronize()
                d = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                comm.all_reduce_max(d)
                t.append(d.item())
            step.use_graph = t[0] <= t[1]
            launch = "hipGraph replay" if step.use_graph else "eager (graph replay slower on this node)"
    for _ in range(a.warmup):
        step(x, tgt)
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step(x, tgt)
    torch.cuda.synchronize()
    comm.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    comm.all_reduce_max(dt)
    dt = dt.item()
    sisdr = r["sisdr"].mean().reshape(1).double()
    comm.all_reduce_sum(sisdr)

    if comm.rank == 0:
        ms = dt / a.steps * 1e3
        out = {
            "metric": "QAT-step samples/sec + SI-SDR, ConvTasNet 2spk 8kHz W8A8",
            "value": round(B_PER_GPU * comm.world * a.steps / dt, 3), "unit": "samples/s",
            "n_gpus": comm.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ConvTasNet 2spk 8 kHz W8A8 QAT step (cfg 2), batch 8 x 4 s per GPU, quantizing phase",
                       "global_batch": B_PER_GPU * comm.world, "segment_samples": T_SAMPLES, "parallelism": f"dp{comm.world}",
                       "kd_lambda": 0.1, "optimizer": "adam lr 1e-3 + clip 5.0",
                       "launch": launch},
            "si_sdr_db": round(sisdr.item() / comm.world, 4), "loss_db": round(r["loss"].item(), 4),
            "step_algorithmic_GB": 74.8,
            "step_algorithmic_frac_of_hbm_peak": round(74.8 / (ms * 1e-3) / HBM_PEAK_GBS, 4),
        }
        out["roofline"] = dominant_kernel_roofline(dev)
        if comm.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_threads)
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
