#!/usr/bin/env python3
"""bench.py -- QAT-step throughput of ConvTasNet-2spk 8 kHz W8A8 on MI355X (BASELINE.json metric).

One "step" = student quantized fwd + float-teacher fwd + SDR-weighted KD loss (PIT) + backward +
(gradient all-reduce at N>1) + global-norm clip 5.0 + Adam, on one batch of 8 x 4 s synthetic
2-speaker mixtures per GPU (cfg 2 of BASELINE.json), in the QUANTIZING phase (the 50-call observer
phase is passed in an untimed calibration before the warm-up).  Inputs are resident in HBM.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed live) and `cpu_baseline` (oracle/ = CPU port of the reference path, rank 0, N=1).

`--workload cfg3` / `cfg4` / `cfg5` time the other built configurations of BASELINE.json the same way (DPTNet 2spk, 1 x 3 s per
GPU; Sepformer 2spk, 1 x 4 s per GPU; HTDemucs 4 x 10 s stereo 44.1 kHz per GPU: SURVEY.md §8 rows a13 / a14 / a15) -- same step, same
JSON shape.  The default run (cfg 2, N = 1) keeps cfg 2 as the headline -- it is the configuration the metric is quoted on -- and
APPENDS those three legs to the same JSON line as `other_workloads` (5 warm-up + 10 timed replays each, same process, each model
freed before the next), so that the driver's own run carries all four workloads.

Every `roofline` object is priced by fqss_amd/roofline_cases.priced(): floor = max(ISSUED flops / dense peak of the matrix dtype the
kernel executes (bf16 2.5 PF, i8 5 PF; fp32 vector 157.3 TF), algorithmic bytes / 8 TB/s); frac = floor / measured time (<= 1).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-teacher-ahead", action="store_true",
                    help="run the frozen teacher beside the student's forward of the SAME step (round-2 schedule) instead of one batch ahead")
    ap.add_argument("--workload", default="cfg2", choices=("cfg2", "cfg3", "cfg4", "cfg5", "infer"),
                    help="cfg2 ConvTasNet 8 x 4 s (default, the metric's configuration); cfg3 DPTNet 1 x 3 s; cfg4 Sepformer 1 x 4 s; "
                         "cfg5 HTDemucs 4 x 10 s stereo 44.1 kHz")
    ap.add_argument("--no-det-leg", action="store_true", help="skip the FQSS_DETERMINISTIC=1 leg (`deterministic_ms_per_step`)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="cfg2 at N=1 only: skip the cfg 3 / 4 / 5 legs that the default run appends as `other_workloads`")
    ap.add_argument("--other-steps", type=int, default=10, help="timed replays of each `other_workloads` leg")
    ap.add_argument("--other-warmup", type=int, default=5, help="warm-up replays of each `other_workloads` leg")
    ap.add_argument("--hd-batch", type=int, default=4, help="cfg5: samples per GPU (htdemucs.yaml: 32 over 8 GPUs)")
    ap.add_argument("--hd-seconds", type=float, default=10.0, help="cfg5: segment length in seconds (htdemucs.yaml: 10)")
    return ap.parse_args()


def dominant_kernel_roofline(dev, ms_step):
    """Times the step's kernels live at their cfg-2 launch shapes (fqss_amd/roofline_cases.py: one case = ONE kernel launch, operands
    rotating over > 256 MiB) with HIP events on the launch stream.  `roofline` = the SINGLE kernel (one kernel symbol, launch-weighted
    over its shapes in the step) with the largest time per step; the two multi-launch gLN-backward operations are timed too but only
    listed (`group_of_launches`).  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (raw
    counters: 2 x FETCH_SIZE + WRITE_SIZE per the gfx950 rule of MI355X_MICROARCH.md; `traffic_x1` = FETCH_SIZE + WRITE_SIZE); null when
    no measurement of that kernel is committed.  Also returns the step-level traffic figures."""
    from fqss_amd import roofline_cases as RC
    cases = RC.build(dev)
    times = [RC.time_case(c) for c in cases]
    groups = RC.summarize(cases, times)
    pmc, pmc_file = {}, None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):       # the newest committed PMC table
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc, pmc_file = json.load(f).get("per_launch_bytes", {}), "profiles/" + name
            break
        except (OSError, ValueError):
            continue

    def obj(g, shapes):
        t = pmc.get(g["kernel"])
        o = RC.roofline_object(g, traffic=t["x2"] if t else None)
        o["traffic_x1"] = t["x1"] if t else None
        if not shapes:
            o.pop("shapes")
        return o
    singles = [g for g in groups if not g["group"]]
    dom = obj(singles[0], True)
    near = [g["kernel"] for g in singles[1:] if g["ms_step"] >= 0.97 * singles[0]["ms_step"]]
    if near:
        dom["tie_break"] = ("single kernels within 3 % of the largest time per step: " + ", ".join(near) +
                            "; the one with the largest isolated time in THIS run is reported")
    others = [obj(g, False) for g in groups if g is not singles[0]]
    missing = [g["kernel"] for g in groups if g["kernel"] not in pmc]
    step_bytes = RC.step_traffic_bytes(cases)
    step = {"step_traffic_GB": round(step_bytes / 1e9, 2),
            "step_traffic_frac_of_hbm_peak": round(step_bytes / (ms_step * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
            "isolated_kernel_ms_sum": round(sum(g["ms_step"] for g in groups), 3),
            "pmc_source": pmc_file, "pmc_missing_kernels": missing}
    return dom, others, step


def _host_memory_gb():
    """memory this process may use: the cgroup limit when there is one, else MemAvailable"""
    lim = None
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(path).read().strip()
            if v.isdigit() and int(v) < (1 << 60):
                lim = int(v) / 1e9
                break
        except OSError:
            pass
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024 / 1e9
    except OSError:
        pass
    vals = [v for v in (lim, avail) if v is not None]
    return min(vals) if vals else 0.0


def cpu_baseline(threads):
    """oracle/ (CPU port of the reference path, fqss_oracle.Trainer) on the host cores of THIS box, full ConvTasNet, quantizing phase:
    ONE step at the metric's own batch of 8 x 4 s when the host has the memory for it (a B = 8 oracle step holds 45 GB and takes
    ~27 s), else 3 timed steps of the bounded sample, batch 2 x 4 s; ~30 s of CPU work either way."""
    import oracle.fqss_oracle as O
    from fqss_amd.smoke import build_pair
    n = threads or min(32, os.cpu_count() or 1)
    torch.set_num_threads(n)
    mem = _host_memory_gb()
    Bc = B_PER_GPU if mem >= 96.0 else 2
    model, fmodel = build_pair("cpu", 0, n_spks=2, kernel_size=16, stride=8)
    s = O.StudentConvTasNetQ(model.state_dict())
    t = O.TeacherConvTasNet(fmodel.state_dict())
    tr = O.Trainer(s, t)
    xs, ts = O.synth_batch(2, T_SAMPLES, seed=0)
    tr.step(xs, ts)                 # observer step (batch 2): sets every range from data
    s.leave_observer_phase()
    tr.step(xs, ts)                 # warm (batch 2): optimizer state, allocator
    if Bc == 2:
        k = 3
        t0 = time.perf_counter()
        for _ in range(k):
            tr.step(xs, ts)
        dt = (time.perf_counter() - t0) / k
        what = f"batch 2 x 4 s (bounded: host memory {mem:.0f} GB; a batch-8 oracle step holds 45 GB), {k} timed steps"
    else:
        # ONE timed step at the metric's batch: it takes ~27 s on 32 cores (the oracle scales badly with the batch: 0.9 samples/s at
        # batch 2, 0.3 at batch 8 -- BASELINE.md 2 measured the same on the survey's host); observer and warm-up steps ran at batch 2
        x, tgt = O.synth_batch(Bc, T_SAMPLES, seed=1)
        k = 1
        t0 = time.perf_counter()
        tr.step(x, tgt)
        dt = time.perf_counter() - t0
        what = f"batch {Bc} x 4 s (the metric's batch), {k} timed step after an observer and a warm-up step at batch 2"
    return {"value": round(Bc / dt, 4), "unit": "samples/s", "cores": n, "kind": "port",
            "sample": f"full ConvTasNetQ QAT step, quantizing phase, {what} ({dt:.2f} s/step), torch CPU fp32"}


DUALPATH = {"cfg3": dict(name="DPTNet", cfg={"name": "DPTNet", "n_src": 2, "kernel_size": 2}, T=24000, lr=4e-4,
                         gemm=(64, 1024, "LSTM input projection (both directions)")),
            "cfg4": dict(name="Sepformer", cfg={"name": "Sepformer", "n_src": 2, "kernel_size": 16, "stride": 8}, T=32000, lr=1.5e-4,
                         gemm=(256, 1024, "feed-forward 256 -> 1024"))}


def _time_launches(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def _pmc_other(key, shape_token):
    """HBM bytes per launch (2 FETCH_SIZE + WRITE_SIZE, separate --pmc passes) of a cfg 3 / 4 / 5 roofline kernel from the newest committed
    profiles/r*_pmc_traffic_cfg345.json (tools/roofline_probe.py --set other); the case must have been measured at THIS shape
    (`shape_token` occurs in its label), else None"""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic_cfg345.json")), reverse=True):
        try:
            cases = json.load(open(path))["cases"]
        except (OSError, ValueError, KeyError):
            continue
        for c in cases:
            if c.get("kernel") == key and shape_token in c.get("label", ""):
                return int(c["traffic_x2"])
    return None


def dominant_kernel_roofline_dualpath(which, rows, Ci, Co, what, seqs):
    """`roofline` of a dual-path workload = the kernel with the largest time per step in the committed steady-state table
    (profiles/r*_cfg3_step_table.txt: k_lstm_fwd_st<128>; profiles/r*_cfg4_step_table.txt: the coded weight-gradient instance of
    k_gemm_x3), timed live at this workload's shape with HIP events on torch's current stream (the stream these launches go to) and
    priced by roofline_cases.priced(): the LSTM recurrence is fp32 FMA on the vector ALU (157.3 TF), the coded weight gradient issues
    three bf16 MFMA products per term (2.5 PF dense)."""
    from fqss_amd import kernels as K
    from fqss_amd import roofline_cases as RC
    if which == "cfg3":
        S, Bq, H = seqs[0], seqs[1], 128            # intra-chunk BiLSTM: 250 steps x (chunks) sequences, both directions in one launch
        pre = torch.randn(S, Bq, 8 * H, device="cuda") * 0.1
        whh, bhh = torch.randn(2, 4 * H, H, device="cuda") * 0.05, torch.zeros(2, 4 * H, device="cuda")
        us = _time_launches(lambda: K.lstm_fwd(pre, whh, bhh, S, Bq, H), 10)
        flops = 2.0 * 2 * (4 * H) * H * S * Bq       # the recurrent product h W_hh^T of both directions
        n = S * Bq
        nbytes = 4.0 * n * 8 * H + 4.0 * n * (2 * H + 8 * H + 4 * H)      # input projection read; h, gates, cell states written (roofline_cases.build_other)
        o = {"kernel": "k_lstm_fwd_st<128>", "what": "BiLSTM recurrence of the intra-chunk path (both directions)", "shape": [S, Bq, H],
             "launch_us": round(us, 1), "launches_per_step": 24, "algorithmic_bytes_per_launch": int(nbytes)}
        o.update(RC.priced(flops, 1, "f32", nbytes, us))
        o["traffic"] = _pmc_other("k_lstm_fwd_st<128>", f"{S} steps x {Bq} sequences")
        o["note"] = ("a chain of 250 dependent time steps on 194 of 256 CUs: bound by the issue of ~360 vector instructions per wave and step "
                     "(256 of them the FMAs) + 2 barriers per step (docs/history/DESIGN_rounds_1-5.md 7, 9); neither roofline binds it")
        return o
    # cfg 4 (profiles/r05_cfg4_step_table.txt): since round 5 the coded weight gradients run as four grouped launches (2.0 ms per step
    # together) and the kernel with the largest time per step is the coded DATA gradient of the student's linears,
    # k_gemm_x3<true, false, false, 1, 2, 2> (fqss_qrow_bwd_x): A = gz (fp32, three bf16 pieces, scaled by delta_w), B = the weight's int8
    # codes (one exact plane): three products per k.  Timed at its heaviest shape, the feed-forward's 256 -> 1024 linear
    gz = torch.randn(rows, Co, device="cuda")
    wc = K.wq_codes(torch.randn(Co, Ci, 1, device="cuda") * 0.05, -torch.ones(Co, 1, 1, device="cuda") * 0.2, torch.ones(Co, 1, 1, device="cuda") * 0.2)
    us = _time_launches(lambda: K.qrow_bwd_x(gz, wc))
    nbytes = 4.0 * rows * Co + Co * Ci + 4.0 * rows * Ci
    o = {"kernel": "k_gemm_x3<true, false, false, 1, 2, 2> (fqss_qrow_bwd_x)", "what": "data gradient of the " + what + " from the weight's int8 codes",
         "shape": [rows, Ci, Co], "launch_us": round(us, 1), "launches_per_step": 96, "algorithmic_bytes_per_launch": int(nbytes)}
    o.update(RC.priced(2.0 * rows * Ci * Co, 3, "bf16", nbytes, us))
    o["traffic"] = _pmc_other("k_gemm_x3", f"dgrad {rows} x {Co} -> {Ci}")
    # ... and the grouped weight gradients beside it: one launch = the four linears of 8 transformer layers (32 jobs)
    q = K.RowWgradQueue()
    lo, hi = torch.tensor([-1.0], device="cuda"), torch.tensor([1.0], device="cuda")
    shapes = [(256, 768), (256, 256), (256, 1024), (1024, 256)] * 8
    ops_ = [(torch.randn(rows, co, device="cuda"), torch.randint(0, 256, (rows, ci), device="cuda", dtype=torch.uint8), torch.zeros(co, ci, device="cuda"),
             torch.zeros(co, device="cuda")) for ci, co in shapes]

    def grouped():
        for a, c, gw_, gb_ in ops_:
            q.push(a, c, lo, hi, gw_, gb_)
        q.flush()
    ug = _time_launches(grouped, 5)
    gb = sum(4.0 * rows * co + rows * ci + 4.0 * co * ci for ci, co in shapes)
    og = {"kernel": "k_gemm_x3_wq_multi<2, 2> (fqss_qrow_bwd_w_group)", "what": "coded weight gradients of 32 linears (8 transformer layers) in one launch",
          "launch_us": round(ug, 1), "launches_per_step": 4, "algorithmic_bytes_per_launch": int(gb)}
    og.update(RC.priced(sum(2.0 * rows * ci * co for ci, co in shapes), 3, "bf16", gb, ug))
    o["other_kernels"] = [og]
    o["note"] = "bound by vector-ALU issue of the operand split and by the latency of a k-tile (docs/history/DESIGN_rounds_1-5.md 7e (4), (7)), not by either roofline"
    return o


def cpu_baseline_dualpath(which, model, fmodel, lr, T):
    """oracle/ on the host cores, bounded: ONE quantizing-phase step on a 0.5 s excerpt (the oracle's LSTM / attention are
    Python-level torch loops: a full 3-4 s step takes minutes), scaled to the workload's segment length"""
    import oracle.fqss_oracle as O
    from fqss_amd.data import synth_batch
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    fsd = {k: v.detach().cpu() for k, v in fmodel.state_dict().items()}
    if which == "cfg3":
        import oracle.dptnet_oracle as D
        s_o, t_o = D.StudentDPTNetQ(sd), D.TeacherDPTNet(fsd)
    else:
        import oracle.sepformer_oracle as S
        s_o, t_o = S.StudentSepformerQ(sd), S.TeacherSepformer(fsd)
    s_o.leave_observer_phase()
    cores = min(16, len(os.sched_getaffinity(0)))      # the GPU box's CPU share; more threads oversubscribe the tiny ops
    torch.set_num_threads(cores)
    tr = O.Trainer(s_o, t_o, lr=lr)
    tr.step(*synth_batch(1, 800, seed=1))               # warm-up
    Tc = min(T, 4000)
    x, tgt = synth_batch(1, Tc, seed=0)
    t0 = time.perf_counter()
    tr.step(x, tgt)
    sec = time.perf_counter() - t0
    return {"value": round((Tc / T) / sec, 4), "unit": "samples/s", "cores": cores, "kind": "port", "extrapolated": Tc != T,
            "sample": f"EXTRAPOLATED: one full QAT step of the oracle on a 1 x {Tc}-sample excerpt ({sec:.1f} s), scaled linearly by {Tc}/{T} "
                      f"to the workload's segment length (a full-length oracle step takes minutes), torch CPU fp32"}


def main_dualpath(a, comm=None):
    """cfg 3 / cfg 4: one sample per GPU (the shipped per-GPU batch), same step and timing protocol as cfg 2.  With `comm` given (the
    `other_workloads` legs of the default run) the JSON object is RETURNED instead of printed and the communicator stays open."""
    import copy
    from fqss_amd.data import synth_batch
    from fqss_amd.kernels import dp_chunks
    from fqss_amd.parallel import Comm, local_device
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
    from fqss_amd.runtime import KDTrainStep
    from fqss_amd.smoke import QCFG
    W = DUALPATH[a.workload]
    leg = comm is not None
    comm = comm or Comm.from_env("cuda")
    assert comm.world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={comm.world}"
    ldev = local_device(comm.local_rank)
    torch.cuda.set_device(ldev)
    dev = torch.device("cuda", ldev)
    torch.manual_seed(0)                                    # same init on every rank
    model = create_model(dict(W["cfg"]))
    fmodel = copy.deepcopy(model).to(dev).eval()
    model = quantize_model(model, dict(QCFG)).to(dev).train()
    T = W["T"]
    x, tgt = synth_batch(1, T, seed=100 + comm.rank, device=dev)
    x2, tgt2 = synth_batch(1, T, seed=200 + comm.rank, device=dev)      # the timed loop alternates two batches (teacher look-ahead, as cfg 2)
    X, TG = (x, x2), (tgt, tgt2)
    ahead = not a.no_teacher_ahead
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=W["lr"], clip=5.0, comm=comm, teacher_ahead=ahead)
    step(x, tgt)                                            # untimed calibration: the 50-call observer phase
    with torch.no_grad():
        for _ in range(49):
            model(x)
    assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))
    step(x, tgt)
    launch = "eager"
    if not a.no_graph:
        step.capture(x, tgt)
        launch = "hipGraph replay"
        if comm.active:
            launch = f"hipGraph replay, {len(step._graphs[0])} backward segments, bucketed all-reduce overlapped"
    if ahead and not a.no_graph:
        launch += "; teacher forward of batch n+1 as its own hipGraph on a second stream beside step n"
    it = 0
    for _ in range(a.warmup):
        step(X[it & 1], TG[it & 1], x_next=X[(it + 1) & 1])
        it += 1
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step(X[it & 1], TG[it & 1], x_next=X[(it + 1) & 1])
        it += 1
    torch.cuda.synchronize()
    comm.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    comm.all_reduce_max(dt)
    dt = dt.item()
    sisdr = r["sisdr"].mean().reshape(1).double()
    comm.all_reduce_sum(sisdr)
    if comm.rank == 0:
        ms = dt / a.steps * 1e3
        L = (T - 1) if a.workload == "cfg3" else (T - 16) // 8 + 1
        rows = 250 * dp_chunks(L, 250)[1]
        out = {"metric": f"QAT-step samples/sec + SI-SDR, {W['name']} 2spk 8kHz W8A8", "value": round(comm.world * a.steps / dt, 3),
               "unit": "samples/s", "n_gpus": comm.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{W['name']} 2spk 8 kHz W8A8 QAT step ({a.workload}), batch 1 x {T // 8000} s per GPU, quantizing phase",
                          "global_batch": comm.world, "segment_samples": T, "parallelism": f"dp{comm.world}", "kd_lambda": 0.1,
                          "optimizer": f"adam lr {W['lr']:g} + clip 5.0", "launch": launch},
               "si_sdr_db": round(sisdr.item() / comm.world, 4), "loss_db": round(r["loss"].item(), 4),
               "params": sum(p.numel() for p in model.parameters()),
               "roofline": dominant_kernel_roofline_dualpath(a.workload, rows, *W["gemm"], seqs=(250, dp_chunks(L, 250)[1]))}
        if comm.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_dualpath(a.workload, model, fmodel, W["lr"], T)
        if leg:
            return out
        print(json.dumps(out), flush=True)
    comm.barrier()
    if not leg:
        comm.close()


def _events_us(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def htdemucs_roofline(B, nh, L, hd):
    """cfg 5: the kernel with the largest time per step in the committed table (profiles/r03_cfg5_step_table.txt) is k_qgemm<3, 2>, the
    six-product pointwise GEMM of the general conv layers (67 launches over ~20 shapes): timed live at its heaviest shape -- the
    level-0 rewrite conv of the waveform branch, 4 x (48 x 3 -> 96) x 110250 frames -- with HIP events on torch's current stream;
    the streaming attention core (round 2's roofline kernel) in its two forms beside it"""
    from fqss_amd import kernels as K
    Kk, Co, M = 144, 96, 110250
    f = K.empty_act((B, Kk, M), "cuda")
    f.normal_()
    w, b = torch.randn(Co, Kk, 1, device="cuda") * 0.1, torch.zeros(Co, device="cuda")
    us = _events_us(lambda: K.pwconv_fwd(f, w, b, six=True))
    from fqss_amd import roofline_cases as RC
    by = 4.0 * B * M * (Kk + Co)
    E = nh * hd
    q, k, v = (torch.randn(B, L, E, device="cuda") * 0.3 for _ in range(3))
    ua = _events_us(lambda: K.attn_long_fwd(q, k, v, nh, True))
    qc, kc, vc = (torch.randint(0, 256, (B, L, E), device="cuda", dtype=torch.uint8) for _ in range(3))
    rng = [(torch.tensor([-0.9], device="cuda"), torch.tensor([0.8], device="cuda")) for _ in range(3)]
    uc = _events_us(lambda: K.attn_long_fwd_c(qc, kc, vc, rng, nh, True))
    fa = 4.0 * L * L * hd * B * nh
    na = B * L * E
    o = {"kernel": "k_qgemm<3, 2> (fqss_pwconv_fwd_x3s)", "what": "pointwise GEMM over the frames of the level-0 rewrite conv (48 x 3 -> 96)",
         "shape": [B, Kk, Co, M], "launch_us": round(us, 1), "algorithmic_bytes_per_launch": int(by)}
    o.update(RC.priced(2.0 * B * Co * Kk * M, 6, "bf16", by, us))
    o["traffic"] = _pmc_other("k_qgemm<3>", f"{B} x (48 x 3 -> {Co}) x {M}")
    oa = {"kernel": "k_attn_long_fwd_x3<%d, false>" % hd, "what": "self-attention of the spectrogram branch, float operands (teacher)",
          "shape": [B, nh, L, hd], "launch_us": round(ua, 1)}
    oa.update(RC.priced(fa, 6, "bf16", 4.0 * 3 * na + 4.0 * na + 8.0 * B * nh * L, ua))
    oc = {"kernel": "k_attn_long_fwd_c<%d>" % hd, "what": "the same on the u8 codes of q, k, v (student)", "shape": [B, nh, L, hd], "launch_us": round(uc, 1)}
    oc.update(RC.priced(fa, 3, "bf16", 3.0 * na + 4.0 * na + 8.0 * B * nh * L, uc))
    o["other_kernels"] = [oa, oc]
    o["note"] = ("fp32-grade arithmetic executed as exact bf16 partial products (6 per term for float operands, 3 on codes): priced on the "
                 "issued products against the 2.5 PF dense bf16 peak")
    return o


def cpu_baseline_htdemucs(model, fmodel, B, T):
    """oracle/htdemucs_oracle.py on the host cores, bounded: ONE quantizing-phase step (student fwd + bwd, teacher fwd, loss) on a
    1 x 1 s excerpt, scaled by the excerpt's share of the workload's samples"""
    import oracle.htdemucs_oracle as H
    cores = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    fsd = {k: v.detach().cpu() for k, v in fmodel.state_dict().items()}
    kw = dict(n_src=model.n_srcs, audio_channels=model.audio_channels, nfft=model.nfft, depth=model.depth,
              t_layers=model.crosstransformer.num_layers, t_heads=model.crosstransformer.layers[0].self_attn.mha.num_heads,
              bottom=bool(model.bottom_channels))
    s_o, t_o = H.HTDemucsOracle(sd, quantized=True, **kw), H.HTDemucsOracle(fsd, quantized=False, **kw)
    s_o.leave_observer_phase()
    Tc = min(T, 44100)
    g = torch.Generator().manual_seed(0)
    src = torch.randn(1, model.n_srcs, model.audio_channels, Tc, generator=g) * 0.1
    mix = src.sum(1)
    t0 = time.perf_counter()
    with torch.no_grad():
        fest = t_o.forward(mix)
    est = s_o.forward(mix)
    loss = H.solver_loss(est, fest, src)[0]
    loss.backward()
    sec = time.perf_counter() - t0
    return {"value": round((Tc / T) / sec, 5), "unit": "samples/s", "cores": cores, "kind": "port", "extrapolated": Tc != T,
            "sample": f"EXTRAPOLATED: one QAT step of the oracle (student fwd + bwd, teacher fwd, loss; no optimizer) on a 1 x {Tc}-sample "
                      f"excerpt ({sec:.1f} s), scaled linearly by {Tc}/{T} to the workload's segment length, torch CPU fp32"}


def main_htdemucs(a, comm=None):
    """cfg 5: HTDemucs, stereo 44.1 kHz, 4 sources, the shipped per-GPU batch (32 / 8 GPUs) x 10 s; step = student fwd + teacher
    fwd + solver loss + bwd (+ all-reduce) + Adam (htdemucs.yaml: lr 3e-4, no clipping), same timing protocol as cfg 2; `comm`: as
    main_dualpath"""
    import copy
    from fqss_amd.parallel import Comm, local_device
    from fqss_amd.quantization.qat.models.load_model import quantize_model
    from fqss_amd.quantization.qat.models.htdemucsq import HTDemucsQ
    from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
    from fqss_amd.runtime import KDTrainStep
    leg = comm is not None
    comm = comm or Comm.from_env("cuda")
    assert comm.world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={comm.world}"
    ldev = local_device(comm.local_rank)
    torch.cuda.set_device(ldev)
    dev = torch.device("cuda", ldev)
    torch.manual_seed(0)
    B, T = a.hd_batch, int(round(a.hd_seconds * 44100))
    model = HTDemucsQ(sources=["drums", "bass", "other", "vocals"], bottom_channels=512, segment=a.hd_seconds)
    fmodel = copy.deepcopy(model).to(dev).eval()
    qcfg = dict(qat=True, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, in_quant=False,
                in_act_n_bits=8, out_quant=True, out_act_n_bits=8, n_splitter=2, n_combiner=2, observer=True)
    model = quantize_model(model, qcfg).to(dev).train()
    g = torch.Generator().manual_seed(42 + comm.rank)
    src = (torch.randn(B, 4, 2, T, generator=g) * 0.1).to(dev)            # synthetic stereo Gaussian stems
    mix = src.sum(1)
    src2 = (torch.randn(B, 4, 2, T, generator=g) * 0.1).to(dev)           # second batch: the timed loop alternates the two (teacher look-ahead)
    MIX, SRC = (mix, src2.sum(1)), (src, src2)
    ahead = not a.no_teacher_ahead
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=3e-4, clip=0.0, comm=comm, loss="l1_sdr", teacher_ahead=ahead)
    step(mix, src)                                          # untimed calibration: the 50-call observer phase
    with torch.no_grad():
        for _ in range(49):
            model(mix)
    assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))
    step(mix, src)
    launch = "eager"
    if not a.no_graph:
        step.capture(mix, src)
        launch = "hipGraph replay"
        if comm.active:
            launch = f"hipGraph replay, {len(step._graphs[0])} backward segments, bucketed all-reduce overlapped"
    if ahead and not a.no_graph:
        launch += "; teacher forward of batch n+1 as its own hipGraph on a second stream beside step n"
    it = 0
    for _ in range(a.warmup):
        step(MIX[it & 1], SRC[it & 1], x_next=MIX[(it + 1) & 1])
        it += 1
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step(MIX[it & 1], SRC[it & 1], x_next=MIX[(it + 1) & 1])
        it += 1
    torch.cuda.synchronize()
    comm.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    comm.all_reduce_max(dt)
    dt = dt.item()
    if comm.rank == 0:
        ms = dt / a.steps * 1e3
        Fr, le = 8, -(-T // 1024)
        out = {"metric": "QAT-step samples/sec, HTDemucs 4 stems stereo 44.1kHz W8A8", "value": round(comm.world * B * a.steps / dt, 3),
               "unit": "samples/s", "n_gpus": comm.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"HTDemucs 4-stem stereo 44.1 kHz W8A8 QAT step (cfg5), batch {B} x {a.hd_seconds:g} s per GPU, "
                                      "quantizing phase, bottom_channels 512", "global_batch": comm.world * B, "segment_samples": T,
                          "parallelism": f"dp{comm.world}", "kd_lambda": 0.1, "optimizer": "adam lr 0.0003, no clipping", "launch": launch},
               "loss": round(r["loss"].item(), 6), "params": sum(p.numel() for p in model.parameters()),
               "roofline": htdemucs_roofline(B, 8, Fr * le, 64)}
        if comm.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_htdemucs(model, fmodel, B, T)
        if leg:
            return out
        print(json.dumps(out), flush=True)
    comm.barrier()
    if not leg:
        comm.close()


def main_infer(a):
    """quantized ConvTasNet inference (eval mode, codes-only dataflow, one hipGraph per request shape) on the cfg 2 batch: 8 x 4 s"""
    from fqss_amd.data import synth_batch
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.runtime import InferRunner
    from fqss_amd.smoke import QCFG
    assert a.gpus == 1, "the inference probe is a single-GPU measurement"
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    model = quantize_model(create_model({"name": "ConvTasNet", "n_src": 2, "kernel_size": 16, "stride": 8}), dict(QCFG)).cuda().train()
    x, _ = synth_batch(8, 32000, seed=0, device="cuda")
    with torch.no_grad():
        for _ in range(50):
            model(x)                                        # observer calibration
    run = InferRunner(model, use_graph=not a.no_graph)
    for _ in range(max(a.warmup, 1)):
        y = run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = run(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / a.steps * 1e3
    fwd_bytes = 74.8e9 / 4.0                                  # one forward = a quarter of the step's algorithmic bytes
    print(json.dumps({"metric": "quantized inference samples/sec, ConvTasNet 2spk 8kHz W8A8", "value": round(8 * a.steps / dt, 2), "unit": "samples/s",
                      "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "u8/f32", "data": "synthetic",
                      "config": {"workload": "ConvTasNet 2spk 8 kHz W8A8 inference (eval mode, codes-only dataflow), batch 8 x 4 s",
                                 "launch": "eager" if a.no_graph else "hipGraph replay"},
                      "roofline": {"kernel": "whole forward", "bound": "hbm", "achieved": round(fwd_bytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(fwd_bytes / (ms * 1e-3) / 8e12, 4), "traffic": None},
                      "out_rms": round(float(y.pow(2).mean().sqrt()), 6)}), flush=True)


def exchange_report(step, comm, ldev, run_step, it, reps=10):
    """the `dist` object of the JSON line: what the gradient exchange ran on and what it cost.  Every rank runs
    `reps` more replays (outside the timed region) with two timing events around the optimizer's wait for the communication stream:
    exposed_comm_ms = (all exchanges joined) - (end of the last backward segment), the MAX over ranks of each rank's median.
    Reference: DDP's bucketed all-reduce under pl.Trainer(strategy="ddp") (asteroid_librimix_trainer.py:125-135), distrib.py:30-59."""
    import torch.distributed as dist
    prop = torch.cuda.get_device_properties(ldev)
    mine = {"rank": comm.rank, "local_rank": comm.local_rank, "device": ldev, "name": prop.name,
            "pci_bus_id": getattr(prop, "pci_bus_id", None), "uuid": str(getattr(prop, "uuid", ""))[:18]}
    nseg = len(step._graphs[0]) if step._graphs is not None else 1
    segs = step.segments if (step.segments is not None and nseg > 1) else [(0, step.arena.numel)]
    info = {"backend": comm.backend if comm.active else None, "world": comm.world,
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if comm.active and comm.backend == "nccl" else None,
            "buckets_MB": [round(4e-6 * (hi - lo), 3) for lo, hi in segs][::-1],        # in exchange order (the network's end first)
            "exposed_comm_ms": 0.0, "devices": [mine]}
    if not comm.active or step._graphs is None:
        return info
    step.comm_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ms = []
    for _ in range(reps):
        run_step(it)
        it += 1
        torch.cuda.synchronize()
        ms.append(step.comm_events[0].elapsed_time(step.comm_events[1]))
    step.comm_events = None
    t = torch.tensor([sorted(ms)[len(ms) // 2]], device=torch.device("cuda", ldev), dtype=torch.float64)
    comm.all_reduce_max(t)
    info["exposed_comm_ms"] = round(t.item(), 4)
    devs = [None] * comm.world
    dist.all_gather_object(devs, mine)
    info["devices"] = devs
    if len({(d["pci_bus_id"], d["uuid"], d["device"]) for d in devs}) != comm.world:
        info["warning"] = "two ranks report the same device"
    return info


def deterministic_leg(a, dev, steps=10, warmup=5):
    """`deterministic_ms_per_step`: the same cfg-2 step under FQSS_DETERMINISTIC=1 -- every fp32 gradient atomic an
    integer atomic on a fixed-point shadow, one rounding pass per arena (kernels.DetMode) -- on a fresh model, 10 timed replays"""
    import gc
    from fqss_amd import kernels as K
    from fqss_amd.data import synth_batch
    from fqss_amd.runtime import KDTrainStep
    from fqss_amd.smoke import build_pair
    prev = os.environ.get("FQSS_DETERMINISTIC")
    os.environ["FQSS_DETERMINISTIC"] = "1"
    try:
        model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
        X = [synth_batch(B_PER_GPU, T_SAMPLES, seed=100 + 100 * i, device=dev) for i in range(2)]
        step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, teacher_ahead=not a.no_teacher_ahead)
        assert step.det is not None
        step(*X[0])
        with torch.no_grad():
            for _ in range(49):
                model(X[0][0])
        step(*X[0])
        step.capture(*X[0])
        for it in range(warmup):
            step(*X[it & 1], x_next=X[(it + 1) & 1][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(warmup, warmup + steps):
            step(*X[it & 1], x_next=X[(it + 1) & 1][0])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return {"deterministic_ms_per_step": round(ms, 3), "deterministic_steps": steps}
    finally:
        K.DetMode.off()
        if prev is None:
            os.environ.pop("FQSS_DETERMINISTIC", None)
        else:
            os.environ["FQSS_DETERMINISTIC"] = prev
        gc.collect()
        torch.cuda.empty_cache()


def main():
    a = parse()
    # `python bench.py --gpus N` on its own (no torch.distributed.run around it): start the N ranks here, as the reference's entry points
    # do (asteroid_librimix_trainer.py:125-135, tasnet_musdbhq_trainer.py:17-30) -- BEFORE anything in this process touches the GPU; this
    # parent only waits and hands the first failing rank's exit code on (fqss_amd/launch.py).  Under a launcher (WORLD_SIZE set) this
    # process IS a rank.
    from fqss_amd.launch import already_launched, spawn_ranks
    if a.gpus > 1 and not already_launched():
        sys.exit(spawn_ranks(a.gpus))
    if a.workload == "cfg5":
        return main_htdemucs(a)
    if a.workload == "infer":
        return main_infer(a)
    assert torch.cuda.is_available(), "bench.py needs ROCm GPUs"
    if a.workload != "cfg2":
        return main_dualpath(a)
    from fqss_amd.data import synth_batch
    from fqss_amd.parallel import Comm, local_device
    from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
    from fqss_amd.runtime import KDTrainStep
    from fqss_amd.smoke import build_pair

    comm = Comm.from_env("cuda")
    assert comm.world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={comm.world}"
    ldev = local_device(comm.local_rank)
    torch.cuda.set_device(ldev)
    dev = torch.device("cuda", ldev)

    model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)   # same init on every rank
    x, tgt = synth_batch(B_PER_GPU, T_SAMPLES, seed=100 + comm.rank, device=dev)   # per-rank shard (weak scaling)
    # a second, different batch: the timed loop alternates the two, so the teacher's look-ahead (its forward of batch n+1 runs on the
    # teacher stream beside the whole of step n) works on a mixture that is NOT the one the step is training on
    x2, tgt2 = synth_batch(B_PER_GPU, T_SAMPLES, seed=200 + comm.rank, device=dev)
    X, TG = (x, x2), (tgt, tgt2)
    ahead = not a.no_teacher_ahead
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, teacher_ahead=ahead)

    # untimed calibration: leave the 50-call observer phase (1 full step + 49 observer forwards), then
    # every timed step runs the quantizers
    step(x, tgt)
    with torch.no_grad():
        for _ in range(49):
            model(x)
    assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))

    step(x, tgt)                       # first quantizing step, eager (starts the activation ranges' Adam clocks)
    if os.environ.get("FQSS_MAIN_PRIO"):      # experiment knob: the step's own stream at a higher priority than the teacher's
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["FQSS_MAIN_PRIO"])))
    launch = "eager"
    if not a.no_graph:
        step.capture(x, tgt)           # whole step -> hipGraphs; every later call is a replay
        launch = "hipGraph replay"
        if comm.active:
            # the backward replays as one hipGraph per gradient bucket; each bucket's RCCL all-reduce is launched between two
            # replays on the communication stream and overlaps the next bucket's backward (no collective inside a graph)
            launch = f"hipGraph replay, {len(step._graphs[0])} backward segments, bucketed all-reduce overlapped"
        if ahead:
            launch += "; teacher forward of batch n+1 as its own hipGraph on a second stream beside step n"
    it = 0
    for _ in range(a.warmup):
        step(X[it & 1], TG[it & 1], x_next=X[(it + 1) & 1])
        it += 1
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step(X[it & 1], TG[it & 1], x_next=X[(it + 1) & 1])      # one student fwd + bwd + update AND one teacher forward per step
        it += 1
    torch.cuda.synchronize()
    comm.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    comm.all_reduce_max(dt)
    dt = dt.item()
    sisdr = r["sisdr"].mean().reshape(1).double()
    comm.all_reduce_sum(sisdr)
    dist_info = exchange_report(step, comm, ldev, lambda i: step(X[i & 1], TG[i & 1], x_next=X[(i + 1) & 1]), it)

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
            # SURVEY.md 8(d) convention (4 B x (in + out) of every LayerQ boundary tensor, x4 for bwd + teacher): 74.8 GB/step ...
            "step_algorithmic_GB": 74.8,
            "step_algorithmic_frac_of_hbm_peak": round(74.8 / (ms * 1e-3) / HBM_PEAK_GBS, 4),
        }
        out["dist"] = dist_info
        # ... and the bytes this build actually has to move (every kernel's operands read once / results written once at their real width)
        out["roofline"], out["roofline_other_kernels"], step = dominant_kernel_roofline(dev, ms)
        out.update(step)
        if comm.world == 1 and not a.no_det_leg and not a.no_graph:
            out.update(deterministic_leg(a, dev))
        if comm.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_threads)
        if comm.world == 1 and not a.no_other_workloads:
            # the other three BASELINE configurations in the SAME driver-run line: the cfg-2 model, its graphs and
            # its memory pools are released first, each leg frees its own before the next
            import copy
            import gc
            del step, model, fmodel, r
            legs = []
            for w in ("cfg3", "cfg4", "cfg5"):
                gc.collect()
                torch.cuda.empty_cache()
                torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
                b = copy.copy(a)
                b.workload, b.steps, b.warmup = w, a.other_steps, a.other_warmup
                t_leg = time.perf_counter()
                try:                                        # a failing leg must not lose the cfg-2 line measured above
                    o = main_htdemucs(b, comm) if w == "cfg5" else main_dualpath(b, comm)
                except Exception as e:                      # noqa: BLE001
                    legs.append({"workload": w, "error": f"{type(e).__name__}: {e}"[:400], "leg_wall_s": round(time.perf_counter() - t_leg, 1)})
                    continue
                leg = {"workload": o["config"]["workload"], "metric": o["metric"], "steps": o["steps"], "warmup": o["warmup"],
                       "ms_per_step": o["ms_per_step"], "value": o["value"], "unit": o["unit"], "launch": o["config"]["launch"],
                       "roofline": o["roofline"], "leg_wall_s": round(time.perf_counter() - t_leg, 1)}
                if "cpu_baseline" in o:
                    leg["cpu_baseline"] = o["cpu_baseline"]
                legs.append(leg)
            out["other_workloads"] = legs
        print(json.dumps(out), flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
