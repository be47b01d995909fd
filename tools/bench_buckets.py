"""Single-GPU anatomy of the data-parallel step's schedule (no exchange): for cfg 2 / 3 / 4 / 5 AT THEIR bench.py BATCH the gradient buckets
(bytes per backward segment), the duration of every segment's hipGraph, and -- what decides how much of the exchange is exposed -- the
compute time that still runs AFTER a bucket is ready (the remaining segments), against a MODEL of a ring all-reduce of that bucket over
8 ranks on xGMI: LATENCY + bandwidth, t = 2 (N-1) alpha + 2 (N-1)/N x bytes / 153 GB/s with alpha = 5 us per ring step (2 (N-1) = 14 steps:
70 us per all-reduce; the floor VERDICT r03 asked for was 30 us) -- one link's 153 GB/s per direction (MI355X_MICROARCH.md / the task
statement).  The model has never been checked against RCCL on a node (no multi-GPU box on this pool).  Also the cost of cutting the
backward into segments at all (1 graph vs n graphs).
    python tools/bench_buckets.py [cfg2] [cfg3] [cfg4] [cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.stress_step import build
from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
from fqss_amd.runtime import KDTrainStep

XGMI_RING_GBS = 153.0                # a ring all-reduce moves every byte over ONE link per hop: 153 GB/s per direction
RING_ALPHA_US, N_RANKS = 5.0, 8      # per ring step (launch + hop latency): 2 (N - 1) steps per all-reduce


def make(which, nb):
    step, x, tgt = build(which, torch.device("cuda", 0), hd_batch=4, hd_seconds=10.0)      # bench.py's shapes
    kw = dict(loss="l1_sdr", clip=0.0) if which == "cfg5" else {}
    s = KDTrainStep(step.model, step.fmodel, lr=1e-4, buckets=nb, **kw)
    s(x, tgt)
    for m in s.model.modules():
        if isinstance(m, GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
    s(x, tgt); s(x, tgt)
    s.capture(x, tgt)
    return s, x, tgt


def main(workloads):
    for which in workloads:
        res = {}
        for nb in (1, 4):
            s, x, tgt = make(which, nb)
            for _ in range(3):
                s(x, tgt)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 10 if which != "cfg5" else 4
            for _ in range(n):
                s(x, tgt)
            torch.cuda.synchronize()
            res[nb] = (time.perf_counter() - t0) / n * 1e3
            if nb == 4:
                graphs, g2 = s._graphs
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(graphs) + 2)]
                s._stage(x, tgt, None)
                ev[0].record()
                for k, gk in enumerate(graphs):
                    gk.replay()
                    ev[k + 1].record()
                g2.replay()
                ev[-1].record()
                torch.cuda.synchronize()
                seg_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(len(graphs))]
                opt_ms = ev[-2].elapsed_time(ev[-1])
                segs = s.segments if s.segments is not None else [(0, s.arena.numel)]
                nseg = len(graphs)
                print(f"{which}: {s.arena.numel * 4 / 1e6:.1f} MB of gradients in {nseg} buckets; step {res[1]:.2f} ms as one backward graph, "
                      f"{res[4]:.2f} ms as {nseg} segment graphs; clip + Adam {opt_ms:.2f} ms", flush=True)
                for k in range(nseg):      # backward segment k = forward segment nseg-1-k
                    lo, hi = segs[nseg - 1 - k]
                    mb = (hi - lo) * 4 / 1e6
                    after = sum(seg_ms[k + 1:])
                    lat = 2 * (N_RANKS - 1) * RING_ALPHA_US * 1e-3                              # ms
                    ring8 = lat + 2 * (N_RANKS - 1) / N_RANKS * mb / 1e3 / XGMI_RING_GBS * 1e3  # ms
                    print(f"   bucket {k} (ready after {'fwd + loss + ' if k == 0 else ''}segment {k}: {seg_ms[k]:.2f} ms): {mb:7.2f} MB, "
                          f"ring all-reduce at 8 ranks ~{ring8:.3f} ms ({lat:.3f} latency + bandwidth), backward still to run behind it "
                          f"{after:.2f} ms -> exposed ~{max(0.0, ring8 - after):.3f} ms", flush=True)
            del s
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main(sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5"])
