"""One data-parallel rank of tests/test_gpu_ddp.py (a fresh process: started before anything touches the GPU).  Both ranks share
GPU 0 and exchange over gloo (FQSS_DIST_BACKEND): what is checked is the step logic -- segmented backward, bucketed exchange on the
communication stream, 1/world folded into clip + Adam, observer-range synchronisation -- not the transport.

    python -m tests.ddp_worker <rank> <world> <port> <out.pt> <scenario>
"""
import os
import sys

import numpy as np


def family(name):
    """-> (model pair in the quantizing phase from the family's reference fixture, full batch x, targets, KDTrainStep keywords)"""
    import torch
    gold = os.path.join(os.path.dirname(__file__), "golden")
    if name == "convtasnet":
        from tests.test_gpu_model import T, _leave_observer, _tiny_pair
        g = np.load(os.path.join(gold, "tiny_step.npz"))
        model, fmodel = _tiny_pair(g, prefix="s50.post_sd.")
        _leave_observer(model)
        return model, fmodel, T(g["x"]).cuda(), T(g["tgt"]).cuda(), dict(kd_lambda=0.1, lr=1e-3, clip=5.0), 2
    if name in ("dptnet", "sepformer"):
        if name == "dptnet":
            from tests.test_gpu_dptnet import T, _forced
            g = np.load(os.path.join(gold, "dpt_tiny_step.npz"))
            kw = dict(kd_lambda=0.1, lr=4e-4, clip=5.0)
        else:
            from tests.test_gpu_sepformer import T, _forced
            g = np.load(os.path.join(gold, "sep_tiny_step.npz"))
            # the speechbrain env's objective: log per sample, thresholded mean (speechbrain_librimix_trainer.py:141-149); per-rank batch 1
            kw = dict(kd_lambda=0.1, lr=1.5e-4, clip=5.0, loss="sisdr_pit_per_sample", loss_threshold=-30.0)
        model, fmodel = _forced(g, 50)
        return model, fmodel, T(g["x"]).cuda(), T(g["tgt"]).cuda(), kw, 4
    from tests.test_gpu_htdemucs import T, _models
    from fqss_amd.quantization.qat import qat_quant as QQ
    g = np.load(os.path.join(gold, "hd_tiny_step.npz"))
    model, fmodel = _models(g)
    sd = model.state_dict()
    with torch.no_grad():
        model(T(g["mix"]).cuda())                     # weight observers record; then the fixture's own (teacher-forced) ranges
        for k in g.files:
            if k.startswith("sd."):
                sd[k[3:]].copy_(T(g[k]))
    for m in model.modules():
        if isinstance(m, QQ.GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
    # the solver's step (solver.py:333-366): L1 + SDR-weighted L1 distillation, Adam without clipping (htdemucs.yaml:77-84)
    return model, fmodel, T(g["mix"]).cuda(), T(g["src"]).cuda(), dict(kd_lambda=0.1, lr=3e-4, clip=0.0, loss="l1_sdr"), 3


def main():
    rank, world, port, out, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                      FQSS_DIST_BACKEND="gloo")
    if scenario == "rccl1":
        # ONE rank, RCCL ("nccl"), the world > 1 schedule forced on: process group, bucket graphs, the all-reduces really issued
        os.environ.update(FQSS_DIST_BACKEND="nccl", FQSS_FORCE_DIST="1", FQSS_FORCE_BUCKETS="1")
    import torch
    from fqss_amd import ops
    from fqss_amd.parallel import Comm
    from fqss_amd.runtime import KDTrainStep
    from tests.test_gpu_model import T, _leave_observer, _tiny_pair
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_step.npz"))
    comm = Comm.from_env("cuda")
    assert comm.world == world and comm.rank == rank
    res = {}
    with ops.poison_carriers(True):
        if scenario.startswith("step"):
            # quantizing phase from the reference's state; eager step, then (graph) capture + replays.  "step_graph:<family>"
            # runs the DDP configurations themselves (cfg 3 / 4 / 5 families): 3-4 gradient buckets, grad-less MHA range parameters,
            # the per-sample objective, the l1_sdr loss
            fam = scenario.split(":")[1] if ":" in scenario else "convtasnet"
            model, fmodel, x, tgt, kw, nb = family(fam)
            xr, tr = shard(x, tgt, rank, world)
            # "step_graph_ahead": bench.py's schedule at world > 1: the teacher's forward of the NEXT batch as its own hipGraph
            # on the teacher stream beside the bucketed step (the next batch is the same shard here, so the 1-rank reference still applies)
            ahead = scenario.startswith("step_graph_ahead")
            step = KDTrainStep(model, fmodel, comm=comm, buckets=nb, teacher_ahead=ahead, **kw)
            assert step.segments is not None and 2 <= len(step.segments) <= nb, (step.segments, nb)     # (tiny nets have fewer cut points)
            nxt = dict(x_next=xr) if ahead else {}
            losses = [step(xr, tr, **nxt)["loss"].item()]
            # the gradient every rank holds behind step 1's exchange (SUM over the ranks; 1/world is folded into clip + Adam), from
            # identical state: the tight check of the exchange itself, before any update has made the comparison chaotic
            res["g1"] = {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters() if v.grad is not None}
            if scenario.startswith("step_graph"):
                step.capture(xr, tr, warmup=0)
                assert len(step._graphs[0]) == len(step.segments)
            for _ in range(2):
                losses.append(step(xr, tr, **nxt)["loss"].item())
            res["losses"] = losses
            res["params"] = {k: v.detach().cpu() for k, v in model.named_parameters()}
            res["gnorm"] = step.arena.gnorm.item()
            res["bucket_bytes"] = [4 * (hi - lo) for lo, hi in step.segments]
        elif scenario == "bench2":
            # bench.py --gpus 2 on the FULL-SIZE cfg-2 model: the same calls in the same order -- calibration, one
            # eager quantizing step, capture into one hipGraph per gradient bucket, replays with the teacher one batch ahead over two
            # alternating batches -- so that the first real RCCL run has only the transport left to prove
            from fqss_amd.data import synth_batch
            from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
            from fqss_amd.smoke import build_pair
            dev = torch.device("cuda", 0)
            model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
            Bq, Tq = 8, 32000
            x, tgt = synth_batch(Bq, Tq, seed=100 + rank, device=dev)
            x2, tgt2 = synth_batch(Bq, Tq, seed=200 + rank, device=dev)
            X, TG = (x, x2), (tgt, tgt2)
            step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, teacher_ahead=True)
            step(x, tgt)
            with torch.no_grad():
                for _ in range(49):
                    model(x)
            assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))
            step(x, tgt)
            step.capture(x, tgt)
            res["n_graphs"] = len(step._graphs[0])
            losses, it = [], 0
            for _ in range(3):
                r = step(X[it & 1], TG[it & 1], x_next=X[(it + 1) & 1])
                losses.append(r["loss"].item())
                it += 1
            res["losses"] = losses
            res["flat_p"] = step.arena.flat_p.detach().cpu().clone()
            res["bucket_bytes"] = [4 * (hi - lo) for lo, hi in step.segments]
        elif scenario == "rccl1":
            # RCCL executes once.  Full-size cfg 2 on a one-rank "nccl" communicator with the bucketed schedule
            # forced on: four backward-segment graphs, the all-reduce of every bucket issued through RCCL on the communication stream
            # BETWEEN two replays (all-reduce(SUM) over one rank = identity), capture next to the process group's watchdog thread
            # (capture_error_mode="thread_local").  The exchanged gradient must equal the un-exchanged one of the same state.
            import torch.distributed as dist
            from fqss_amd.data import synth_batch
            from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
            from fqss_amd.smoke import build_pair
            assert comm.backend == "nccl" and comm.active and comm.world == 1 and dist.get_backend() == "nccl"
            calls = []
            inner = comm.all_reduce_sum
            comm.all_reduce_sum = lambda t: (calls.append((t.numel(), torch.cuda.current_stream().cuda_stream)), inner(t))[1]
            dev = torch.device("cuda", 0)
            model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
            x, tgt = synth_batch(8, 32000, seed=100, device=dev)
            step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, teacher_ahead=True)
            assert step.segments is not None and len(step.segments) == 4
            step(x, tgt)
            with torch.no_grad():
                for _ in range(49):
                    model(x)
            assert all(m.n_iter >= 50 for m in model.modules() if isinstance(m, GradientActivationFakeQuantize))
            step(x, tgt)                                  # eager quantizing step: 4 buckets through RCCL already
            res["eager_calls"] = len(calls)
            main = torch.cuda.current_stream().cuda_stream
            assert all(s != main for _, s in calls)     # every collective was enqueued on the communication stream
            ref = step._fwd_bwd(x, tgt)                   # same state, NO exchange: the reference gradient
            g_ref, loss_ref = step.arena.flat_g.detach().clone(), ref["loss"].item()
            step.capture(x, tgt, warmup=0)
            res["n_graphs"] = len(step._graphs[0])
            step.arena.flat_g.fill_(float("nan"))
            del calls[:]
            r = step.replay_fwd_bwd(x, tgt)
            torch.cuda.synchronize()
            res["replay_calls"] = [n for n, _ in calls]
            res["bucket_elems"] = [hi - lo for lo, hi in step.segments]
            res["loss_ref"], res["loss_replay"] = loss_ref, r["loss"].item()
            g = step.arena.flat_g.detach()
            res["finite"] = bool(torch.isfinite(g).all())
            res["g_err"] = float((g - g_ref).norm() / g_ref.norm())
            worst = 0.0
            for p, o in zip(step.arena.params, step.arena.offsets):
                a, b = g[o:o + p.numel()], g_ref[o:o + p.numel()]
                nb_ = float(b.norm())
                if nb_ > 1e-9:
                    worst = max(worst, float((a - b).norm()) / nb_)
            res["g_worst"] = worst
            p0 = step.arena.flat_p.detach().clone()
            x2, tgt2 = synth_batch(8, 32000, seed=200, device=dev)
            X, TG = (x, x2), (tgt, tgt2)
            res["losses"] = [step(X[i & 1], TG[i & 1], x_next=X[(i + 1) & 1])["loss"].item() for i in range(3)]
            res["moved"] = float((step.arena.flat_p - p0).abs().max())
        else:
            # observer phase on per-rank data, then the one-time synchronisation of the observed activation ranges -- or, with
            # "observer_nosync", the reference's behaviour (qat_quant.py:230-232 writes .data, DDP never re-synchronises): every rank
            # keeps the ranges its own 50 calls left, and trains on from there
            x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
            xr, tr = shard(x, tgt, rank, world)
            model, fmodel = _tiny_pair(g)
            sync = scenario == "observer"
            step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, buckets=2, sync_observer_ranges=sync)
            step(xr, tr)                                  # call 1 with a backward (weight observers, first Adam step)
            with torch.no_grad():
                for _ in range(49):
                    model(xr)
            rng = lambda: {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "activation_fake_quantize" in k}
            res["before"] = rng()
            assert step._ranges_synced == (not sync)
            step._maybe_sync_ranges()
            assert step._ranges_synced
            res["after"] = rng()
            if not sync:
                res["losses"] = [step(xr, tr)["loss"].item() for _ in range(2)]       # quantizing steps on per-rank grids
                res["after_steps"] = rng()
                res["weights"] = {k: v.detach().cpu() for k, v in model.named_parameters() if "fake_quantize" not in k}
    torch.cuda.synchronize()
    torch.save(res, out)
    comm.barrier()
    comm.close()


def shard(x, tgt, rank, world):
    """rank r's samples (the fixture batch is split over the ranks; a one-sample fixture goes to every rank); every rank but 0 also
    time-shifts its share so the shards differ"""
    if x.shape[0] >= world:
        lo, hi = rank * x.shape[0] // world, (rank + 1) * x.shape[0] // world
    else:
        lo, hi = 0, x.shape[0]
    xs, ts = x[lo:hi].clone(), tgt[lo:hi].clone()
    if rank:
        xs, ts = xs.roll(37 * rank, -1).contiguous(), ts.roll(37 * rank, -1).contiguous()
    return xs, ts


if __name__ == "__main__":
    main()
