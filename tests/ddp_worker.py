"""One data-parallel rank of tests/test_gpu_ddp.py (a fresh process: started before anything touches the GPU).  Both ranks share
GPU 0 and exchange over gloo (FQSS_DIST_BACKEND): what is checked is the step logic -- segmented backward, bucketed exchange on the
communication stream, 1/world folded into clip + Adam, observer-range synchronisation -- not the transport.

    python -m tests.ddp_worker <rank> <world> <port> <out.pt> <scenario>
"""
import os
import sys

import numpy as np


def main():
    rank, world, port, out, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                      FQSS_DIST_BACKEND="gloo")
    import torch
    from fqss_amd import ops
    from fqss_amd.parallel import Comm
    from fqss_amd.runtime import KDTrainStep
    from tests.test_gpu_model import T, _leave_observer, _tiny_pair
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_step.npz"))
    comm = Comm.from_env("cuda")
    assert comm.world == world and comm.rank == rank
    x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
    xr, tr = shard(x, tgt, rank, world)
    res = {}
    with ops.poison_carriers(True):
        if scenario.startswith("step"):
            # quantizing phase from the reference's state after 50 steps; eager step, then (graph) capture + replays
            model, fmodel = _tiny_pair(g, prefix="s50.post_sd.")
            _leave_observer(model)
            step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, buckets=2)
            assert step.segments is not None and len(step.segments) == 2
            losses = [step(xr, tr)["loss"].item()]
            if scenario == "step_graph":
                step.capture(xr, tr, warmup=0)
                assert len(step._graphs[0]) == 2
            for _ in range(2):
                losses.append(step(xr, tr)["loss"].item())
            res["losses"] = losses
            res["params"] = {k: v.detach().cpu() for k, v in model.named_parameters()}
            res["gnorm"] = step.arena.gnorm.item()
        else:
            # observer phase on per-rank data, then the one-time synchronisation of the observed activation ranges
            model, fmodel = _tiny_pair(g)
            step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, buckets=2)
            with torch.no_grad():
                for _ in range(50):
                    model(xr)
            rng = lambda: {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "activation_fake_quantize" in k}
            res["before"] = rng()
            assert not step._ranges_synced
            step._maybe_sync_ranges()
            assert step._ranges_synced
            res["after"] = rng()
    torch.cuda.synchronize()
    torch.save(res, out)
    comm.barrier()
    comm.close()


def shard(x, tgt, rank, world):
    """rank r's samples (the fixture batch is split over the ranks; rank 1 also time-shifts its share so the shards differ)"""
    lo = rank * x.shape[0] // world
    hi = (rank + 1) * x.shape[0] // world
    xs, ts = x[lo:hi].clone(), tgt[lo:hi].clone()
    if rank:
        xs, ts = xs.roll(37 * rank, -1).contiguous(), ts.roll(37 * rank, -1).contiguous()
    return xs, ts


if __name__ == "__main__":
    main()
