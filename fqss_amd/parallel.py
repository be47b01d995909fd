"""Data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference wraps the Lightning system in DistributedDataParallel (strategy="ddp",
asteroid_librimix_trainer.py:125-135): per-rank batches, gradients averaged by a bucketed
all-reduce, nothing else synchronised (observer ranges diverge per rank).
Here the whole gradient is ONE flat fp32 buffer (20.5 MB for ConvTasNet), so the exchange is a
single all-reduce(SUM) and the 1/world factor is folded into the clip+Adam kernel.
"""
import datetime
import os
import sys

import torch
import torch.distributed as dist


def local_device(local_rank):
    """index of the GPU rank `local_rank` of this node computes on: its own.  A launch with more ranks than visible devices is refused --
    two RCCL ranks on one card cannot form a communicator, and silently doubling up would report a node's throughput from half its GPUs
    -- except under FQSS_DIST_BACKEND=gloo, where the tests deliberately run two ranks on GPU 0 (tests/test_gpu_ddp.py)."""
    n = torch.cuda.device_count()
    if local_rank < n:
        return local_rank
    if n > 0 and os.environ.get("FQSS_DIST_BACKEND") == "gloo":
        return local_rank % n
    raise RuntimeError(f"fqss_amd.parallel: LOCAL_RANK {local_rank} but {n} visible GPU(s): one rank per GPU "
                       "(check --gpus / HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)")


class CommError(RuntimeError):
    """a collective of the gradient / range exchange failed or timed out on this rank (docs/history/DESIGN_rounds_1-5.md 6, "When the exchange fails")"""


class Comm:
    """What happens when the exchange fails (a peer died, a link error, a hang).  gloo (CPU ranks, the GPU tests' transport): the
    collective is synchronous, it raises on the surviving ranks as soon as the peer's socket closes (after FQSS_DIST_TIMEOUT_S,
    default 600 s, at the latest) and is re-raised as CommError naming the rank and the operation.  RCCL ("nccl"): `dist.all_reduce` only
    ENQUEUES the collective on the communication stream, so `_run` sees enqueue-time errors only (-> CommError); a failure while the
    collective is in flight is found by the process group's watchdog, and TORCH_NCCL_ASYNC_ERROR_HANDLING=1 (TearDown mode, set below
    unless the caller chose otherwise) makes it abort the communicator and END THE PROCESS with SIGABRT -- no Python exception, no
    CommError; launch.spawn_ranks / torch.distributed.run see the non-zero exit and end the other ranks.  Either way nothing is retried
    and nothing is restarted in place: a step whose gradients were not averaged must not reach the optimizer, and a process that has initialised the GPU must never
    be replaced by another (no exec on this pool); the rank exits non-zero, torch.distributed.run ends the job, and the job resumes
    from the last checkpoint (process.py) as a NEW launch."""

    def __init__(self, rank=0, world=1, local_rank=0, backend=None, force=False):
        self.rank, self.world, self.local_rank, self.backend = rank, world, local_rank, backend
        # force: issue the collectives even at world 1 (FQSS_FORCE_DIST=1) -- a one-rank RCCL communicator really executes its
        # all-reduce kernels on the communication stream, which is how the exchange schedule (bucket graphs, stream ordering, capture
        # next to the process group's watchdog) is exercised on a one-GPU box (tests/test_gpu_ddp.py)
        self.force = bool(force)
        self.active = world > 1 or self.force

    @classmethod
    def from_env(cls, device_type="cuda"):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        force = os.environ.get("FQSS_FORCE_DIST", "0") == "1"
        if (world > 1 or force) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # RCCL ("nccl" on ROCm) on GPUs; FQSS_DIST_BACKEND=gloo lets several ranks share one GPU in tests
            backend = os.environ.get("FQSS_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
            if device_type == "cuda":
                torch.cuda.set_device(local_device(local))
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            timeout = datetime.timedelta(seconds=float(os.environ.get("FQSS_DIST_TIMEOUT_S", "600")))
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
            return cls(rank, world, local, backend, force)
        return cls(rank, world, local, dist.get_backend() if dist.is_initialized() else None, force and dist.is_initialized())

    def _run(self, what, fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except RuntimeError as e:           # DistBackendError / DistNetworkError are RuntimeErrors
            msg = f"fqss_amd.parallel: {what} failed on rank {self.rank} of {self.world} ({self.backend}): {e}"
            print(msg, file=sys.stderr, flush=True)
            raise CommError(msg) from e

    def all_reduce_sum(self, t):
        if self.active:
            self._run("all_reduce(SUM)", dist.all_reduce, t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_max(self, t):
        if self.active:
            self._run("all_reduce(MAX)", dist.all_reduce, t, op=dist.ReduceOp.MAX)
        return t

    def broadcast(self, t, src=0):
        if self.active:
            self._run("broadcast", dist.broadcast, t, src)
        return t

    def barrier(self):
        if self.active:
            self._run("barrier", dist.barrier)

    def shard(self, n_items):
        """contiguous shard [lo, hi) of n_items for this rank (batch-sharded data parallel)"""
        per = n_items // self.world
        return self.rank * per, (self.rank + 1) * per

    def sync_observer_ranges(self, model):
        """optional (NOT reference behaviour): average the activation ranges over ranks once the
        50-call observer phase ends, so every rank quantizes on the same grid"""
        if not self.active:
            return
        from .quantization.qat.qat_quant import GradientActivationFakeQuantize
        rs = [m for m in model.modules() if isinstance(m, GradientActivationFakeQuantize)]
        if not rs:
            return
        flat = torch.cat([torch.cat([m.min_range.data, m.max_range.data]) for m in rs])
        self._run("all_reduce(SUM) of the observer ranges", dist.all_reduce, flat, op=dist.ReduceOp.SUM)
        flat /= self.world
        for i, m in enumerate(rs):
            m.min_range.data.copy_(flat[2 * i:2 * i + 1])
            m.max_range.data.copy_(flat[2 * i + 1:2 * i + 2])

    def close(self):
        if dist.is_initialized():
            dist.destroy_process_group()
