"""Data parallelism: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference wraps the Lightning system in DistributedDataParallel (strategy="ddp",
asteroid_librimix_trainer.py:125-135): per-rank batches, gradients averaged by a bucketed
all-reduce, nothing else synchronised (observer ranges diverge per rank).
Here the whole gradient is ONE flat fp32 buffer (20.5 MB for ConvTasNet), so the exchange is a
single all-reduce(SUM) and the 1/world factor is folded into the clip+Adam kernel.
"""
import os

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, rank=0, world=1, local_rank=0, backend=None):
        self.rank, self.world, self.local_rank, self.backend = rank, world, local_rank, backend

    @classmethod
    def from_env(cls, device_type="cuda"):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # RCCL ("nccl" on ROCm) on GPUs; FQSS_DIST_BACKEND=gloo lets several ranks share one GPU in tests
            backend = os.environ.get("FQSS_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
            if device_type == "cuda":
                torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            return cls(rank, world, local, backend)
        return cls(rank, world, local, dist.get_backend() if dist.is_initialized() else None)

    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_max(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def broadcast(self, t, src=0):
        if self.world > 1:
            dist.broadcast(t, src)
        return t

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def shard(self, n_items):
        """contiguous shard [lo, hi) of n_items for this rank (batch-sharded data parallel)"""
        per = n_items // self.world
        return self.rank * per, (self.rank + 1) * per

    def sync_observer_ranges(self, model):
        """optional (NOT reference behaviour): average the activation ranges over ranks once the
        50-call observer phase ends, so every rank quantizes on the same grid"""
        if self.world <= 1:
            return
        from .quantization.qat.qat_quant import GradientActivationFakeQuantize
        rs = [m for m in model.modules() if isinstance(m, GradientActivationFakeQuantize)]
        if not rs:
            return
        flat = torch.cat([torch.cat([m.min_range.data, m.max_range.data]) for m in rs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= self.world
        for i, m in enumerate(rs):
            m.min_range.data.copy_(flat[2 * i:2 * i + 1])
            m.max_range.data.copy_(flat[2 * i + 1:2 * i + 2])

    def close(self):
        if dist.is_initialized():
            dist.destroy_process_group()
