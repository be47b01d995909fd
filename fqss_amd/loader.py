"""Batch order and background prefetch for the file-backed datasets.

The reference feeds its trainers from `torch.utils.data.DataLoader(LibriMix(...), shuffle=True, batch_size, num_workers, drop_last=True)`
(asteroid_librimix_trainer.py:53-67; Lightning swaps in a DistributedSampler under DDP).  At a 13 ms step a loader that reads WAV segments on
the training thread IS the step time, and worker processes that hand CPU tensors over a pipe only move the problem: here ONE reader thread
runs a batch ahead of the step --

    reader thread, batch n+1:  WAV segments -> a pinned staging buffer -> one H2D copy, the resampler and the SNR mixer on a SIDE stream
                               (fqss_resample_fir / fqss_snr_mix, one launch each for the whole batch) -> a static device slot, an event
    training thread, batch n:  `for x, tgt in loader` makes its stream wait for the slot's event (no host sync) and gets views of the slot

-- so the mixture of batch n+1 is on the device while step n replays, which is also what the teacher look-ahead of KDTrainStep wants
(`lookahead()` yields (x, tgt, x_next)).  Slots are reused round-robin: a slot is rewritten only after the stream work that consumed it
has been enqueued AND the reader's side stream has been told to wait for it (an event recorded when a later batch is taken).

Batch order: `epoch_batches` restates the index order of torch's RandomSampler / SequentialSampler + BatchSampler (single process) and
of DistributedSampler (world > 1): a seeded permutation per epoch, padded to a multiple of the world, rank r takes every world-th index."""
import queue
import threading

import torch

# held by the reader thread around a batch's device work and by KDTrainStep.capture() around a hipGraph capture
DEVICE_WORK_LOCK = threading.RLock()


def epoch_batches(n_items, batch_size, shuffle, drop_last=True, rank=0, world=1, seed=0, epoch=0):
    """list of index lists for one epoch.  world == 1: torch.utils.data.RandomSampler draws its permutation seed from the global torch
    RNG when iteration starts (sampler.py: `int(torch.empty((), dtype=torch.int64).random_().item())`) -- done the same way here, so
    `torch.manual_seed(s)` fixes the order as it does in the reference; world > 1: DistributedSampler (`seed + epoch`)."""
    if world > 1:
        if shuffle:
            g = torch.Generator().manual_seed(int(seed) + int(epoch))
            idx = torch.randperm(n_items, generator=g).tolist()
        else:
            idx = list(range(n_items))
        total = -(-n_items // world) * world
        if len(idx) < total:                       # DistributedSampler(drop_last=False) pads by wrapping around
            idx = (idx * (total // max(1, len(idx)) + 1))[:total]
        idx = idx[rank:total:world]
    elif shuffle:
        g = torch.Generator().manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        idx = torch.randperm(n_items, generator=g).tolist()
    else:
        idx = list(range(n_items))
    out = [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]
    if drop_last and out and len(out[-1]) < batch_size:
        out.pop()
    return out


class Prefetcher:
    """iterate `dataset.batch(indices, stage=...)` over `batches` with a reader thread `depth` batches ahead.

    dataset.batch(indices, stage) -> (mixture [B, 1, T], sources [B, S, T]) on `device`, enqueued on the CURRENT stream of the calling
    thread; `stage` is a pinned float32 buffer the dataset may use for its host->device copy (None on the CPU backend)."""

    def __init__(self, dataset, batches, device, depth=2):
        self.dataset, self.batches, self.device = dataset, list(batches), torch.device(device)
        self.depth = max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        if self.cuda and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.nslot = self.depth + 2
        self.wait_s = 0.0           # host time the training thread spent blocked on the reader (a loader that keeps up: ~0)

    def __len__(self):
        return len(self.batches)

    # ---- reader thread ----------------------------------------------------------------------------------------------------
    def _reader(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
                side = torch.cuda.Stream(self.device)
            for j, indices in enumerate(self.batches):
                self._free.acquire()                       # at most `depth` batches ready or in the making
                if self._stop.is_set():
                    return
                s = j % self.nslot
                if not self.cuda:
                    self._q.put((j, self.dataset.batch(indices, None), None))
                    continue
                with DEVICE_WORK_LOCK, torch.cuda.stream(side):
                    ev = self._taken[(j - self.depth) % self.nslot] if j >= self.depth else None
                    if ev is not None:
                        side.wait_event(ev)                # everything that read this slot's previous batch is enqueued before `ev`
                    if self._ready[s] is not None:
                        self._ready[s].synchronize()       # the pinned buffer's previous H2D copy (long done)
                    mix, src = self.dataset.batch(indices, self._stage[s])
                    if self._slots[s] is None or self._slots[s][0].shape != mix.shape or self._slots[s][1].shape != src.shape:
                        self._slots[s] = (torch.empty_like(mix), torch.empty_like(src))
                    self._slots[s][0].copy_(mix)
                    self._slots[s][1].copy_(src)
                    ready = torch.cuda.Event()
                    ready.record(side)
                    self._ready[s] = ready
                    del mix, src                           # temporaries of the side stream: freed into the side stream's pool
                self._q.put((j, self._slots[s], ready))
            self._q.put((None, None, None))
        except BaseException as e:                         # handed to the training thread: a loader must not die silently
            self._q.put((-1, e, None))

    def __iter__(self):
        import time
        self._q = queue.Queue()
        self._free = threading.Semaphore(self.depth)
        self._stop = threading.Event()
        self._taken = [None] * self.nslot
        self._ready = [None] * self.nslot
        self._slots = [None] * self.nslot
        self._stage = [None] * self.nslot
        if self.cuda:
            n = self.dataset.stage_elems(max(len(b) for b in self.batches)) if self.batches else 0
            self._stage = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(self.nslot)]
        th = threading.Thread(target=self._reader, name="fqss-prefetch", daemon=True)
        th.start()
        try:
            while True:
                t0 = time.perf_counter()
                j, item, ready = self._q.get()
                self.wait_s += time.perf_counter() - t0
                if j is None:
                    break
                if j == -1:
                    raise item
                if self.cuda:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ready)
                    ev = torch.cuda.Event()
                    ev.record(cur)                         # all stream work of the batches taken BEFORE this one is in front of `ev`
                    self._taken[j % self.nslot] = ev
                    for t in item:                         # the slots were allocated on the reader's stream
                        t.record_stream(cur)
                self._free.release()
                yield item
        finally:
            self._stop.set()
            self._free.release()
            th.join(timeout=30)

    def lookahead(self):
        return with_lookahead(self)


def with_lookahead(batches):
    """(x, tgt, x_next) per batch: x_next is the mixture the NEXT iteration yields as x (the same tensor object; None at the end)"""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur[0], cur[1], nxt[0]
        cur = nxt
    yield cur[0], cur[1], None
