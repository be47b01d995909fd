"""Start one process per GPU from a single command line.

The reference's entry points train on every GPU of the node by themselves: `pl.Trainer(strategy="ddp", devices="auto")` re-launches the
script once per device (asteroid_librimix_trainer.py:125-135), the tasnet env starts `python3 musdbhq_train.py --rank r` per GPU with
`subprocess.Popen` and ends every rank when one dies (tasnet_musdbhq_trainer.py:17-57), speechbrain goes through
`ddp_init_group` (speechbrain_librimix_trainer.py:592).  `spawn_ranks` is that launcher for `bench.py --gpus N` and
`python -m fqss_amd.train`: N fresh children of the same command line, each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in its environment (what `torch.distributed.run` would set; `Comm.from_env` reads them), rendezvous on 127.0.0.1.

Rules it keeps (they come from the pool this runs on, and are good practice anywhere):
  * the parent must not have touched the GPU: children are STARTED, the parent is never replaced (no exec after HIP init), and a
    parent that holds a HIP context would be one more process on the card;
  * when a rank exits non-zero the others are ended (SIGTERM, then SIGKILL after a grace period) -- by PID, never by pattern -- and
    the parent returns that rank's exit code: a step whose gradients were not averaged must not continue (parallel.CommError);
  * rank 0 inherits stdout (bench.py prints its ONE JSON line there), every rank inherits stderr.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def already_launched():
    """true inside a rank started by torch.distributed.run / spawn_ranks (or any launcher that exports WORLD_SIZE)"""
    return "WORLD_SIZE" in os.environ


def visible_gpus():
    """number of GPUs this process may use, WITHOUT creating a HIP context (device_count() does not initialise the runtime on this
    image; is_available() does)"""
    import torch
    try:
        return int(torch.cuda.device_count())
    except Exception:       # noqa: BLE001 -- a CPU-only build
        return 0


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv=None, env=None, grace_s=10.0, poll_s=0.2):
    """run `argv` (default: this very command line) as ranks 0..n-1 and wait; returns the exit code to leave with (0 = every rank
    succeeded, otherwise the first failing rank's code, 128 + signal for a rank ended by a signal)"""
    if n < 1:
        raise ValueError(f"spawn_ranks: {n} ranks")
    argv = list(argv) if argv is not None else [sys.executable] + sys.argv
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(_free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this image
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL))
        code = 0
        live = set(range(n))
        while live and code == 0:
            time.sleep(poll_s)
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = 128 - rc if rc < 0 else rc
                    print(f"fqss_amd.launch: rank {r} of {n} exited with {rc}; ending the other ranks", file=sys.stderr, flush=True)
        return code
    finally:
        _end(procs, grace_s)


def _end(procs, grace_s):
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            p.send_signal(signal.SIGTERM)
        except OSError:
            pass
    t_end = time.time() + grace_s
    for p in alive:
        try:
            p.wait(max(0.0, t_end - time.time()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()
