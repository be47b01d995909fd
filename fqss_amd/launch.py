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
  * rank 0 inherits stdout (bench.py prints its ONE JSON line there), every rank inherits stderr;
  * the ranks never outlive the launcher: SIGTERM / SIGHUP / SIGINT to the parent (a `timeout`, a job scheduler, the test harness) run
    the same clean-up as a failing rank (Python would otherwise exit on SIGTERM without its `finally`), every child is the leader of
    its own process group (ended as a group: a rank's data-loader helpers go with it) and carries PR_SET_PDEATHSIG, so even a parent
    that is SIGKILLed takes them along;
  * one rank per visible GPU: more ranks than devices is an error here, not a wrap-around;
  * the rendezvous port is held by the parent (bound, not listening, SO_REUSEADDR -- which rank 0's TCPStore also sets, so it can
    listen on it) until the ranks have ended: nobody else on a busy node is handed the same "free" port in between.
"""
import ctypes
import os
import signal
import socket
import subprocess
import sys
import threading
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


def _reserve_port():
    """(socket, port): an ephemeral port on 127.0.0.1, kept bound (never listening) by the caller for as long as the ranks live"""
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    return s, s.getsockname()[1]


def _child_setup():
    """in the child, before exec: die with the launcher (PR_SET_PDEATHSIG = 1) -- a SIGKILLed parent runs no handler"""
    try:
        ctypes.CDLL(None).prctl(1, signal.SIGTERM)
    except Exception:       # noqa: BLE001 -- not Linux: the signal handlers below still cover SIGTERM / SIGHUP / SIGINT
        pass


class _Terminated(SystemExit):
    pass


def spawn_ranks(n, argv=None, env=None, grace_s=10.0, poll_s=0.2, gpus=True):
    """run `argv` (default: this very command line) as ranks 0..n-1 and wait; returns the exit code to leave with (0 = every rank
    succeeded, otherwise the first failing rank's code, 128 + signal for a rank ended by a signal or for a signal sent to the launcher).
    gpus: the ranks compute on GPUs, one each -- refused when fewer are visible (FQSS_DIST_BACKEND=gloo, the tests' several-ranks-on-
    one-card mode, excepted)."""
    if n < 1:
        raise ValueError(f"spawn_ranks: {n} ranks")
    base = dict(os.environ if env is None else env)
    if gpus and base.get("FQSS_DIST_BACKEND") != "gloo":
        have = visible_gpus()
        if n > have:
            raise RuntimeError(f"fqss_amd.launch: {n} ranks asked for, {have} GPU(s) visible: one rank per GPU")
    argv = list(argv) if argv is not None else [sys.executable] + sys.argv
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    held = None
    if "MASTER_PORT" not in base:
        held, port = _reserve_port()
        base["MASTER_PORT"] = str(port)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this image
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []

    def on_signal(signum, _frame):
        raise _Terminated(128 + signum)

    main_thread = threading.current_thread() is threading.main_thread()
    old = {}
    if main_thread:                      # signal.signal only works there; elsewhere PDEATHSIG + the caller's own handling remain
        for sg in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            old[sg] = signal.signal(sg, on_signal)
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True,
                                          preexec_fn=_child_setup))
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
    except _Terminated as t:
        print(f"fqss_amd.launch: signal {t.code - 128}: ending the {n} ranks", file=sys.stderr, flush=True)
        return t.code
    finally:
        for sg, h in old.items():        # (a second signal during the clean-up must not abandon it half-way)
            signal.signal(sg, signal.SIG_IGN)
        _end(procs, grace_s)
        for sg, h in old.items():
            signal.signal(sg, h)
        if held is not None:
            held.close()


def _signal_group(p, sig):
    """the rank and whatever it started (its own session / process group, by pgid = its pid -- never by pattern)"""
    try:
        os.killpg(p.pid, sig)
    except OSError:
        try:
            p.send_signal(sig)
        except OSError:
            pass


def _end(procs, grace_s):
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        _signal_group(p, signal.SIGTERM)
    t_end = time.time() + grace_s
    for p in alive:
        try:
            p.wait(max(0.0, t_end - time.time()))
        except subprocess.TimeoutExpired:
            _signal_group(p, signal.SIGKILL)
            p.wait()
