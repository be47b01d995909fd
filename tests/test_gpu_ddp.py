"""SURVEY.md 8(e) on real kernels: two data-parallel ranks (fresh child processes sharing GPU 0, gloo transport) run
`KDTrainStep(comm=...)` on their own shards -- segmented backward, one all-reduce per gradient bucket on the communication stream
overlapping the next segment's backward, 1/world folded into clip + Adam, eagerly and as hipGraph replays -- and must end with the
parameters of a 1-rank run that is fed the AVERAGE of the per-rank gradients (the reference's DDP semantics: per-rank input
normalisation and per-rank loss, gradients averaged; asteroid_librimix_trainer.py:125-135)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(tmp_path, scenario, world=2):
    port = str(29000 + (os.getpid() * 7 + hash(scenario)) % 2000)
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(world)]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    ps = [subprocess.Popen([sys.executable, "-m", "tests.ddp_worker", str(r), str(world), port, outs[r], scenario], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            raise
        logs.append(out)
    assert all(p.returncode == 0 for p in ps), "\n----\n".join(logs)
    return [torch.load(o, weights_only=False) for o in outs]


@pytest.mark.parametrize("scenario", ["step_eager", "step_graph", "step_graph_ahead", "step_graph:dptnet", "step_graph:sepformer", "step_graph:htdemucs",
                                      "step_graph_ahead:sepformer"])
def test_two_rank_step_equals_one_rank_step_on_the_averaged_gradients(tmp_path, scenario):
    _two_rank_check(tmp_path, scenario, later_atol=None)


@pytest.mark.parametrize("scenario", ["step_graph", "step_graph:dptnet", "step_graph:sepformer", "step_graph:htdemucs"])
def test_two_rank_later_steps_in_deterministic_mode(tmp_path, scenario, monkeypatch):
    """under FQSS_DETERMINISTIC=1 (integer-shadow gradient sums, include/fqss.h fqss_set_deterministic) the
    run-to-run noise of the fp32 atomics is gone from both sides of the comparison, so the later steps of the two-rank run are held to
    0.3 dB of the one-rank run on the averaged gradients for EVERY family (1.0 dB without the mode for the deeper ones)"""
    monkeypatch.setenv("FQSS_DETERMINISTIC", "1")
    _two_rank_check(tmp_path, scenario, later_atol=0.3)


def _two_rank_check(tmp_path, scenario, later_atol):
    """ConvTasNet (2 buckets, eager and replay) and the DDP configurations themselves -- cfg 4 `speechbrain_librimix_trainer.py:592`,
    cfg 5 `htdemucs_musdbhq/distrib.py:51-59` (find_unused_parameters=True), cfg 3 for the dual-path LSTM family: 3-4 gradient
    buckets through capture + replay, the grad-less MHA range parameters of SURVEY A.2 Q1 (zeros in the flat buffer, skipped by
    Adam), the per-sample objective of the speechbrain env, the l1_sdr loss without clipping."""
    from fqss_amd import ops
    from fqss_amd.runtime import KDTrainStep
    from tests.ddp_worker import family, shard
    ranks = _run_ranks(tmp_path, scenario)
    fam = scenario.split(":")[1] if ":" in scenario else "convtasnet"
    model, fmodel, x, tgt, kw, nb = family(fam)
    shards = [shard(x, tgt, r, 2) for r in range(2)]
    # 1-rank reference: per step, the gradients of every shard (each with its own input normalisation and loss), averaged, then ONE
    # clip + Adam -- 1/world enters as the gradient scale exactly as it does on the ranks
    ref = KDTrainStep(model, fmodel, **kw)
    losses = [[], []]
    g1 = None
    with ops.poison_carriers(True):
        for it in range(3):
            gsum = torch.zeros_like(ref.arena.flat_g)
            for r, (xs, ts) in enumerate(shards):
                losses[r].append(ref._fwd_bwd(xs, ts)["loss"].item())
                gsum += ref.arena.flat_g
            if it == 0:      # (the arenas are laid out in bucket order on the ranks: compared per parameter, through the .grad views)
                ref.arena.flat_g.copy_(gsum)
                g1 = {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters() if v.grad is not None}
            ref.arena.flat_g.copy_(gsum)
            ref.arena.clip_adam_step(ref.lr, ref.clip, grad_scale=0.5)
    # TIGHT: step 1 starts from identical state, so the gradient every rank holds behind the bucketed exchange must be the
    # sum of the per-shard gradients of the one-rank run element for element -- a mis-scaled or missing bucket cannot hide in Adam's
    # +-lr steps here.  Whole-vector error and the worst element, relative to the gradient's largest entry.
    gn = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g1.values())))
    for r in range(2):
        assert set(ranks[r]["g1"]) == set(g1)
        err = float(torch.sqrt(sum(((ranks[r]["g1"][k] - v).double() ** 2).sum() for k, v in g1.items())))
        assert err <= 1e-4 * gn, (r, err, gn)
        for k, v in g1.items():      # every tensor: its own norm, plus a floor at the whole gradient's noise level for the near-zero ones
            dk = float((ranks[r]["g1"][k] - v).norm())
            assert dk <= 1e-3 * float(v.norm()) + 2e-6 * gn, (r, k, dk, float(v.norm()))
    assert all(torch.equal(ranks[0]["g1"][k], ranks[1]["g1"][k]) for k in g1)
    want = {k: v.detach().cpu() for k, v in model.named_parameters()}
    print(scenario, "gradient buckets (bytes):", ranks[0]["bucket_bytes"], "losses", ranks[0]["losses"], losses[0])
    assert 2 <= len(ranks[0]["bucket_bytes"]) <= nb
    for r in range(2):
        # step 1 starts from identical state: same loss to fp32 noise; later steps feel the (chaotic) quantized forward of updated weights
        np.testing.assert_allclose(ranks[r]["losses"][0], losses[r][0], rtol=2e-6)
        # (tiny DPTNet / Sepformer / HTDemucs: a handful of flipped bins behind an update move a later loss by several tenths of a dB --
        # the B1 gates; with the split-K atomics a run is a sample: 0.61 dB was seen once at step 3 of tiny Sepformer.  The tight check
        # of this test is the step-1 gradient above; steps 2-3 only have to stay in the neighbourhood)
        if later_atol is None:
            np.testing.assert_allclose(ranks[r]["losses"], losses[r], atol=0.05 if fam == "convtasnet" else 1.0, rtol=1e-2 if fam != "htdemucs" else 5e-2)
        else:
            np.testing.assert_allclose(ranks[r]["losses"], losses[r], atol=min(later_atol, 0.05 if fam == "convtasnet" else later_atol), rtol=0)
    lr = kw["lr"]
    unused, worst, n_off, n_all = 0, 0.0, 0, 1
    for k, v in want.items():
        a, b = ranks[0]["params"][k], ranks[1]["params"][k]
        assert torch.equal(a, b), k                                        # the replicas stay bit-identical to each other
        # Adam moves a parameter by ~lr per step: fp32-noise-level gradient differences (atomics order) stay far below that
        d, tol = (a - v).abs(), 0.3 * lr + 1e-3 * float(v.abs().max())
        if fam == "convtasnet":
            assert float(d.max()) <= tol, (k, float(d.max()))
        else:
            # deeper / attention networks: a gradient element at the fp32-noise level may take the other sign in the segmented run,
            # and Adam turns a sign into +-lr per step -- rare elements, bounded by the three steps taken
            worst = max(worst, float(d.max()) / lr)
            n_off += int((d > tol).sum())
            n_all += d.numel()
            assert float(d.max()) <= 6.5 * lr + 1e-3 * float(v.abs().max()), (k, float(d.max()))
    print(scenario, "parameters off by more than 0.3 lr:", n_off, "of", n_all, "worst", worst, "lr")
    # (tiny DPTNet, 121-sample batch: its own single-rank run lands on a step-2 loss of 0.554 or 0.567 dB from one launch to the next --
    #  fp32-atomics noise in the first gradient, Adam's sign-like first update, weights re-rounded to their grids -- and 0 .. 1.6 % of the
    #  parameters sit one Adam step apart; measured over repeated runs)
    assert n_off <= 0.05 * n_all, (n_off, n_all)
    if fam in ("dptnet", "sepformer"):
        # the attention core's `attn` / `softmax` quantizers only observe, their ranges never receive a gradient
        for k, p in model.named_parameters():
            if ("fake_quantize_attn" in k or "fake_quantize_softmax" in k) and k.endswith("_range"):
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                unused += 1
        assert unused >= 8
    if kw["clip"] > 0:      # (the norm of the THIRD step's gradient: behind two updates of a chaotic quantized net for the deeper families)
        # The tiny DPTNet's step 2 has two outcomes from one launch to the next (loss 0.554 | 0.567 dB, see above: atomics noise in the
        # first gradient + Adam's sign-like first update): when the two sides of this comparison took different ones -- step-2 losses
        # more than 1 % apart -- the third gradient's norm is compared between two trajectories, not two exchanges (seen once in round 6:
        # 0.573 vs 0.550, norms 13.8 vs 11.3); the 20 % bound is for runs on the same trajectory, a bifurcated pair only has to stay
        # within a factor of 1.5.  The exchange itself is gated by the step-1 gradient above and, for the later steps, by
        # test_two_rank_later_steps_in_deterministic_mode.
        same = fam == "convtasnet" or abs(ranks[0]["losses"][1] - losses[0][1]) <= 1e-2 * abs(losses[0][1])
        bound = 1e-2 if fam == "convtasnet" else (0.2 if same else 0.5)
        assert abs(ranks[0]["gnorm"] - ref.arena.gnorm.item()) <= bound * ref.arena.gnorm.item(), (same, ranks[0]["losses"], losses[0])


def test_replicas_keep_their_own_observer_ranges_like_the_reference(tmp_path):
    """`sync_observer_ranges=False` = the reference's behaviour (qat_quant.py:230-232 writes .data during the observer phase, DDP
    only averages gradients): the activation ranges of the replicas differ when the phase ends and nothing re-synchronises them;
    the quantizing steps that follow run on per-rank grids, the range PARAMETERS move by the same averaged gradients (their
    difference persists), and every other parameter stays bit-identical over the ranks."""
    ranks = _run_ranks(tmp_path, "observer_nosync")
    differing = 0
    for k, v0 in ranks[0]["before"].items():
        v1 = ranks[1]["before"][k]
        differing += int(not torch.equal(v0, v1))
        for r in range(2):
            assert torch.equal(ranks[r]["after"][k], ranks[r]["before"][k]), k       # no synchronisation happened
    assert differing > 20
    still = sum(int(not torch.equal(ranks[0]["after_steps"][k], ranks[1]["after_steps"][k])) for k in ranks[0]["after_steps"])
    assert still > 20
    for k, v in ranks[0]["weights"].items():
        assert torch.equal(v, ranks[1]["weights"][k]), k
    assert all(np.isfinite(ranks[r]["losses"]).all() for r in range(2))


def test_observer_ranges_are_synchronised_once_over_the_ranks(tmp_path):
    """documented deviation: the reference's replicas keep the activation ranges their own 50 observer calls
    left (qat_quant.py:230-232 writes .data, DDP never re-synchronises); here every rank takes the mean when the phase ends"""
    ranks = _run_ranks(tmp_path, "observer")
    diff = 0
    for k, v0 in ranks[0]["before"].items():
        v1 = ranks[1]["before"][k]
        diff += int(not torch.equal(v0, v1))
        mean = (v0 + v1) / 2
        for r in range(2):
            np.testing.assert_allclose(ranks[r]["after"][k].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
        assert torch.equal(ranks[0]["after"][k], ranks[1]["after"][k]), k
    assert diff > 20          # the per-rank observations did differ (tiny model: 52 range tensors)


def test_full_size_cfg2_two_ranks_replay_like_bench(tmp_path):
    """`bench.py --gpus 2` on the FULL-SIZE cfg-2 model (8 x 4 s per rank), two ranks on this one GPU over gloo: calibration, capture into one
    hipGraph per gradient bucket, three replays with the teacher one batch ahead over alternating batches.  The replicas must stay
    bit-identical (every rank applies the same reduced gradient) and the loss finite: what is left for the first run on a real node is
    the RCCL transport."""
    ranks = _run_ranks(tmp_path, "bench2")
    assert ranks[0]["n_graphs"] == ranks[1]["n_graphs"] >= 2
    assert all(np.isfinite(l) for r in ranks for l in r["losses"]), [r["losses"] for r in ranks]
    assert torch.isfinite(ranks[0]["flat_p"]).all()
    assert torch.equal(ranks[0]["flat_p"], ranks[1]["flat_p"])
    print("bench2: buckets (bytes)", ranks[0]["bucket_bytes"], "losses", ranks[0]["losses"], ranks[1]["losses"])
    # ... and the run is the SAME run whatever else the GPU is doing (round 5): two processes on one card slow each other's memory
    # pipes down, which is what exposed the missing wait states behind the asm 16-B stores (csrc/fqss_dev.h store16, csrc/qgemm.hip
    # st16_sc1: the grouped weight-gradient kernel then wrote its next slab ADDRESS in place of a few slab values, and ~4 runs in 10
    # came back with losses of 4.7 ... 10 instead of 3.29 here, some with NaN parameters -- profiles/r05_store_hazard.txt).  The
    # per-layer weight-gradient kernels (FQSS_GROUP_WGRAD=0: no slabs) give the reference trajectory.
    os.environ["FQSS_GROUP_WGRAD"] = "0"
    try:
        ref = _run_ranks(tmp_path, "bench2")
    finally:
        del os.environ["FQSS_GROUP_WGRAD"]
    for r in range(2):
        np.testing.assert_allclose(ranks[r]["losses"], ref[r]["losses"], rtol=2e-2)


def test_rccl_executes_the_bucketed_exchange_at_world_one(tmp_path):
    """RCCL itself (backend "nccl"), on the one GPU this pool has: a one-rank communicator with the world > 1 schedule forced on
    (FQSS_FORCE_DIST=1, FQSS_FORCE_BUCKETS=1) -- full-size cfg 2, four backward-segment graphs, the four bucket all-reduces issued
    through RCCL on the communication stream between the replays, captured next to the process group's watchdog.  all-reduce(SUM)
    over one rank is the identity, so the exchanged gradient must equal the un-exchanged gradient of the same state: a collective that
    ran before its bucket was written, or a replay that did not wait for it, shows up as NaN (the arena is poisoned) or a wrong slice.
    Reference: NCCL under DDP (train_env/htdemucs_musdbhq/distrib.py:51-59, asteroid_librimix_trainer.py:125-135)."""
    r, = _run_ranks(tmp_path, "rccl1", world=1)
    print("rccl1:", {k: v for k, v in r.items()})
    assert r["n_graphs"] == 4 and r["eager_calls"] >= 4
    assert r["replay_calls"] == list(reversed(r["bucket_elems"]))        # one collective per bucket, last segment first, whole slices
    assert r["finite"] and abs(r["loss_replay"] - r["loss_ref"]) <= 1e-6 * abs(r["loss_ref"])
    assert r["g_err"] <= 1e-4 and r["g_worst"] <= 2e-4, (r["g_err"], r["g_worst"])
    assert all(np.isfinite(r["losses"])) and r["moved"] > 0


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (the driver's own command form): bench.py starts the two ranks itself
    (fqss_amd/launch.py; reference: pl.Trainer(strategy="ddp", devices="auto"), asteroid_librimix_trainer.py:125-135, and the tasnet env's
    Popen per GPU, tasnet_musdbhq_trainer.py:17-30) before touching the GPU, rank 0 prints the ONE JSON line.  Both ranks share GPU 0
    here, so the transport is gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(FQSS_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["config"]["global_batch"] == 16 and o["scaling"] == "weak"
    assert "4 backward segments" in o["config"]["launch"] and o["value"] > 0 and "roofline" in o
    # what the exchange ran on and what it cost: backend, world, one entry per rank, the four bucket sizes in
    # exchange order, the exposed exchange time from rank events
    d = o["dist"]
    assert d["backend"] == "gloo" and d["world"] == 2 and d["rccl_version"] is None and len(d["devices"]) == 2
    assert [x["rank"] for x in d["devices"]] == [0, 1] and "warning" in d          # both ranks on GPU 0 here: the line says so
    assert len(d["buckets_MB"]) == 4 and abs(sum(d["buckets_MB"]) - 20.6) < 0.5 and d["exposed_comm_ms"] >= 0.0


def test_bench_over_rccl_on_every_visible_gpu(tmp_path):
    """`python bench.py --gpus N` for N = the GPUs of this box over RCCL (backend "nccl"), N > 1 only: one rank per device, the JSON
    line's `dist` object names N distinct devices and the RCCL version.  On this pool's one-GPU boxes the test skips; the driver's
    8-GPU node runs it (its first sight of xGMI traffic).  Reference: pl.Trainer(strategy="ddp"), asteroid_librimix_trainer.py:125-135."""
    import json
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one visible GPU: RCCL between devices cannot run here (world-1 RCCL: test_rccl_executes_the_bucketed_exchange_at_world_one)")
    n = min(n, 6)                                # (the pool's process guard: at most 6 processes on the cards)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "FQSS_DIST_BACKEND")}
    env.update(PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    o = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    d = o["dist"]
    assert o["n_gpus"] == n and d["backend"] == "nccl" and d["rccl_version"] and d["world"] == n
    assert len({x["device"] for x in d["devices"]}) == n and "warning" not in d
    assert torch.isfinite(torch.tensor(o["loss_db"])) and o["value"] > 0
