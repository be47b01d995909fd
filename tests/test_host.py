"""CPU tests of the host side: C-ABI surface, module/graph-rewrite mirror, layout helpers, and the
world_size-2 data-parallel plumbing (gloo).  No kernel is launched here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """the library loads and exports exactly what include/fqss.h declares (and _lib binds all of it)"""
    from fqss_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fqss.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(fqss_\w+)\s*\(", hdr, flags=re.M))
    assert declared and declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    if not os.path.exists(_lib.SO_PATH):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")])
    lib = _lib.load(strict=True)
    assert lib.fqss_version() == 100
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    from fqss_amd import _lib
    from fqss_amd.quantization.qat import qat_quant as QQ
    q = QQ.GradientActivationFakeQuantize(True)
    with pytest.raises(_lib.FqssError):
        q(torch.randn(4, 4))
    w = QQ.GradientWeightFakeQuantize(True, (4, 3, 1))
    with pytest.raises(_lib.FqssError):
        w(torch.randn(4, 3, 1))


def test_product_never_imports_oracle():
    """only smoke.py (the driver's checker entry) may reference oracle/ inside the package"""
    pkg = os.path.join(ROOT, "fqss_amd")
    offenders = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py") and f != "smoke.py":
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M):
                    offenders.append(os.path.join(dp, f))
    assert not offenders, offenders


def test_quantizer_api_surface():
    from fqss_amd.quantization.qat import qat_quant as QQ
    q = QQ.get_activation_quantizer(True, n_bits=8)
    assert isinstance(q, QQ.GradientActivationFakeQuantize)
    assert list(q.state_dict().keys()) == ["min_range", "max_range"]          # scratch buffers are non-persistent
    assert q.min_range.shape == (1,) and q.min_range.item() == -0.5 and q.max_range.item() == 0.5
    assert (q.max_observations, q.alpha, q.n_iter, q.observer_mode) == (50, 0.9, 0, True)
    modes = [q.next_mode() for _ in range(52)]
    assert modes[:50] == [1] * 50 and modes[50:] == [2, 2] and q.n_iter == 50
    q.enable_observer(False)
    w = QQ.get_weight_quantizer(True, (7, 3, 16), ch_out_idx=1)
    assert w.min_range.shape == (1, 3, 1) and w.axis == 1
    with pytest.raises(NotImplementedError):
        QQ.GradientActivationFakeQuantize(True, n_bits=4)
    # the public helper names users extend the library with (qat_quant.py:88-107, qat_utils.py:12-255)
    from fqss_amd.quantization.qat import qat_utils as QU
    for name in ("round_ste", "floor_ste", "grad_sign", "grad_scale", "clip_ste", "linear_quantize"):
        assert callable(getattr(QQ, name)), name
    for name in ("quant_encoderq", "quant_decoderq", "quant_conv1d", "quant_conv2d", "quant_convtr1d", "quant_convtr2d", "quant_conv1d_nl",
                 "quant_conv1d_gn_nl", "quant_conv2d_nl", "quant_convtr1d_nl", "quant_convtr2d_nl", "quant_groupnorm", "quant_layernorm",
                 "quant_batchnorm", "quant_embedding", "quant_nl", "quant_linear", "quant_linear_nl", "quant_mha", "quant_lstm", "quant_add",
                 "quant_sub", "quant_mul", "quant_div", "quant_const", "torch_weight_quantizer", "torch_activation_quantizer",
                 "quantize_known_modules", "quantize_modules", "replace_encoderq", "replace_decoderq", "replace_weight_quantizer",
                 "replace_activation_quantizer", "torch_dym_activation_quantizer", "replace_dym_activation_quantizer"):
        assert callable(getattr(QU, name)), name
    # the loss names of train_env/asteroid_librimix/wsdr.py:10-102
    from fqss_amd.train_env.asteroid_librimix import wsdr
    for name in ("SDR", "PairwiseWSDR", "sisdr", "sdr", "pairwise_wsisdr", "pairwise_wsdsdr"):
        assert callable(getattr(wsdr, name)), name
    with pytest.raises(AssertionError):
        wsdr.PairwiseWSDR("sdr")


def test_graph_rewrite_and_state_dict_layout(golden):
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat.models.convtasnetq import ConvBlock, ConvTasNetQ
    from fqss_amd.quantization.qat.models.load_model import create_model, enable_observer, quantize_model
    from fqss_amd.smoke import QCFG
    g = golden("tiny_step")
    torch.manual_seed(0)
    m = ConvTasNetQ(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)
    assert list(m.state_dict().keys()) == [k[4:] for k in g.files if k.startswith("fsd.")]
    m = quantize_model(m, dict(QCFG))
    assert list(m.state_dict().keys()) == list(g["sd_keys"])
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == g["sd0." + k].shape, k
    blk = m.masker.TCN[0]
    assert isinstance(blk, ConvBlock)
    assert isinstance(blk.shared_block[0], QL.Conv1dNlQ) and isinstance(blk.shared_block[1], nn.Identity)
    assert isinstance(blk.shared_block[2], QL.GroupNormQ) and isinstance(blk.res_conv, QL.Conv1dQ)
    assert isinstance(m.encoder, QL.Conv1dEncoderQ) and m.encoder.conv1d.in_channels == 2
    assert isinstance(m.decoder, QL.ConvTr1dDecoderQ) and m.decoder.weight_fake_quantize.axis == 1
    assert isinstance(m.mul, QL.MulQ) and isinstance(m.masker.adds[0], QL.AddQ)
    enable_observer(m, False)
    assert not m.encoder.activation_fake_quantize.observer_mode
    full = quantize_model(create_model({"name": "ConvTasNet", "n_src": 2, "kernel_size": 16, "stride": 8}), dict(QCFG))
    assert len(full.state_dict()) == 948
    assert sum(p.numel() for p in full.parameters()) == 5133123
    with pytest.raises(NotImplementedError):
        create_model({"name": "ConvTasNetMusic"})


def test_htdemucs_graph_rewrite_and_state_dict_layout(golden):
    """cfg 5: the HTDemucs module tree quantizes into the reference's key set, in the reference's order (hd_tiny_step.npz holds the
    reference's float and quantized state_dicts of the tiny configuration); full-size parameter count of the float model"""
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat.models.htdemucsq import HTDemucsQ
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    g = golden("hd_tiny_step")
    kw = dict(sources=["a", "b"], audio_channels=2, channels=8, nfft=2048, depth=4, bottom_channels=16, t_layers=3, t_heads=2)
    f = HTDemucsQ(**kw)
    assert list(f.state_dict().keys()) == [k[4:] for k in g.files if k.startswith("fsd.")]
    q = dict(qat=True, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, in_quant=False,
             in_act_n_bits=8, out_quant=True, out_act_n_bits=8, n_splitter=2, n_combiner=2, observer=True)
    m = quantize_model(HTDemucsQ(**kw), q)
    ref = [k[4:] for k in g.files if k.startswith("sd0.")]
    assert list(m.state_dict().keys()) == ref
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == g["sd0." + k].shape, k
    assert isinstance(m.encoder[0].conv, QL.Conv2dEncoderQ) and isinstance(m.tencoder[0].conv, QL.Conv1dEncoderQ)
    assert isinstance(m.decoder[3].conv_tr, QL.ConvTr2dDecoderQ) and m.decoder[3].conv_tr.residual_error_block.train_res_dec
    assert isinstance(m.tdecoder[3].conv_tr, QL.ConvTr1dDecoderQ) and not m.tdecoder[3].conv_tr.residual_error_block.train_res_dec
    assert isinstance(m.encoder[1].dconv.layers[0][0], QL.Conv1dGnNlQ) and isinstance(m.encoder[1].dconv.layers[0][6].mul, QL.MulQ)
    assert isinstance(m.crosstransformer.layers[1].cross_attn, QL.MultiheadAttentionQ)
    assert isinstance(m.crosstransformer.layers[0].norm_out.const, QL.ConstQ)
    full = create_model({"name": "HTDemucs"})
    assert sum(p.numel() for p in full.parameters()) == 26899536          # (measured on the reference: HTDemucsQ() defaults, 4 sources)


def test_dptnet_graph_rewrite_and_state_dict_layout(golden):
    """cfg 3: the DPTNet module tree quantizes into the reference's key set, in the reference's order (298 keys at the
    fixture's size: LSTMQ's four weight quantizers, MultiheadAttentionQ's seven activation + two weight quantizers, ...)"""
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat.models.dptnetq import DPTNetQ, TransformerEncoderLayer
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.smoke import QCFG
    g = golden("dpt_tiny_step")
    m = DPTNetQ(n_spks=2, kernel_size=2, enc_dim=16, feature_dim=8, hidden_dim=12, layer=2, segment_size=10)
    assert sorted(m.state_dict().keys()) == sorted(k[4:] for k in g.files if k.startswith("fsd."))
    m = quantize_model(m, dict(QCFG))
    assert list(m.state_dict().keys()) == list(g["sd_keys"])
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == g["sd0." + k].shape, k
    t = m.separator.DPT.row_transformer[0].transformer
    assert isinstance(t, TransformerEncoderLayer)
    assert isinstance(t.self_attn, QL.MultiheadAttentionQ) and isinstance(t.lstm, QL.LSTMQ) and isinstance(t.linear, QL.LinearQ)
    assert isinstance(t.norm1, QL.LayerNormQ) and isinstance(t.add_norm2, QL.AddQ)
    assert isinstance(m.encoder.conv1d_U, QL.Conv1dEncoderQ) and isinstance(m.encoder.relu, nn.Identity)
    assert isinstance(m.decoder.basis_signals, QL.LinearDecoderQ) and m.decoder.basis_signals.n_combiner == 2
    assert isinstance(m.separator.DPT.output[1], QL.Conv2dQ) and isinstance(m.separator.output[0], QL.Conv1dNlQ)
    full = quantize_model(create_model({"name": "DPTNet", "n_src": 2, "kernel_size": 2}), dict(QCFG))
    assert sum(p.numel() for p in full.parameters()) == 2895345
    with pytest.raises(Exception):          # no CPU fallback: the ops refuse host tensors
        full(torch.zeros(1, 1, 4000))


def test_sepformer_graph_rewrite_and_state_dict_layout(golden):
    """cfg 4: the Sepformer module tree quantizes into the reference's key set and order (825 keys at the fixture's size),
    incl. the positional table buffer, ConstQ, and the trainable residual decoder (train_res_dec=True)"""
    from fqss_amd.quantization.qat import qat_layers as QL
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.quantization.qat.models.sepformerq import MaskGenerator, SepformerQ
    from fqss_amd.smoke import QCFG
    g = golden("sep_tiny_step")
    m = SepformerQ(n_spks=2, kernel_size=16, stride=8, n_filters=16, n_repeats=1, n_heads=4, chunk_size=10)
    m.masker = MaskGenerator(2, 16, n_repeats=1, n_heads=4, chunk_size=10, n_ffn=32)
    m = quantize_model(m, dict(QCFG))
    assert list(m.state_dict().keys()) == list(g["sd_keys"])
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == g["sd0." + k].shape, k
    blk = m.masker.layers[0].intra_transformer_block
    assert isinstance(blk.pos.const, QL.ConstQ) and isinstance(blk.pos_add, QL.AddQ) and isinstance(blk.layers[0].ffn[1], QL.NlQ)
    assert isinstance(m.decoder, QL.ConvTr1dDecoderQ) and m.decoder.residual_error_block.train_res_dec
    assert m.decoder.residual_error_block.weight_fake_quantize_dec.axis == 1
    full = quantize_model(create_model({"name": "Sepformer", "n_src": 2, "kernel_size": 16, "stride": 8}), dict(QCFG))
    assert sum(p.numel() for p in full.parameters()) > 25_000_000


def test_rowmat_layouts():
    from fqss_amd.kernels import empty_act, rowmat
    assert rowmat(torch.empty(2, 3, 77)) == (6, 77, 77)
    b = empty_act((2, 6, 77), "cpu")
    assert b.shape == (2, 6, 77) and rowmat(b) == (12, 77, 80)
    assert rowmat(b.reshape(2, 2, 3, 77)) == (12, 77, 80) and rowmat(b.reshape(4, 3, 77)) == (12, 77, 80)
    assert rowmat(b.unsqueeze(1)) == (12, 77, 80)
    assert rowmat(torch.empty(4, 800).t()) is None and rowmat(torch.empty(2, 3, 77)[:, :, ::2]) is None
    assert rowmat(torch.empty(4, 1, 800)) == (4, 800, 800)
    # one row inside a padded buffer keeps the buffer's aligned row stride (the C side checks alignment even for a single row)
    assert rowmat(empty_act((1, 1, 77), "cpu")) == (1, 77, 80) and rowmat(torch.empty(1, 1, 77)) == (1, 77, 77)


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fqss_amd.parallel import Comm
    comm = Comm.from_env(device_type="cpu")
    assert comm.world == world and comm.backend == "gloo"
    lo, hi = comm.shard(16)
    g = torch.full((1000,), float(rank + 1))
    comm.all_reduce_sum(g)                      # the flat-gradient exchange
    t = torch.tensor([float(rank)])
    comm.all_reduce_max(t)                      # the bench's max-over-ranks timing
    b = torch.tensor([3.0 if rank == 0 else -1.0])
    comm.broadcast(b, 0)
    comm.barrier()
    q.put((rank, lo, hi, float(g[0]), float(g.sum()), float(t), float(b)))
    comm.close()


def test_data_parallel_plumbing_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    ps = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [(o[1], o[2]) for o in out] == [(0, 8), (8, 16)]           # disjoint batch shards
    assert all(o[3] == 3.0 and o[4] == 3000.0 for o in out)          # sum over ranks
    assert all(o[5] == 1.0 and o[6] == 3.0 for o in out)


def _ddp_dying_peer(rank, world, port, q):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FQSS_DIST_BACKEND="gloo", FQSS_DIST_TIMEOUT_S="30")
    from fqss_amd.parallel import Comm, CommError
    comm = Comm.from_env("cpu")
    g = torch.ones(8)
    comm.all_reduce_sum(g)                          # one good exchange first
    if rank == 1:
        os._exit(3)                                 # the peer dies between two steps
    try:
        comm.all_reduce_sum(torch.ones(8))
    except CommError as e:
        q.put(("CommError", str(e)))
        sys.exit(7)                                 # ... and the survivor leaves non-zero, nothing retried
    q.put(("no error", ""))


def test_exchange_failure_raises_and_ends_the_rank():
    """docs/history/DESIGN_rounds_1-5.md 6 "When the exchange fails" (VERDICT r03 next #7): a peer that dies makes the next collective raise CommError on the
    survivor (rank and operation named) within the configured timeout; the rank ends non-zero -- no retry, no hang"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    ps = [ctx.Process(target=_ddp_dying_peer, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    kind, msg = q.get(timeout=120)
    for p in ps:
        p.join(60)
    assert kind == "CommError" and "rank 0 of 2" in msg and "all_reduce(SUM)" in msg, (kind, msg)
    assert ps[0].exitcode == 7 and ps[1].exitcode == 3


_RANK_SCRIPT = """
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "allreduce":
    import torch
    from fqss_amd.parallel import Comm
    os.environ["FQSS_DIST_BACKEND"] = "gloo"
    c = Comm.from_env("cpu")
    t = torch.tensor([float(r + 1)])
    c.all_reduce_sum(t)
    c.barrier()
    if r == 0:
        print("SUM", t.item(), "WORLD", c.world, flush=True)
    else:
        print("this line must not reach the parent's stdout", flush=True)
    c.close()
elif mode == "die":
    if r == 1:
        sys.exit(5)
    time.sleep(120)          # a rank that would wait forever for its dead peer: the launcher must end it
elif mode.startswith("pids:"):
    import subprocess
    helper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])     # a rank's own child (a loader worker)
    with open(os.path.join(mode[5:], f"rank{r}.pid"), "w") as f:
        f.write(f"{os.getpid()} {helper.pid}")
    time.sleep(300)
"""


def test_launcher_refuses_more_ranks_than_gpus(monkeypatch):
    """`--gpus N` with fewer visible devices must fail loudly, not double ranks up on one card"""
    from fqss_amd import launch, parallel
    monkeypatch.delenv("FQSS_DIST_BACKEND", raising=False)
    monkeypatch.setattr(launch, "visible_gpus", lambda: 1)
    with pytest.raises(RuntimeError, match="2 ranks asked for, 1 GPU"):
        launch.spawn_ranks(2, [sys.executable, "-c", "pass"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert parallel.local_device(0) == 0
    with pytest.raises(RuntimeError, match="LOCAL_RANK 1 but 1 visible"):
        parallel.local_device(1)
    monkeypatch.setenv("FQSS_DIST_BACKEND", "gloo")          # the tests' two-ranks-on-GPU-0 mode
    assert parallel.local_device(1) == 0


def _gone(pid):
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(")")[-1].split()[0] == "Z"          # a zombie nobody has reaped yet is gone for this purpose
    except OSError:
        return True


@pytest.mark.parametrize("how", ["SIGTERM", "SIGHUP", "SIGKILL"])
def test_launcher_takes_its_ranks_along_when_it_is_ended(tmp_path, how):
    """a launcher ended by `timeout` / a scheduler / the harness must not leave N ranks holding the GPUs and the
    rendezvous port.  SIGTERM and SIGHUP run the launcher's clean-up (every rank's process GROUP gets SIGTERM, so the rank's own
    children go too); SIGKILL cannot be handled -- the ranks carry PR_SET_PDEATHSIG and die with their parent."""
    import signal
    import subprocess
    import time
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    drive = ("import sys; from fqss_amd.launch import spawn_ranks; "
             "sys.exit(spawn_ranks(2, [sys.executable, %r, sys.argv[1]], grace_s=3.0, gpus=False))" % str(script))
    p = subprocess.Popen([sys.executable, "-c", drive, f"pids:{tmp_path}"], env=env, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not all((tmp_path / f"rank{r}.pid").exists() and (tmp_path / f"rank{r}.pid").read_text().count(" ") == 1 for r in range(2)):
        assert time.time() - t0 < 60 and p.poll() is None
        time.sleep(0.1)
    pids = [int(x) for r in range(2) for x in (tmp_path / f"rank{r}.pid").read_text().split()]
    assert not any(_gone(q) for q in pids)
    p.send_signal(getattr(signal, how))
    rc = p.wait(30)
    assert rc == {"SIGTERM": 128 + 15, "SIGHUP": 128 + 1, "SIGKILL": -9}[how], (rc, p.stderr.read()[-500:])
    t0 = time.time()
    # SIGKILL: the ranks get SIGTERM through PDEATHSIG; their helpers are then orphans of a dead session leader -- the group rule does
    # not reach them in that one case, which is checked for the ranks only
    want = pids if how != "SIGKILL" else pids[0::2]
    while not all(_gone(q) for q in want):
        assert time.time() - t0 < 15, [q for q in want if not _gone(q)]
        time.sleep(0.1)
    for q in pids:                       # tidy up whatever the SIGKILL case left
        try:
            os.kill(q, signal.SIGKILL)
        except OSError:
            pass



def test_launcher_starts_ranks_and_propagates_a_failing_rank(tmp_path):
    """fqss_amd/launch.spawn_ranks (what `bench.py --gpus N` and `python -m fqss_amd.train` use when no launcher is around them; reference:
    tasnet_musdbhq_trainer.py:17-57 -- one Popen per GPU, every rank ended when one dies): rank environment, rendezvous over gloo,
    rank 0 alone owns stdout; a rank that exits non-zero ends the others and its code is the launcher's."""
    import subprocess
    import time
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    drive = ("import sys; from fqss_amd.launch import spawn_ranks, already_launched; assert not already_launched(); "
             "sys.exit(spawn_ranks(2, [sys.executable, %r, sys.argv[1]], grace_s=3.0, gpus=False))" % str(script))
    p = subprocess.run([sys.executable, "-c", drive, "allreduce"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if not l.startswith("[Gloo]")]       # (gloo announces its connections on stdout)
    assert lines == ["SUM 3.0 WORLD 2"], p.stdout
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", drive, "die"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 5 and "rank 1 of 2 exited with 5" in p.stderr, (p.returncode, p.stderr[-500:])
    assert time.time() - t0 < 60          # rank 0 did not sleep its 120 s out


def test_checkpoint_interchange_lightning_ckpt(tmp_path):
    """a Lightning `.ckpt` of the asteroid env ({"state_dict": {"model.<key>", "fmodel.<key>"}}) loads through
    `load_pretrain`'s order-based mapping with the teacher's entries dropped (convtasnetq.py:225-237); `create_pretrained_model`
    falls back to it when the strict load fails (load_model.py:76-102); a mismatching checkpoint is refused"""
    import torch
    from fqss_amd.quantization.qat.models.load_model import create_model, create_pretrained_model, quantize_model
    from fqss_amd.smoke import QCFG
    cfg = {"name": "ConvTasNet", "n_src": 2, "kernel_size": 16, "stride": 8}
    torch.manual_seed(3)
    src = quantize_model(create_model(cfg), dict(QCFG))
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    ck = {"state_dict": {**{"model." + k: v.clone() for k, v in src.state_dict().items()},
                         **{"fmodel." + k: v.clone() for k, v in create_model(cfg).state_dict().items()}}}
    path = tmp_path / "epoch=3.ckpt"
    torch.save(ck, path)
    dst = quantize_model(create_model(cfg), dict(QCFG))
    dst.load_pretrain(str(path))
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    via = create_pretrained_model(dict(cfg, model_path=str(path), quantization=dict(QCFG)))
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), via.state_dict().values()))
    bad = {"state_dict": {k: v for i, (k, v) in enumerate(ck["state_dict"].items()) if i > 0}}
    torch.save(bad, tmp_path / "bad.ckpt")
    with pytest.raises(AssertionError):
        dst.load_pretrain(str(tmp_path / "bad.ckpt"))


def test_library_has_no_packed_fp32_op_with_a_swapped_second_source():
    """gfx950 erratum guard (docs/history/DESIGN_rounds_1-5.md 9, tools/ubench/pk_opsel_repro.hip): v_pk_add/mul/fma_f32 whose LOW result takes the HIGH half of
    the second source (op_sel[1] = 1) returns wrong values in lanes 48-63 while another stream's bf16-MFMA GEMM is resident on the
    same CU -- the cause of both two-stream events of round 2 (the deleted four-frames-per-lane decoder carried 128-256 of them,
    the branchy bias sums of k_mulq_bwd two).  hipcc's SLP vectorizer makes them out of scalar code; the library is built with
    -fno-slp-vectorize and this test disassembles every gfx950 code object of the built .so and requires that none is left."""
    import importlib.util
    from fqss_amd import _lib
    spec = importlib.util.spec_from_file_location("scan_isa", os.path.join(ROOT, "tools", "scan_isa.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    n_obj, unsafe, packed = scan.census(_lib.SO_PATH)
    assert n_obj >= 20, n_obj                     # one code object per translation unit with kernels
    assert not unsafe, f"packed fp32 instructions with op_sel[1] = 1 in: {unsafe}"
    # the matcher itself
    assert scan.UNSAFE.search("v_pk_add_f32 v[4:5], v[8:9], v[16:17] op_sel:[0,1] op_sel_hi:[1,0]")
    assert scan.UNSAFE.search("v_pk_fma_f32 v[0:1], s[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
    assert not scan.UNSAFE.search("v_pk_add_f32 v[4:5], v[8:9], v[16:17] op_sel:[1,0] op_sel_hi:[0,1]")
    assert not scan.UNSAFE.search("v_pk_add_f32 v[50:51], v[50:51], 0 op_sel_hi:[1,0]")


def test_pmc_reduce_matches_kernel_symbols_and_fails_loudly():
    """tools/roofline_probe.py --reduce: manifest names ("k_qgemm<1>") are matched to profiler names ("fqss::k_qgemm<1, 3>(...)") on the
    kernel SYMBOL; a case without its dispatches is an error (round 2 silently lost 11 of 15 kernels to a substring match)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("roofline_probe", os.path.join(ROOT, "tools", "roofline_probe.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    assert rp._is("k_qgemm", "void fqss::k_qgemm<1, 3>(fqss::QGemmArgs)")
    assert rp._is("k_qwgrad2", "void fqss::k_qwgrad2<3>(fqss::QGemmArgs)")
    assert not rp._is("k_qwgrad", "void fqss::k_qwgrad2<3>(fqss::QGemmArgs)")
    assert rp._is("k_axpby", "fqss::k_axpby(float const*, float const*, float, float, float*, long)")
    rows = [(1, "void fqss::k_minmax(...)", 1.0), (2, "void fqss::k_qgemm<1, 3>(fqss::QGemmArgs)", 10.0), (3, "void fqss::k_qgemm<1, 3>(fqss::QGemmArgs)", 14.0),
            (4, "void fqss::k_gnq_bwd_rows(...)", 5.0), (5, "void fqss::k_gnq_bwd_apply<true>(...)", 7.0)]
    man = [dict(kernel="k_qgemm<1>", label="dgrad", iters=2), dict(kernel="k_gnq_bwd_rows+apply<true>", label="gln bwd", iters=1)]
    assert rp._assign(rows, man) == [12.0, 12.0]
    with pytest.raises(SystemExit):
        rp._assign(rows, [dict(kernel="k_qwgrad2", label="wgrad", iters=1)])


def test_residual_fork_bookkeeping_with_a_late_consumer(monkeypatch):
    """ops._ForkState / Fork2.backward -- the dgrad q-GEMM of one fork branch adds the OTHER branch's gradient
    in its epilogue.  Topology the GPU gates never build: TWO element-wise consumers on the other branch with the conv's backward
    ordered BETWEEN them (EwQ#2 leaves o2, the conv fuses own + o2, EwQ#1 leaves o1).  Fork2.backward must subtract what the conv
    really added (the snapshot o2), not the later-accumulated o1 + o2.  Bookkeeping only: K.axpby is replaced by the torch
    expression it computes."""
    from fqss_amd import ops
    monkeypatch.setattr(ops.K, "axpby", lambda a, b, sb, sa=1.0: sa * a + sb * b)
    own, o1, o2 = torch.randn(3, 5), torch.randn(3, 5), torch.randn(3, 5)

    class Ctx:
        pass

    def run(order):
        fk = ops._ForkState()
        fused = None
        for ev in order:
            if ev == "o1":
                fk.leave(o1, 1)
            elif ev == "o2":
                fk.leave(o2, 1)
            else:
                fused = fk.take(0, own.shape)
        g0 = own + fused if fused is not None else own               # the conv's dgrad (+ epilogue add)
        g1 = (o1 + o2) if ("o1" in order and "o2" in order) else (o2 if "o2" in order else o1)   # autograd's own sum on branch 1
        if order in (["o2", "conv"], ["o1", "conv"]):
            g1 = o2 if order[0] == "o2" else o1                      # single consumer: autograd hands over the very tensor
        ctx = Ctx()
        ctx.fk = fk
        g, _ = ops.Fork2.backward(ctx, g0, g1)
        assert fk.other is None and fk.taken is None and fk.fused_branch is None
        return g, (own + g1)

    for order in (["o2", "conv", "o1"], ["o2", "o1", "conv"], ["o2", "conv"], ["conv", "o2", "o1"], ["o1", "conv"]):
        g, want = run(order)
        torch.testing.assert_close(g, want, msg=str(order))
