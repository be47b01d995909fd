"""Evaluation side on the GPU: process.model_infer / swap_channel_order / SI-SNR against vectors from
the REAL reference's process.py (tests/golden/infer.npz), through the C ABI (fqss_sisnr_matrix, fqss_infer_ola, fqss_infer_normalize)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TINY = dict(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _model(g):
    from fqss_amd.quantization.qat.models.convtasnetq import ConvTasNetQ
    from fqss_amd.quantization.qat.models.load_model import enable_observer, quantize_model
    from fqss_amd.smoke import QCFG
    m = quantize_model(ConvTasNetQ(**TINY), dict(QCFG))
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd.")}, strict=True)
    enable_observer(m, False)
    return m.cuda().eval()


def test_sisnr_matrix_and_swap(golden):
    from fqss_amd import kernels as K
    from fqss_amd.process import si_snr, swap_channel_order
    g = golden("infer")
    est, clean = T(g["swap.in"]).cuda(), T(g["clean"]).cuda()
    db, mp = K.sisnr_matrix(est, clean, want_map=True)
    np.testing.assert_allclose(db.cpu().numpy(), g["sisnr"], rtol=1e-5, atol=1e-4)
    assert mp.cpu().tolist() == [[1, -1], [0, -1]]            # both estimates move (and flip sign)
    assert torch.equal(swap_channel_order(est, clean).cpu(), T(g["swap.out"]))
    np.testing.assert_allclose(si_snr(est, clean.flip(0)).cpu().numpy(), [g["sisnr"][0, 1], g["sisnr"][1, 0]], rtol=1e-5, atol=1e-4)
    # three sources (later estimates override earlier claims, unclaimed targets keep their own estimate): against the oracle
    import oracle.fqss_oracle as O
    e3 = torch.stack([clean[0], clean[1] * 0.5, clean[0] + clean[1]])
    c3 = torch.stack([clean[1], clean[0], clean[0] - clean[1]])
    assert torch.equal(swap_channel_order(e3, c3).cpu(), O.swap_channel_order(e3.cpu(), c3.cpu()))


def test_model_infer_matches_the_reference(golden):
    from fqss_amd.process import metric_evaluation, model_infer
    g = golden("infer")
    m = _model(g)
    mix, clean = T(g["mix"]).cuda(), T(g["clean"]).cuda()
    outs = dict(whole=model_infer(m, mix, n_srcs=2), chunked=model_infer(m, mix, n_srcs=2, segment=1000, overlap=0.25, target=clean),
                chunked_nt=model_infer(m, mix, n_srcs=2, segment=1000, overlap=0.25))
    for key, got in outs.items():
        want = g[key]
        assert tuple(got.shape) == want.shape
        err = np.abs(got.cpu().numpy() - want)
        scale = np.abs(want).max()
        stats = (key, float(err.max() / scale), float(np.sqrt(np.mean(err ** 2)) / scale), float(np.mean(err > 1e-3 * scale)))
        # eval mode quantizes: a different accumulation order flips a few 8-bit bins inside the network, each moving the samples in
        # its receptive field by a few output steps (the reference's own backends differ the same way)
        assert stats[1] <= 0.08 and stats[2] <= 5e-3 and stats[3] <= 0.05, stats
    s, _, _ = metric_evaluation(outs["chunked"], clean)
    assert np.isfinite(s)
    with pytest.raises(RuntimeError):
        model_infer(m, mix, device="cpu")


def test_data_side_augmentation_batched(golden):
    """process.generate_2mix_snr / generate_3mix_snr / generate_mix_noise + max_clip, batched on the device
    (fqss_snr_mix), against the REAL reference's process.py run item by item"""
    from fqss_amd.process import generate_2mix_snr, generate_3mix_snr, generate_mix_noise
    g = golden("data_aug")
    s = T(g["s"]).cuda()
    i, j, snr = g["cases"][:, 0].astype(int), g["cases"][:, 1].astype(int), T(g["cases"][:, 2]).cuda()
    a, b = s[i], s[j]
    # 10^(snr/10), the energy means and the peak come out of different reduction orders: 2e-6 relative
    np.testing.assert_allclose(generate_2mix_snr(a, b, snr).cpu().numpy(), g["mix2"], rtol=5e-6, atol=2e-7)
    np.testing.assert_allclose(generate_2mix_snr(a, b, snr, clip=False).cpu().numpy(), g["mix2_noclip"], rtol=5e-6, atol=2e-7)
    np.testing.assert_allclose(generate_mix_noise(a, b, snr.abs() + 6.0).cpu().numpy(), g["noise"], rtol=5e-6, atol=2e-7)
    np.testing.assert_allclose(generate_3mix_snr(s[0], s[1], s[2], 1.5, -2.0).cpu().numpy(), g["mix3"], rtol=5e-6, atol=2e-7)
    np.testing.assert_allclose(generate_2mix_snr(torch.zeros(4000, device="cuda"), s[1], 3.0).cpu().numpy(), g["zero"], rtol=5e-6, atol=2e-7)
    assert float(generate_2mix_snr(a, b, snr).abs().max()) <= 0.9 + 1e-6


def test_infer_runner_graph_replay_is_bit_identical(golden):
    """the serving form of the quantized forward: eval mode on the codes-only dataflow, captured per request shape; bit-identical to the
    plain eval forward, for two shapes served alternately"""
    from fqss_amd.runtime import InferRunner
    g = golden("infer")
    m = _model(g)
    run = InferRunner(m)
    mix = T(g["mix"]).cuda()
    xs = [mix[:, :2400].reshape(1, 1, -1).contiguous(), mix[:, :3000].reshape(1, 1, -1).repeat(2, 1, 1).contiguous()]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
    for _ in range(2):
        for x, w in zip(xs, want):
            assert torch.equal(run(x), w)
    assert len(run._graphs) == 2


def test_polyphase_resampler_and_librimix_batches(tmp_path):
    """SURVEY 8(f) rank 4, the rest of the data side: the 16 -> 8 kHz resampler of the LibriMix dataset (librimix_dataset.py:54, one
    launch for a whole batch of clips) against the oracle's fp64 restatement of torchaudio's published sinc kernel, and the CSV-driven
    batch assembler (same item contract as librimix_dataset.py:93-170) on synthetic PCM16 files"""
    import wave
    import pandas as pd
    import oracle.fqss_oracle as O
    from fqss_amd import kernels as K
    from fqss_amd.train_env.asteroid_librimix.librimix_dataset import LibriMix
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(5, 24001, generator=gen) * 0.1
    for orig, new in ((16000, 8000), (44100, 16000), (8000, 16000)):
        y = K.resample(x.cuda(), orig, new).cpu()
        ref = O.resample_sinc(x, orig, new)
        assert y.shape == ref.shape
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=0, atol=3e-7)
    # a tone below / above the new Nyquist frequency: kept / removed
    t = torch.arange(16000) / 16000.0
    keep, kill = torch.sin(2 * np.pi * 1000 * t), torch.sin(2 * np.pi * 5000 * t)
    yk = K.resample(torch.stack([keep, kill]).cuda(), 16000, 8000).cpu()
    assert abs(float(yk[0, 100:-100].pow(2).mean().sqrt()) - 2 ** -0.5) < 1e-2 and float(yk[1, 100:-100].abs().max()) < 2e-2
    # ---- a tiny on-disk LibriMix: 3 utterances (one too short), 2 sources, 16 kHz PCM16
    def write(path, sig):
        with wave.open(str(path), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((np.clip(sig, -1, 1 - 2 ** -15) * 32768).astype("<i2").tobytes())
    rows, rs = [], np.random.RandomState(0)
    for i, n in enumerate((20000, 9000, 26000)):
        s1, s2 = 0.2 * rs.randn(n).astype(np.float32), 0.1 * rs.randn(n).astype(np.float32)
        paths = [tmp_path / f"s1_{i}.wav", tmp_path / f"s2_{i}.wav", tmp_path / f"mix_{i}.wav"]
        for p_, sig in zip(paths, (s1, s2, s1 + s2)):
            write(p_, sig)
        rows.append({"mixture_ID": i, "mixture_path": str(paths[2]), "source_1_path": str(paths[0]), "source_2_path": str(paths[1]), "length": n})
    pd.DataFrame(rows).to_csv(tmp_path / "mixture_train_mix_clean.csv", index=False)
    ds = LibriMix(str(tmp_path), task="sep_clean", sample_rate=16000, resample=0.5, n_src=2, segment=1)
    assert len(ds) == 2                                             # the 9000-sample utterance is dropped (< 1 s)
    mix, src = ds[0]
    assert mix.shape == (1, 8000) and src.shape == (2, 8000) and mix.is_cuda
    # without augmentation the mixture file is the sum of the sources (to PCM16 rounding) -- and resampling is linear
    np.testing.assert_allclose(mix[0].cpu().numpy(), src.sum(0).cpu().numpy(), atol=3e-4)
    mixes, srcs = ds.batch([0, 1, 1])
    assert mixes.shape == (3, 1, 8000) and srcs.shape == (3, 2, 8000)
    ds_aug = LibriMix(str(tmp_path), task="sep_clean", sample_rate=16000, resample=0.5, n_src=2, segment=1,
                      augmentation_cfg={"distribution": "uniform", "param0": -2.5, "param1": 2.5, "prob": 1.0})
    mixes, srcs = ds_aug.batch([0, 1])
    e = lambda v: v.pow(2).mean(-1)
    # generate_2mix_snr rescales ONE source so that the pair sits at the drawn SNR in [-2.5, 2.5] dB: the mixture is a1 s1 + a2 s2
    a = torch.linalg.lstsq(srcs.transpose(1, 2), mixes.transpose(1, 2)).solution.squeeze(-1)      # [B, 2] gains
    snr = 10 * torch.log10(e(srcs[:, 0] * a[:, :1]) / e(srcs[:, 1] * a[:, 1:]))
    assert bool(((snr >= -2.6) & (snr <= 2.6)).all()) and bool((a.min(1).values <= 1.0 + 1e-4).all()), (snr, a)


def test_wsdr_loss_names_match_their_formulas():
    """SDR / PairwiseWSDR of train_env/asteroid_librimix/wsdr.py:10-95 (evaluation forms over the moment kernel) against the
    formulas written out in fp64 torch, every sdr_type / zero_mean / take_log combination, with and without weights"""
    from fqss_amd.train_env.asteroid_librimix import wsdr
    g = torch.Generator().manual_seed(11)
    B, S, T = 3, 2, 4001
    tgt = torch.randn(B, S, T, generator=g) * 0.3 + 0.05
    est = tgt[:, [1, 0]] * 0.7 + 0.2 * torch.randn(B, S, T, generator=g) - 0.02
    wts = torch.rand(B, generator=g) + 0.5
    EPS = 1e-8

    def pair_ref(kind, zero_mean, take_log, weights):
        t, e = tgt.double(), est.double()
        if zero_mean:
            t, e = t - t.mean(2, keepdim=True), e - e.mean(2, keepdim=True)
        st, se = t.unsqueeze(1), e.unsqueeze(2)
        if kind in ("sisdr", "sdsdr"):
            proj = (se * st).sum(3, keepdim=True) * st / ((st ** 2).sum(3, keepdim=True) + EPS)
        else:
            proj = st.repeat(1, S, 1, 1)
        noise = se - st if kind in ("sdsdr", "snr") else se - proj
        r = (proj ** 2).sum(3) / ((noise ** 2).sum(3) + EPS)
        if weights is not None:
            r = r * weights.double()[:, None, None]
        return 10 * torch.log10(r + EPS) if take_log else -r

    def sdr_ref(kind, zero_mean, take_log, weights):
        t, e = tgt.double(), est.double()
        if zero_mean:
            t, e = t - t.mean(-1, keepdim=True), e - e.mean(-1, keepdim=True)
        proj = (e * t).sum(-1, keepdim=True) * t / ((t ** 2).sum(-1, keepdim=True) + EPS)
        noise = e - proj if kind == "sisdr" else e - t
        r = (proj ** 2).sum(-1) / ((noise ** 2).sum(-1) + EPS)
        if weights is not None:
            r = r * weights.double()[:, None, None]       # the reference's own broadcast (wsdr.py:37): [B, n_src] x [B, 1, 1] -> [B, B, n_src]
        r = r.mean()
        return 10 * torch.log10(r + EPS) if take_log else r

    ed, td, wd = est.cuda(), tgt.cuda(), wts.cuda()
    for zero_mean in (True, False):
        for take_log in (True, False):
            for w_host, w_dev in ((None, None), (wts, wd)):
                for kind in ("snr", "sisdr", "sdsdr"):
                    got = wsdr.PairwiseWSDR(kind, zero_mean=zero_mean, take_log=take_log)(ed, td, w_dev).cpu().double()
                    torch.testing.assert_close(got, pair_ref(kind, zero_mean, take_log, w_host), rtol=2e-5, atol=2e-5)
                for kind in ("sisdr", "sdr"):
                    got = wsdr.SDR(kind, zero_mean=zero_mean, take_log=take_log)(ed, td, w_dev).cpu().double()
                    torch.testing.assert_close(got, sdr_ref(kind, zero_mean, take_log, w_host), rtol=2e-5, atol=2e-5)
    assert wsdr.pairwise_wsisdr(ed, td).shape == (B, S, S)
    with pytest.raises(TypeError):
        wsdr.PairwiseWSDR("sisdr")(ed[0], td[0])


def test_dynamic_activation_quantizer_matches_torch_affine():
    """TorchDymActivationFakeQuantize (qat_quant.py:56-72) on the device against torch.fake_quantize_per_tensor_affine on the host"""
    from fqss_amd.quantization.qat import qat_quant as QQ
    from fqss_amd.quantization.qat import qat_utils as QU
    x = torch.randn(7, 33, 129, generator=torch.Generator().manual_seed(3)) * 1.7 + 0.3
    src = QQ.GradientActivationFakeQuantize(True)
    src.factor = 0.9
    dq = QU.torch_dym_activation_quantizer(src)
    assert isinstance(dq, QQ.TorchDymActivationFakeQuantize)
    y = dq(x.cuda()).cpu()
    mn, mx = 0.9 * x.min(), 0.9 * x.max()
    scale = float((mx - mn) / 255)
    zp = int(torch.round(mn / scale))
    zp = -zp if mn < 0 else zp
    ref = torch.fake_quantize_per_tensor_affine(x, scale=scale, zero_point=zp, quant_min=0, quant_max=255)
    assert torch.equal(y, ref)
    holder = torch.nn.Sequential(torch.nn.Identity())
    QU.replace_dym_activation_quantizer(holder, "0", src)
    assert isinstance(holder[0], QQ.TorchDymActivationFakeQuantize)
