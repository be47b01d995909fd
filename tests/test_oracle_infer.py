"""Pins the evaluation-side oracle functions (oracle/fqss_oracle.py: si_snr, swap_channel_order, model_infer; SURVEY.md §8(f)
rank 1) against vectors produced by the REAL reference's process.py (tools/make_goldens_infer.py).  CPU only."""
import numpy as np
import torch

import oracle.fqss_oracle as O

torch.set_num_threads(2)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _student(g):
    s = O.StudentConvTasNetQ({k[3:]: T(g[k]) for k in g.files if k.startswith("sd.")}, layers_per_stack=2)
    s.enable_observer(False)
    s.leave_observer_phase()
    return s


def test_sisnr_and_swap(golden):
    g = golden("infer")
    est, clean = T(g["swap.in"]), T(g["clean"])
    got = np.array([[float(O.si_snr(est[p:p + 1], clean[q])) for q in range(2)] for p in range(2)])
    np.testing.assert_allclose(got, g["sisnr"], rtol=1e-5, atol=1e-4)
    assert torch.equal(O.swap_channel_order(est, clean), T(g["swap.out"]))


def test_model_infer(golden):
    g = golden("infer")
    s = _student(g)
    fwd = lambda x: s.forward(x)
    mix, clean = T(g["mix"]), T(g["clean"])
    with torch.no_grad():
        whole = O.model_infer(fwd, mix, 2)
        chunked = O.model_infer(fwd, mix, 2, segment=1000, overlap=0.25, target=clean)
        chunked_nt = O.model_infer(fwd, mix, 2, segment=1000, overlap=0.25)
    for got, key in ((whole, "whole"), (chunked, "chunked"), (chunked_nt, "chunked_nt")):
        want = g[key]
        err = np.abs(got.numpy() - want)
        # eval mode quantizes: fp32 noise flips a few 8-bit bins, each moving a handful of output samples by one output step
        assert err.max() <= 0.02 * np.abs(want).max() and np.mean(err > 1e-6) < 0.02, (key, err.max(), np.mean(err > 1e-6))


def test_export_wrappers(golden):
    """the affine (scale, zero-point) export form vs the REAL reference's TorchWeightFakeQuantize /
    TorchActivationFakeQuantize (tests/golden/export.npz, tools/make_goldens_export.py): values and integer codes bit-exact"""
    g = golden("export")
    for tag in ("w0", "w1", "w2d"):
        y, codes, scales = O.weight_export(T(g[tag + ".w"]), T(g[tag + ".min"]), T(g[tag + ".max"]), int(g[tag + ".axis"]))
        assert torch.equal(scales, T(g[tag + ".scales"])) and torch.equal(y, T(g[tag + ".y"]))
        assert torch.equal(codes.to(torch.int8), T(g[tag + ".codes"]))
    for tag in ("a0", "a1", "a2", "a3"):
        lo, hi = (float(v) for v in g[tag + ".range"])
        y, codes, scale, zp = O.act_export(T(g[tag + ".x"]), lo, hi)
        assert scale == float(g[tag + ".scale"]) and zp == int(g[tag + ".zero_point"])
        assert torch.equal(y, T(g[tag + ".y"])) and int(codes.min()) >= 0 and int(codes.max()) <= 255


def test_data_side_augmentation(golden):
    """the SNR augmentation vs the REAL reference's process.py (tests/golden/data_aug.npz)"""
    g = golden("data_aug")
    s = T(g["s"])
    for n, (i, j, snr) in enumerate(g["cases"]):
        a, b = s[int(i)], s[int(j)]
        np.testing.assert_allclose(O.generate_2mix_snr(a, b, float(snr)).numpy(), g["mix2"][n], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(O.generate_2mix_snr(a, b, float(snr), clip=False).numpy(), g["mix2_noclip"][n], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(O.generate_mix_noise(a, b, abs(float(snr)) + 6.0).numpy(), g["noise"][n], rtol=1e-6, atol=1e-7)
    m3 = O.generate_2mix_snr(s[0], O.generate_2mix_snr(s[1], s[2], -2.0), 1.5)
    np.testing.assert_allclose(m3.numpy(), g["mix3"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.generate_2mix_snr(torch.zeros(4000), s[1], 3.0).numpy(), g["zero"], rtol=1e-6, atol=1e-7)
