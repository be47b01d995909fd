"""True-integer export wrappers on the GPU against the REAL reference's TorchWeightFakeQuantize /
TorchActivationFakeQuantize (tests/golden/export.npz): de-quantized values and integer codes bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_weight_and_activation_wrappers_bit_exact(golden):
    from fqss_amd.quantization.qat import qat_quant as Q
    g = golden("export")
    for tag in ("w0", "w1", "w2d"):
        w, axis = T(g[tag + ".w"]).cuda(), int(g[tag + ".axis"])
        q = Q.GradientWeightFakeQuantize(True, tuple(w.shape), ch_out_idx=axis).cuda()
        with torch.no_grad():
            q.min_range.copy_(T(g[tag + ".min"])); q.max_range.copy_(T(g[tag + ".max"]))
        t = Q.TorchWeightFakeQuantize(q)
        np.testing.assert_array_equal(t.scales.cpu().numpy(), g[tag + ".scales"])
        assert torch.equal(t(w).cpu(), T(g[tag + ".y"])), tag
        assert torch.equal(t.integer(w).cpu(), T(g[tag + ".codes"])), tag
    for tag in ("a0", "a1", "a2", "a3"):
        x = T(g[tag + ".x"]).cuda()
        q = Q.GradientActivationFakeQuantize(True).cuda()
        with torch.no_grad():
            q.min_range.fill_(float(g[tag + ".range"][0])); q.max_range.fill_(float(g[tag + ".range"][1]))
        t = Q.TorchActivationFakeQuantize(q)
        assert t.scale == float(g[tag + ".scale"]) and t.zero_point == int(g[tag + ".zero_point"]), tag
        y = t(x)
        assert torch.equal(y.cpu(), T(g[tag + ".y"])), tag
        codes = t.integer(x)
        assert codes.dtype == torch.uint8
        np.testing.assert_array_equal(((codes.float() - t.zero_point) * t.scale).cpu().numpy(), g[tag + ".y"])


def test_export_integer_state_and_replace():
    from fqss_amd.quantization.qat import qat_quant as Q
    from fqss_amd.quantization.qat.qat_utils import replace_activation_quantizer, replace_weight_quantizer
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.smoke import QCFG
    m = quantize_model(create_model({"name": "ConvTasNet", "n_src": 2, "kernel_size": 16, "stride": 8}), dict(QCFG)).cuda()
    st = Q.export_integer_state(m)
    assert len(st) == 301 and st["encoder.weight_fake_quantize"]["axis"] == 0 and st["decoder.weight_fake_quantize"]["axis"] == 1
    assert st["masker.bottleneck.1.activation_fake_quantize"]["quant_max"] == 255
    layer = m.masker.bottleneck[1]
    replace_weight_quantizer(layer, "weight_fake_quantize", layer.weight_fake_quantize)
    replace_activation_quantizer(layer, "activation_fake_quantize", layer.activation_fake_quantize)
    assert isinstance(layer.weight_fake_quantize, Q.TorchWeightFakeQuantize) and isinstance(layer.activation_fake_quantize, Q.TorchActivationFakeQuantize)
    w = layer.conv1d.weight.detach()
    assert layer.weight_fake_quantize.integer(w).dtype == torch.int8


def test_integer_checkpoint_round_trip(tmp_path):
    """SURVEY 8(f) rank 3 leftover (VERDICT r02 missing #5): the trained model STORED as integers -- int8 weight codes + per-channel
    steps, activation ranges + affine parameters, the unquantized float tensors -- and restored: W_q = delta * code is bit for bit the
    weight the QAT forward multiplies with, so the eval output of the restored model equals the original's; the file is a quarter of
    the fp32 state_dict."""
    import os
    from fqss_amd.quantization.qat import qat_quant as Q
    from fqss_amd.quantization.qat.qat_utils import load_integer_checkpoint, save_integer_checkpoint, weight_quantizer_owners
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.data import synth_batch
    from fqss_amd.smoke import QCFG
    torch.manual_seed(0)
    cfg = {"name": "ConvTasNet", "n_src": 2, "kernel_size": 16, "stride": 8}
    m = quantize_model(create_model(dict(cfg)), dict(QCFG)).cuda().train()
    x, _ = synth_batch(2, 4000, seed=5, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            m(x)                                                     # weight observers record, activation ranges move off their init
    for q in m.modules():
        if isinstance(q, Q.GradientActivationFakeQuantize):
            q.n_iter = q.max_observations
    m.eval()
    with torch.no_grad():
        y0 = m(x)
    p_int, p_f32 = str(tmp_path / "model.int8.pth"), str(tmp_path / "model.f32.pth")
    save_integer_checkpoint(m, p_int)
    torch.save(m.state_dict(), p_f32)
    assert len(weight_quantizer_owners(m)) == 101
    assert os.path.getsize(p_int) < 0.30 * os.path.getsize(p_f32), (os.path.getsize(p_int), os.path.getsize(p_f32))
    st = torch.load(p_int, weights_only=True)
    assert st["format"] == "fqss-int8-v1" and all(e["codes"].dtype == torch.int8 for e in st["weights"].values())
    torch.manual_seed(123)                                            # a DIFFERENT init: everything must come from the file
    m2 = quantize_model(create_model(dict(cfg)), dict(QCFG)).cuda()
    load_integer_checkpoint(m2, p_int).eval()
    with torch.no_grad():
        y1 = m2(x)
    assert torch.equal(y0, y1)
    # the stored codes ARE the codes of the training quantizer of the restored weights (idempotence of the grid)
    from fqss_amd import kernels as K
    wqm, w, name = weight_quantizer_owners(m2)[5]
    _, codes = K.wq_fwd(w.detach(), wqm.axis, wqm.min_range.detach(), wqm.max_range.detach(), want_idx=True)
    assert torch.equal(codes.cpu(), st["weights"][name]["codes"])
