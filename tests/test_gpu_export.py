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
