"""Model factory / quantize driver with the reference's entry points (load_model.py:11-102)."""
import torch

from ..qat_layers import LayerQ
from ..qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize
from .convtasnetq import ConvTasNetQ
from .dptnetq import DPTNetQ
from .htdemucsq import HTDemucsQ
from .sepformerq import SepformerQ


def set_mac_op(model, mode=False):
    for _, m in model.named_modules():
        if isinstance(m, LayerQ):
            m.do_mac_op = mode


def enable_observer(model, mode=False):
    for _, m in model.named_modules():
        if isinstance(m, (GradientWeightFakeQuantize, GradientActivationFakeQuantize)):
            m.enable_observer(mode)


def create_model(model_cfg):
    name = model_cfg["name"]
    if name == "ConvTasNet":
        return ConvTasNetQ(n_spks=model_cfg.get("n_src", 1), kernel_size=model_cfg.get("kernel_size", 32),
                           stride=model_cfg.get("stride", 16))
    if name == "DPTNet":
        return DPTNetQ(n_spks=model_cfg.get("n_src", 2), kernel_size=model_cfg.get("kernel_size", 2))
    if name == "Sepformer":
        return SepformerQ(n_spks=model_cfg.get("n_src", 2), kernel_size=model_cfg.get("kernel_size", 16),
                          stride=model_cfg.get("stride", 8))
    if name == "HTDemucs":
        path = model_cfg.get("model_path", None)
        if path:
            return HTDemucsQ(**_load_any(path)["kwargs"])
        return HTDemucsQ(sources=model_cfg.get("sources", ["drums", "bass", "other", "vocals"]))
    if name == "ConvTasNetMusic":
        raise NotImplementedError(f"{name}: the tasnet_musdbhq environment is outside SURVEY.md §8; this build serves ConvTasNet, DPTNet, "
                                  "Sepformer and HTDemucs")
    raise AssertionError("Model {} is not supported!".format(name))


def quantize_model(model, quant_cfg):
    if quant_cfg.get("qat", False):
        g = quant_cfg.get
        model.set_splitter_combiner(g("n_splitter", 1), g("n_combiner", 1))
        model.quantize_model(gradient_based=g("gradient_based", True), weight_quant=g("weight_quant", True),
                             weight_n_bits=g("weight_n_bits", 8), act_quant=g("act_quant", True),
                             act_n_bits=g("act_n_bits", 8), inout_nl_quant=g("inout_nl_quant", False),
                             in_quant=g("in_quant", False), in_act_n_bits=g("in_act_n_bits", 8),
                             out_quant=g("out_quant", False), out_act_n_bits=g("out_act_n_bits", 8))
        enable_observer(model, g("observer", False))
    return model


def load_checkpoint(path):
    """a TRUSTED local checkpoint (the reference calls plain torch.load): `weights_only=False`, because reference / demucs-style
    packages ({'state', 'kwargs', ...}) pickle non-tensor objects (e.g. fractions.Fraction for `segment`) that torch >= 2.6 refuses
    by default"""
    return torch.load(path, map_location="cpu", weights_only=False)


def _load_any(path):
    if path.startswith("https"):
        return torch.hub.load_state_dict_from_url(path, map_location="cpu", check_hash=True)
    return load_checkpoint(path)


def create_pretrained_model(model_cfg):
    """quantize FIRST, then load a quantized checkpoint (val/infer path, load_model.py:76-102)"""
    model = quantize_model(create_model(model_cfg), model_cfg["quantization"])
    path = model_cfg.get("model_path", None)
    if path is None:
        return model
    sd = _load_any(path)
    try:
        for key in ("state", "state_dict"):
            if key in sd:
                sd = sd[key]
                break
        model.load_state_dict(sd, strict=True)
    except (RuntimeError, KeyError, TypeError):       # key / shape mismatch: the reference's order-based fallback (load_model.py:92-100)
        try:
            model.load_pretrain(path)
        except (AssertionError, RuntimeError, KeyError):
            raise SystemExit("Error: mismatch models weights. Please check if the model configurations match to model weights!")
    return model
