"""Graph rewrite: replace float modules of a model by their LayerQ counterparts, addressed by
attribute path.  Same entry points as the reference's quantization/qat/qat_utils.py
(quantize_modules :301-310, replace_encoderq/decoderq :312-332, OP_LIST_TO_QUANTIZE_METHOD :354-401);
a fused tuple such as (Conv1d, PReLU) becomes ONE LayerQ at the first path and nn.Identity at the rest
(:273-287)."""
import copy

import torch
import torch.nn as nn

from . import qat_layers as QL
from .qat_layers import Add, Const, Div, Mul, Sub


def _get_module(model, path):
    m = model
    for tok in path.split("."):
        m = getattr(m, tok)
    return m


def _set_module(model, path, module):
    toks = path.split(".")
    m = model
    for tok in toks[:-1]:
        m = getattr(m, tok)
    setattr(m, toks[-1], module)


def _pick(params, *names):
    return {n: params.get(n) for n in names}


_WA = ("gradient_based", "act_quant", "weight_quant", "act_n_bits", "weight_n_bits")
_A = ("gradient_based", "act_quant", "act_n_bits")


def _wa(cls):
    return lambda *mods_and_params: cls(*mods_and_params[:-1], **_pick(mods_and_params[-1], *_WA))


def _a(cls):
    return lambda *mods_and_params: cls(*mods_and_params[:-1], **_pick(mods_and_params[-1], *_A))


quant_conv1d, quant_conv2d = _wa(QL.Conv1dQ), _wa(QL.Conv2dQ)
quant_convtr1d, quant_convtr2d = _wa(QL.ConvTranspose1dQ), _wa(QL.ConvTranspose2dQ)
quant_conv1d_nl, quant_conv2d_nl = _wa(QL.Conv1dNlQ), _wa(QL.Conv2dNlQ)
quant_conv1d_gn_nl = _wa(QL.Conv1dGnNlQ)
quant_convtr1d_nl, quant_convtr2d_nl = _wa(QL.ConvTranspose1dNlQ), _wa(QL.ConvTranspose2dNlQ)
quant_groupnorm, quant_layernorm, quant_batchnorm = _a(QL.GroupNormQ), _a(QL.LayerNormQ), _a(QL.BatchNormQ)
quant_embedding, quant_lstm, quant_mha = _wa(QL.EmbeddingQ), _wa(QL.LSTMQ), _wa(QL.MultiheadAttentionQ)
quant_linear, quant_linear_nl = _wa(QL.LinearQ), _wa(QL.LinearNlQ)
quant_nl, quant_add, quant_sub, quant_mul, quant_div, quant_const = (_a(QL.NlQ), _a(QL.AddQ), _a(QL.SubQ), _a(QL.MulQ),
                                                                    _a(QL.DivQ), _a(QL.ConstQ))


def quant_encoderq(encoder, p):
    if isinstance(encoder[0], nn.Conv1d):
        return QL.Conv1dEncoderQ(encoder, **_pick(p, "n_splitter", "gradient_based", "act_quant", "inout_nl_quant",
                                                  "weight_quant", "in_quant", "act_n_bits", "weight_n_bits", "in_act_n_bits"))
    if isinstance(encoder[0], nn.Conv2d):
        return QL.Conv2dEncoderQ(encoder, **p)
    raise AssertionError("No support!")


def quant_decoderq(decoder, p):
    kw = _pick(p, "n_combiner", "gradient_based", "act_quant", "inout_nl_quant", "weight_quant", "weight_n_bits",
               "act_n_bits", "out_quant", "out_act_n_bits", "train_res_dec")
    if isinstance(decoder[0], nn.ConvTranspose1d):
        return QL.ConvTr1dDecoderQ(decoder, **kw)
    if isinstance(decoder[0], nn.ConvTranspose2d):
        return QL.ConvTr2dDecoderQ(decoder, **kw)
    if isinstance(decoder[0], nn.Linear):
        return QL.LinearDecoderQ(decoder, **kw)
    raise AssertionError("No support!")


_NLS = (nn.PReLU, nn.ReLU, nn.Tanh, nn.Sigmoid, nn.GELU, nn.GLU)
OP_LIST_TO_QUANTIZE_METHOD = {
    nn.Conv1d: quant_conv1d, nn.Conv2d: quant_conv2d,
    nn.ConvTranspose1d: quant_convtr1d, nn.ConvTranspose2d: quant_convtr2d,
    **{(nn.Conv1d, nl): quant_conv1d_nl for nl in _NLS},
    **{(nn.Conv1d, nn.GroupNorm, nl): quant_conv1d_gn_nl for nl in _NLS},
    **{(nn.Conv2d, nl): quant_conv2d_nl for nl in _NLS},
    (nn.ConvTranspose1d, nn.GELU): quant_convtr1d_nl, (nn.ConvTranspose2d, nn.GELU): quant_convtr2d_nl,
    nn.GroupNorm: quant_groupnorm, nn.LayerNorm: quant_layernorm,
    nn.BatchNorm1d: quant_batchnorm, nn.BatchNorm2d: quant_batchnorm, nn.Embedding: quant_embedding,
    **{nl: quant_nl for nl in (nn.PReLU, nn.ReLU, nn.LeakyReLU, nn.Sigmoid, nn.Tanh, nn.GELU, nn.GLU)},
    nn.LSTM: quant_lstm, nn.MultiheadAttention: quant_mha,
    nn.Linear: quant_linear, (nn.Linear, nn.ReLU): quant_linear_nl, (nn.Linear, nn.GELU): quant_linear_nl,
    Add: quant_add, Sub: quant_sub, Mul: quant_mul, Div: quant_div, Const: quant_const,
}


def quantize_known_modules(mod_list, params_dict):
    types = tuple(type(m) for m in mod_list)
    key = types[0] if len(types) == 1 else types
    method = OP_LIST_TO_QUANTIZE_METHOD.get(key)
    if method is None:
        raise NotImplementedError("Cannot quantize modules: {}".format(key))
    fused = [method(*mod_list, params_dict)]
    for _ in mod_list[1:]:
        ident = nn.Identity()
        ident.training = mod_list[0].training
        fused.append(ident)
    return fused


def quantize_modules(model, modules_to_quantize, params_dict={}, inplace=True, replacer_func=quantize_known_modules):
    if not inplace:
        model = copy.deepcopy(model)
    mods = [_get_module(model, p) for p in modules_to_quantize]
    for path, new in zip(modules_to_quantize, replacer_func(mods, params_dict)):
        _set_module(model, path, new)
    return model


def _replace_io(model, paths, params_dict, builder):
    new = builder([_get_module(model, p) for p in paths], params_dict)
    _set_module(model, paths[0], new)
    for p in paths[1:]:
        _set_module(model, p, nn.Identity())


def replace_encoderq(model, modules_to_replace, params_dict):
    _replace_io(model, modules_to_replace, params_dict, quant_encoderq)


def replace_decoderq(model, modules_to_replace, params_dict):
    _replace_io(model, modules_to_replace, params_dict, quant_decoderq)


# true-integer export (reference qat_utils.py:246-255, 334-351): swap a learned quantizer for its affine torch-form wrapper
def torch_weight_quantizer(quantizer):
    from .qat_quant import TorchWeightFakeQuantize
    return TorchWeightFakeQuantize(quantizer)


def torch_activation_quantizer(quantizer):
    from .qat_quant import TorchActivationFakeQuantize
    return TorchActivationFakeQuantize(quantizer)


def torch_dym_activation_quantizer(quantizer):
    from .qat_quant import TorchDymActivationFakeQuantize
    return TorchDymActivationFakeQuantize(quantizer)


def replace_weight_quantizer(model, module_to_replace, module):
    _set_module(model, module_to_replace, torch_weight_quantizer(module))


def replace_activation_quantizer(model, module_to_replace, module):
    _set_module(model, module_to_replace, torch_activation_quantizer(module))


def replace_dym_activation_quantizer(model, module_to_replace, module):
    _set_module(model, module_to_replace, torch_dym_activation_quantizer(module))


# ---------------------------------------------------------------------------------------------------------------------------------
# Integer checkpoint (the reference only sketches the export: qat_utils.py:334-351 swaps quantizers for their
# torch affine twins and keeps fp32 weights).  Here the trained model is STORED as what a W8A8 deployment loads: int8 weight codes +
# per-channel steps on the training grid (delta = 2 max(|min|, |max|) / 255, codes in [-128, 127]: qat_quant.py:126-135), every
# activation quantizer's range (the 8-bit grid delta = (max - min) / 255) next to its torch affine form (scale, zero point), and
# the few float tensors that are not quantized (biases, norms, PReLU slopes).  Loading rebuilds W_q = delta * code -- bit for bit the
# weight the QAT forward multiplies with -- so the eval output of a model restored from the integer file equals the original's.
# ---------------------------------------------------------------------------------------------------------------------------------
INT_CKPT_FORMAT = "fqss-int8-v1"


def weight_quantizer_owners(model):
    """[(weight quantizer, weight parameter, parameter name)] of every GradientWeightFakeQuantize of the model: through the LayerQ that
    owns it (its float submodule's `.weight`), the attention projections and the LSTM matrices"""
    from .qat_quant import GradientWeightFakeQuantize
    names = {id(p): n for n, p in model.named_parameters()}
    out = []

    def fits(wqm, w):
        return isinstance(wqm, GradientWeightFakeQuantize) and isinstance(w, nn.Parameter) and tuple(wqm.min_range.shape) == tuple(
            1 if d != wqm.axis else w.shape[d] for d in range(w.dim()))

    for layer in model.modules():
        wqm = getattr(layer, "weight_fake_quantize", None)
        if isinstance(wqm, GradientWeightFakeQuantize):
            for cand in ("conv1d", "convTr1d", "residual_encoder", "residual_decoder", "linear", "conv2d", "convTr2d"):
                conv = getattr(layer, cand, None)
                if conv is not None and fits(wqm, getattr(conv, "weight", None)):
                    out.append((wqm, conv.weight, names[id(conv.weight)]))
                    break
        if isinstance(layer, QL.MultiheadAttentionQ):
            for wqm, w in ((layer.weight_fake_quantize_in, layer.mha.in_proj_weight), (layer.weight_fake_quantize_out, layer.mha.out_proj.weight)):
                if fits(wqm, w):
                    out.append((wqm, w, names[id(w)]))
        if isinstance(layer, QL.LSTMQ):
            for pname, wqm in layer.weight_quantizers_dict.items():
                if fits(wqm, getattr(layer.lstm, pname, None)):
                    w = getattr(layer.lstm, pname)
                    out.append((wqm, w, names[id(w)]))
    return out


def integer_state(model):
    """the model as integers: {"format", "weights": {parameter name: codes int8, delta [C], axis, min/max range}, "activations":
    {quantizer path: min, max, scale, zero_point}, "float": every other state_dict entry, "keys": the state_dict's key order}"""
    from ... import kernels as K
    from .qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize, TorchActivationFakeQuantize
    sd = model.state_dict()
    weights, covered = {}, set()
    qname = {id(m): n for n, m in model.named_modules()}
    for wqm, w, pname in weight_quantizer_owners(model):
        if wqm.observer_mode:
            raise ValueError(f"{pname}: its weight observer never ran (nothing to export before the first training forward)")
        wq, codes = K.wq_fwd(w.detach(), wqm.axis, wqm.min_range.detach(), wqm.max_range.detach(), want_idx=True)
        a = torch.maximum(wqm.min_range.detach().abs(), wqm.max_range.detach().abs())
        weights[pname] = {"codes": codes.cpu(), "delta": ((2.0 * a) / 255.0).flatten().cpu(), "axis": wqm.axis,
                          "min_range": wqm.min_range.detach().cpu(), "max_range": wqm.max_range.detach().cpu(), "quantizer": qname[id(wqm)]}
        covered |= {pname, qname[id(wqm)] + ".min_range", qname[id(wqm)] + ".max_range"}
    acts = {}
    for name, m in model.named_modules():
        if isinstance(m, GradientActivationFakeQuantize):
            t = TorchActivationFakeQuantize(m)
            acts[name] = {"min_range": m.min_range.detach().cpu(), "max_range": m.max_range.detach().cpu(), "scale": t.scale,
                          "zero_point": t.zero_point, "n_iter": int(m.n_iter)}
            covered |= {name + ".min_range", name + ".max_range"}
    return {"format": INT_CKPT_FORMAT, "weights": weights, "activations": acts,
            "float": {k: v.detach().cpu() for k, v in sd.items() if k not in covered}, "keys": list(sd.keys())}


def save_integer_checkpoint(model, path):
    torch.save(integer_state(model), path)


def load_integer_checkpoint(model, path_or_state):
    """restore a quantized model (same architecture, already through quantize_model) from an integer checkpoint: weights become
    delta * code, ranges are set, observers are switched off (weight observers have run, activation quantizers quantize)"""
    from .qat_quant import GradientActivationFakeQuantize, GradientWeightFakeQuantize
    st = path_or_state if isinstance(path_or_state, dict) else torch.load(path_or_state, weights_only=True)
    if st.get("format") != INT_CKPT_FORMAT:
        raise ValueError(f"not an integer checkpoint of format {INT_CKPT_FORMAT}")
    sd = dict(st["float"])
    for pname, e in st["weights"].items():
        shape = [1] * e["codes"].dim()
        shape[e["axis"]] = -1
        sd[pname] = e["delta"].reshape(shape) * e["codes"].float()          # W_q = delta * code: one exact fp32 product per element
        sd[e["quantizer"] + ".min_range"], sd[e["quantizer"] + ".max_range"] = e["min_range"], e["max_range"]
    for name, e in st["activations"].items():
        sd[name + ".min_range"], sd[name + ".max_range"] = e["min_range"], e["max_range"]
    missing = [k for k in st["keys"] if k not in sd]
    assert not missing, missing
    model.load_state_dict({k: sd[k] for k in st["keys"]}, strict=True)
    by_name = dict(model.named_modules())
    for m in model.modules():
        if isinstance(m, GradientWeightFakeQuantize):
            m.observer_mode = False
    for name, e in st["activations"].items():
        by_name[name].n_iter = e["n_iter"]
    return model
