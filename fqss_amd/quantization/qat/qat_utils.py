"""Graph rewrite: replace float modules of a model by their LayerQ counterparts, addressed by
attribute path.  Same entry points as the reference's quantization/qat/qat_utils.py
(quantize_modules :301-310, replace_encoderq/decoderq :312-332, OP_LIST_TO_QUANTIZE_METHOD :354-401);
a fused tuple such as (Conv1d, PReLU) becomes ONE LayerQ at the first path and nn.Identity at the rest
(:273-287)."""
import copy

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
