"""smoke(): one tiny QAT step of ConvTasNetQ on cuda:0 through the HIP path, checked against the
oracle (oracle/ is test infrastructure: imported here only as the checker)."""
import copy

import torch

QCFG = {"qat": True, "gradient_based": True, "weight_quant": True, "weight_n_bits": 8, "act_quant": True,
        "act_n_bits": 8, "in_quant": False, "in_act_n_bits": 8, "out_quant": True, "out_act_n_bits": 8,
        "n_splitter": 2, "n_combiner": 2, "observer": True}


def build_pair(device="cuda", seed=0, **kw):
    """(student, teacher) like train_utils.create_pretrained_model, from random init"""
    from .quantization.qat.models.convtasnetq import ConvTasNetQ
    from .quantization.qat.models.load_model import quantize_model
    torch.manual_seed(seed)
    model = ConvTasNetQ(**kw)
    fmodel = copy.deepcopy(model)
    model = quantize_model(model, dict(QCFG))
    return model.to(device).train(), fmodel.to(device).eval()


def run_smoke():
    from oracle import fqss_oracle as O
    from .runtime import KDTrainStep
    kw = dict(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)
    model, fmodel = build_pair("cuda", 0, **kw)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    fsd = {k: v.detach().cpu().clone() for k, v in fmodel.state_dict().items()}
    x, tgt = O.synth_batch(2, 800, seed=0)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0)
    r = step(x.cuda(), tgt.cuda())
    torch.cuda.synchronize()
    ref = O.Trainer(O.StudentConvTasNetQ(sd0, layers_per_stack=2), O.TeacherConvTasNet(fsd, layers_per_stack=2)).step(x, tgt)
    got, want = r["loss"].item(), ref["loss"].item()
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    err = (r["est"].cpu() - ref["est"].detach()).abs().max().item()
    assert err < 1e-5, err
    print(f"smoke ok: loss {got:.6f} (oracle {want:.6f}), max|est - oracle| {err:.2e}")
