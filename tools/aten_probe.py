"""which ATen element-wise / copy ops does one HTDemucs (cfg 5) quantizing step still issue, and from where?  torch.profiler on the CPU side
(op name, input shapes, innermost fqss_amd frame).  Dev tool (GPU box): python tools/aten_probe.py"""
import collections
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fqss_amd.quantization.qat.models.load_model import quantize_model
    from fqss_amd.quantization.qat.models.htdemucsq import HTDemucsQ
    from fqss_amd.runtime import KDTrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, T = 4, 441000
    model = HTDemucsQ(sources=["drums", "bass", "other", "vocals"], bottom_channels=512, segment=10.0)
    fmodel = copy.deepcopy(model).to(dev).eval()
    qcfg = dict(qat=True, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, in_quant=False,
                in_act_n_bits=8, out_quant=True, out_act_n_bits=8, n_splitter=2, n_combiner=2, observer=True)
    model = quantize_model(model, qcfg).to(dev).train()
    g = torch.Generator().manual_seed(42)
    src = (torch.randn(B, 4, 2, T, generator=g) * 0.1).to(dev)
    mix = src.sum(1)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=3e-4, clip=0.0, loss="l1_sdr")
    step.use_graph = False
    step(mix, src)
    with torch.no_grad():
        for _ in range(49):
            model(mix)
    step(mix, src)
    step(mix, src)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        step(mix, src)
        torch.cuda.synchronize()
    seen = collections.Counter()
    numel = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::reshape",
                                                           "aten::as_strided", "aten::slice", "aten::select", "aten::unsqueeze", "aten::squeeze",
                                                           "aten::permute", "aten::transpose", "aten::t", "aten::expand", "aten::detach", "aten::alias",
                                                           "aten::_unsafe_view", "aten::narrow", "aten::resize_", "aten::result_type", "aten::to",
                                                           "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::lift_fresh", "aten::flatten",
                                                           "aten::unflatten", "aten::chunk", "aten::split", "aten::unbind", "aten::view_as", "aten::contiguous"):
            continue
        frame = next((s for s in ev.stack if "fqss_amd" in s and "autograd/function" not in s), "?")
        shapes = str([s for s in (ev.input_shapes or []) if s][:2])
        key = (ev.name, frame.split("fqss_amd/")[-1][:70], shapes[:60])
        seen[key] += 1
    for k, v in seen.most_common(60):
        print(v, k)


if __name__ == "__main__":
    main()
