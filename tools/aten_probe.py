"""which ATen copies / fills / adds on large tensors does one HTDemucs (cfg 5) quantizing step still issue, and from where?  The tensor
methods that launch them are wrapped and the innermost fqss_amd frames recorded.  Dev tool (GPU box): python tools/aten_probe.py"""
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
    # Python-level call sites of the big copies: wrap the tensor methods that launch them
    import traceback
    sites = collections.Counter()

    def wrap(owner, name):
        real = getattr(owner, name)

        def f(*a, **k):
            t = a[0] if a and isinstance(a[0], torch.Tensor) else None
            if name == "cat":
                t = a[0][0]
            if t is not None and t.is_cuda and t.numel() >= (1 << 20) and not (name == "contiguous" and t.is_contiguous()):
                fr = [x for x in traceback.extract_stack()[:-1] if "fqss_amd" in x.filename]
                where = " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-3:])
                sites[(name, tuple(t.shape), where)] += 1
            return real(*a, **k)
        setattr(owner, name, f)
    for nm in ("contiguous", "clone", "copy_", "zero_", "fill_", "add_", "add", "mul", "sub", "__add__", "__mul__", "__sub__", "__iadd__"):
        wrap(torch.Tensor, nm)
    for nm in ("cat", "zeros_like", "zeros", "stack"):
        wrap(torch, nm)
    import torch.nn.functional as F
    wrap(F, "pad")
    step(mix, src)
    torch.cuda.synchronize()
    for k, v in sites.most_common(70):
        print(v, k)


if __name__ == "__main__":
    main()
