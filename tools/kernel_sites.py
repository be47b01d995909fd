"""Which Python call sites issue the launches of selected C-ABI entry points during one HTDemucs (cfg 5) quantizing step, and how much
GPU time each site's launches take (every selected call bracketed by HIP events on its stream: eager step, so the figures include the
launch gaps of an eager step -- use them to RANK sites).   python tools/kernel_sites.py fqss_actq_bwd fqss_axpby ...   (GPU box)"""
import collections
import copy
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from fqss_amd import _lib
    from fqss_amd.quantization.qat.models.load_model import quantize_model
    from fqss_amd.quantization.qat.models.htdemucsq import HTDemucsQ
    from fqss_amd.runtime import KDTrainStep
    names = sys.argv[1:] or ["fqss_actq_bwd"]
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
    real = _lib.call
    recs = []

    def spy(name, *a):
        if not any(name.startswith(n) for n in names):
            return real(name, *a)
        fr = [x for x in traceback.extract_stack()[:-1] if "fqss_amd" in x.filename and "_lib.py" not in x.filename]
        where = " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-4:])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(name, *a)
        e1.record()
        recs.append((name, where, e0, e1))
        return r
    _lib.call = spy
    import fqss_amd.kernels as K
    step(mix, src)
    torch.cuda.synchronize()
    _lib.call = real
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, where, e0, e1 in recs:
        a = agg[(name, where)]
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = sum(v[1] for v in agg.values())
    print(f"{len(recs)} launches of {names}: {tot:.2f} ms (event-bracketed, eager)")
    for (name, where), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{ms:7.3f} ms {n:4d} x  {name:28s} {where}")


if __name__ == "__main__":
    main()
