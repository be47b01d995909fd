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


def build(workload, dev):
    from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
    from fqss_amd.runtime import KDTrainStep
    qcfg = dict(qat=True, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, in_quant=False,
                in_act_n_bits=8, out_quant=True, out_act_n_bits=8, n_splitter=2, n_combiner=2, observer=True)
    torch.manual_seed(0)
    if workload == "cfg5":
        from fqss_amd.quantization.qat.models.htdemucsq import HTDemucsQ
        B, T = 4, 441000
        model = HTDemucsQ(sources=["drums", "bass", "other", "vocals"], bottom_channels=512, segment=10.0)
        fmodel = copy.deepcopy(model).to(dev).eval()
        model = quantize_model(model, qcfg).to(dev).train()
        g = torch.Generator().manual_seed(42)
        src = (torch.randn(B, 4, 2, T, generator=g) * 0.1).to(dev)
        return model, KDTrainStep(model, fmodel, kd_lambda=0.1, lr=3e-4, clip=0.0, loss="l1_sdr"), src.sum(1), src
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from fqss_amd.data import synth_batch
    W = bench.DUALPATH[workload]
    model = create_model(dict(W["cfg"]))
    fmodel = copy.deepcopy(model).to(dev).eval()
    model = quantize_model(model, qcfg).to(dev).train()
    x, tgt = synth_batch(1, W["T"], seed=100, device=dev)
    return model, KDTrainStep(model, fmodel, kd_lambda=0.1, lr=W["lr"], clip=5.0), x, tgt


def main():
    from fqss_amd import _lib
    args = sys.argv[1:]
    workload = "cfg5"
    if args and args[0] in ("cfg3", "cfg4", "cfg5"):
        workload, args = args[0], args[1:]
    names = args or ["fqss_actq_bwd"]
    dev = torch.device("cuda", 0)
    model, step, mix, src = build(workload, dev)
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
        if os.environ.get("KSITES_ARGS"):      # the small integer arguments (rows, columns, row strides): which launches take an unaligned form
            where += "  " + str(tuple(v for v in a if isinstance(v, int) and 0 <= v < (1 << 24)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(name, *a)
        e1.record()
        recs.append((name, where, e0, e1))
        return r
    _lib.call = spy
    # the ATen elementwise launches (copies, fills, adds) by call site, timed the same way ("aten" among the names)
    if "aten" in names:
        def wrap(owner, name):
            fn = getattr(owner, name)

            def f(*a, **k):
                t = a[0] if a and isinstance(a[0], torch.Tensor) else None
                if t is None or not t.is_cuda or t.numel() < 4096 or (name == "contiguous" and t.is_contiguous()):
                    return fn(*a, **k)
                fr = [x for x in traceback.extract_stack()[:-1] if "fqss_amd" in x.filename]
                where = " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-3:])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a, **k)
                e1.record()
                recs.append((f"aten.{name} {tuple(t.shape)}", where, e0, e1))
                return r
            setattr(owner, name, f)
        for nm in ("contiguous", "clone", "copy_", "zero_", "fill_", "add_", "add", "mul", "sub", "__add__", "__mul__", "__sub__", "__iadd__",
                   "reshape", "index_select", "narrow_copy"):      # (reshape: copies when the view does not exist)
            wrap(torch.Tensor, nm)
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
