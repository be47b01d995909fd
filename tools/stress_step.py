"""The student's forward next to the teacher's stream: in the quantizing phase the forward is deterministic, so the output of a step run with
the teacher on its second stream must equal, bit for bit, the output of the same step with the teacher on the student's stream.
    python tools/stress_step.py [cfg2|cfg3|cfg4|cfg5] [rounds]          -> mismatching rounds (0 is the only acceptable answer)"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(which, dev, hd_batch=2, hd_seconds=4.0):
    """hd_batch / hd_seconds: the cfg-5 shape (the stress gates use 2 x 4 s; bench.py's is 4 x 10 s: tools/bench_buckets.py passes it)"""
    from fqss_amd import runtime as R
    from fqss_amd.data import synth_batch
    if which == "cfg2":
        from fqss_amd.smoke import build_pair
        model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
        x, tgt = synth_batch(8, 32000, seed=0, device=dev)
        return R.KDTrainStep(model, fmodel, lr=0.0), x, tgt
    if which in ("cfg3", "cfg4"):
        import bench
        from fqss_amd.quantization.qat.models.load_model import create_model, quantize_model
        from fqss_amd.smoke import QCFG
        W = bench.DUALPATH[which]
        torch.manual_seed(0)
        model = create_model(dict(W["cfg"]))
        fmodel = copy.deepcopy(model).to(dev).eval()
        model = quantize_model(model, dict(QCFG)).to(dev).train()
        x, tgt = synth_batch(1, W["T"], seed=100, device=dev)
        return R.KDTrainStep(model, fmodel, lr=0.0), x, tgt
    from fqss_amd.quantization.qat.models.htdemucsq import HTDemucsQ
    from fqss_amd.quantization.qat.models.load_model import quantize_model
    torch.manual_seed(0)
    B, T = hd_batch, int(round(44100 * hd_seconds))
    model = HTDemucsQ(sources=["drums", "bass", "other", "vocals"], bottom_channels=512, segment=hd_seconds)
    fmodel = copy.deepcopy(model).to(dev).eval()
    qcfg = dict(qat=True, gradient_based=True, weight_quant=True, weight_n_bits=8, act_quant=True, act_n_bits=8, in_quant=False,
                in_act_n_bits=8, out_quant=True, out_act_n_bits=8, n_splitter=2, n_combiner=2, observer=True)
    model = quantize_model(model, qcfg).to(dev).train()
    g = torch.Generator().manual_seed(42)
    src = (torch.randn(B, 4, 2, T, generator=g) * 0.1).to(dev)
    return R.KDTrainStep(model, fmodel, lr=0.0, clip=0.0, loss="l1_sdr"), src.sum(1), src


def run(which="cfg2", rounds=10, bwd=None):
    """-> number of rounds whose output / loss differ from the one-stream step; with bwd (or BWD=1 in the environment) -> (that number,
    gradient noise floor quiet-vs-quiet, worst gradient deviation of a step shadowed by a busy second stream), both relative to max|g|"""
    from fqss_amd import runtime as R
    saved = R.TEACHER_STREAM
    try:
        return _run(which, rounds, R, bool(os.environ.get("BWD")) if bwd is None else bwd)
    finally:
        R.TEACHER_STREAM = saved


def _run(which, rounds, R, bwd):
    from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
    dev = torch.device("cuda", 0)
    step, x, tgt = build(which, dev)
    step.use_graph = False
    step(x, tgt)
    for m in step.model.modules():
        if isinstance(m, GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
    step(x, tgt)
    step(x, tgt)
    R.TEACHER_STREAM = False
    r = step(x, tgt)
    est0, loss0 = r["est"].clone(), float(r["loss"])
    r = step(x, tgt)
    assert torch.equal(r["est"], est0), "the forward is not deterministic even on one stream"
    R.TEACHER_STREAM = True
    bad = 0
    for i in range(rounds):
        r = step(x, tgt)
        torch.cuda.synchronize()
        ok = torch.equal(r["est"], est0) and abs(float(r["loss"]) - loss0) <= 1e-6 * abs(loss0)
        bad += 0 if ok else 1
    print({"workload": which, "rounds": rounds, "mismatching": bad})
    if not bwd:
        return bad
    if True:
        # the BACKWARD next to a busy second stream (at world > 1 the gradient all-reduce runs beside it): gradients of a step whose
        # whole duration is shadowed by teacher passes on another stream vs the quiet step; fp32 atomics set the noise floor
        R.TEACHER_STREAM = False
        g0 = None
        floor = 0.0
        for i in range(3):
            step(x, tgt)
            torch.cuda.synchronize()
            g = step.arena.flat_g.clone()
            if g0 is None:
                g0 = g
            else:
                floor = max(floor, float((g - g0).abs().max() / g0.abs().max()))
        side = torch.cuda.Stream()
        worst = 0.0
        for i in range(rounds):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(6):
                    step.teacher(x)
            step(x, tgt)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            dev_ = float((step.arena.flat_g - g0).abs().max() / g0.abs().max())
            if dev_ > max(20 * floor, 2e-6):          # which parameters moved, and by how much of their own scale
                names = {id(p): n for n, p in step.model.named_parameters()}
                rows = []
                for p_, off in zip(step.arena.params, step.arena.offsets):
                    a, b = step.arena.flat_g[off:off + p_.numel()], g0[off:off + p_.numel()]
                    d = float((a - b).abs().max())
                    if d > 20 * floor * float(g0.abs().max()):
                        rows.append((d / (float(b.abs().max()) + 1e-30), names.get(id(p_), "?"), int(((a - b).abs() > 0).sum()), p_.numel()))
                rows.sort(reverse=True)
                print("round", i, "deviation", dev_, [(round(r, 6), n, k, m) for r, n, k, m in rows[:8]])
            worst = max(worst, dev_)
        print({"workload": which, "gradient noise floor (quiet vs quiet)": floor, "worst next to a busy stream": worst})
    return bad, floor, worst


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "cfg2", int(sys.argv[2]) if len(sys.argv) > 2 else 10)
