"""Shared check of the gradient-bucket schedule (runtime.KDTrainStep(buckets=n)): a model's backward run as n separately
launched segments must give the loss and every gradient of the one-pass backward, eagerly and as one hipGraph per segment."""
import torch


def check_backward_segments(build, x, tgt, nb, nseg, step_kw=None, tol=2e-4):
    from fqss_amd import ops
    from fqss_amd.quantization.qat import qat_quant as QQ
    from fqss_amd.runtime import KDTrainStep
    out = {}
    for n in (1, nb):
        model, fmodel = build()
        step = KDTrainStep(model, fmodel, lr=0.0, buckets=n, **(step_kw or {}))
        assert (step.segments is not None) == (n > 1)
        with ops.poison_carriers(True):
            step(x, tgt)
            for m in model.modules():
                if isinstance(m, QQ.GradientActivationFakeQuantize):
                    m.n_iter = m.max_observations
            step(x, tgt)
            r = step(x, tgt)
            res = [(r["loss"].item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})]
            if n > 1:
                assert len(step.segments) == nseg, step.segments
                step.capture(x, tgt, warmup=0)
                assert len(step._graphs[0]) == nseg
                step.arena.flat_g.fill_(float("nan"))
                r = step(x, tgt)
                res.append((r["loss"].item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
        out[n] = res
        del step, model, fmodel
        torch.cuda.empty_cache()
    (ref_loss, ref_g), = out[1]
    for loss, g in out[nb]:
        assert abs(loss - ref_loss) <= 1e-6 * max(abs(ref_loss), 1e-3)
        for k, v in ref_g.items():
            nv = float(v.norm())
            assert torch.isfinite(g[k]).all(), k
            if nv > 1e-9:
                assert float((g[k] - v).norm()) <= tol * nv, (k, float((g[k] - v).norm()) / nv)


def check_batched_tables(build, x, tgt, n_weights, step_kw=None, tol=1e-3):
    """runtime.QuantTables (every weight fake-quant of the model in one launch each way, dL/dW_q through the step's arena) against the
    per-layer quantizers, same state, one quantizing step: same loss, same output, every gradient"""
    from fqss_amd.quantization.qat import qat_quant as QQ
    from fqss_amd.runtime import KDTrainStep
    out = {}
    for batched in (False, True):
        model, fmodel = build()
        with torch.no_grad():
            model(x)
        for m in model.modules():
            if isinstance(m, QQ.GradientActivationFakeQuantize):
                m.n_iter = m.max_observations
        step = KDTrainStep(model, fmodel, lr=0.0, batched_quantizers=batched, **(step_kw or {}))
        r = step(x, tgt)
        out[batched] = (r["loss"].item(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}, r["est"].clone())
        if batched:
            nq = sum(isinstance(m, QQ.GradientWeightFakeQuantize) for m in model.modules())
            # (a quantizer outside the tables -- the ch_out_idx = 1 one of the trainable residual decoder -- keeps its own launches)
            assert len(step.tables.weights) == n_weights and 0 <= nq - n_weights <= 1, (len(step.tables.weights), nq)
        del step, model, fmodel
    assert out[True][0] == out[False][0] and torch.equal(out[True][2], out[False][2])
    for k, v in out[False][1].items():
        nv = float(v.norm())
        assert float((out[True][1][k] - v).norm()) <= tol * max(nv, 1e-9), k
