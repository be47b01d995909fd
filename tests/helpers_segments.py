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
