"""Diagnostic (GPU box): step 51 of the tiny ConvTasNetQ from the reference's state, on the benchmarked KDTrainStep path
(fast codes + deferred tables; eager and hipGraph replay) and on the un-fused fp32 path, against tests/golden/tiny_step.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqss_amd import ops                                   # noqa: E402
from fqss_amd.runtime import KDTrainStep                   # noqa: E402
from tests.test_gpu_model import T, _leave_observer   # noqa: E402

FAMILY = sys.argv[1] if len(sys.argv) > 1 else "convtasnet"
if FAMILY == "convtasnet":
    from tests.test_gpu_model import _tiny_pair            # noqa: E402
    fixture = "tiny_step"
elif FAMILY == "dptnet":
    from tests.test_gpu_dptnet import _tiny_pair           # noqa: E402
    fixture = "dpt_tiny_step"
else:
    from tests.test_gpu_sepformer import _tiny_pair        # noqa: E402
    fixture = "sep_tiny_step"
g = np.load(os.path.join(ROOT, "tests/golden", fixture + ".npz"))
x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
coef = min(1.0, 5.0 / (float(g["s51.gnorm"]) + 1e-6))


def run(kind):
    model, fmodel = _tiny_pair(g, prefix="s50.post_sd.")
    _leave_observer(model)
    if kind == "unfused":
        step = KDTrainStep(model, fmodel, batched_quantizers=False, coded=False, lr=0.0)
    else:
        step = KDTrainStep(model, fmodel, lr=0.0)
    with ops.poison_carriers(True):
        r = step(x, tgt)
        if kind == "replay":
            step.capture(x, tgt, warmup=0)
            r = step(x, tgt)
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}
    return r, grads


res = {}
for kind in ("unfused", "eager", "replay"):
    r, grads = run(kind)
    res[kind] = (r, grads)
    est = r["est"].cpu().numpy()
    print(f"== {kind}: loss {r['loss'].item():.5f} (ref {float(g['s51.loss']):.5f})  kd {r['kd'].item():.6f} ({float(g['s51.kd']):.6f}) "
          f"gnorm {r['gnorm'].item():.5f} ({float(g['s51.gnorm']):.5f})  est rms err {np.sqrt(np.mean((est - g['s51.est']) ** 2)) / np.sqrt(np.mean(g['s51.est'] ** 2)):.3e}")
    errs = []
    for n, gv in grads.items():
        k = "s51.grad." + n
        if k in g.files:
            ref = g[k] / coef
            errs.append((float(np.linalg.norm(gv - ref) / (np.linalg.norm(ref) + 1e-12)), n, float(np.linalg.norm(ref))))
        else:
            assert np.abs(gv).max() == 0, n
    errs.sort(reverse=True)
    e = np.array([a for a, _, _ in errs])
    print(f"   grads vs golden: n {len(e)} max {e.max():.3e} median {np.median(e):.3e} p90 {np.percentile(e, 90):.3e}")
    for a, n, nr in errs[:8]:
        print(f"      {a:.3e}  |ref| {nr:.3e}  {n}")
for a, b in (("eager", "unfused"), ("replay", "eager")):
    ga, gb = res[a][1], res[b][1]
    e = [(float(np.linalg.norm(ga[n] - gb[n]) / (np.linalg.norm(gb[n]) + 1e-12)), n) for n in ga if np.abs(gb[n]).max() > 0]
    e.sort(reverse=True)
    print(f"== {a} vs {b}: loss {res[a][0]['loss'].item():.6f} / {res[b][0]['loss'].item():.6f}; grads max {e[0][0]:.3e} median {np.median([v for v, _ in e]):.3e}; worst {e[:4]}")
