"""Diagnostic (GPU box): tiny ConvTasNetQ, HIP path vs the oracle stepped side by side.
Prints per-step loss deviation and per-parameter gradient error at chosen steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle.fqss_oracle as O
from fqss_amd.smoke import build_pair
from fqss_amd.runtime import KDTrainStep

torch.set_num_threads(8)
g = np.load("tests/golden/tiny_step.npz")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
TINY = dict(n_spks=2, kernel_size=16, stride=8, n_filters=32, bn_chan=16, hid_chan=32, n_blocks=2, n_repeats=1)
model, fmodel = build_pair("cuda", 0, **TINY)
sd0 = {k[4:]: T(g[k]) for k in g.files if k.startswith("sd0.")}
fsd = {k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}
model.load_state_dict(sd0); fmodel.load_state_dict(fsd)
step = KDTrainStep(model, fmodel)
x, tgt = T(g["x"]), T(g["tgt"])
tr = O.Trainer(O.StudentConvTasNetQ(sd0, layers_per_stack=2), O.TeacherConvTasNet(fsd, layers_per_stack=2))
resync = len(sys.argv) > 1 and sys.argv[1] == "resync"
names = [n for n, p in model.named_parameters()]
for s in range(1, 54):
    r = step(x.cuda(), tgt.cuda())
    ro = tr.step(x, tgt)
    # gradient comparison (flat_g still holds this step's raw grads)
    worst = []
    for n, p in model.named_parameters():
        go = tr.s.p[n].grad
        gh = p.grad.detach().cpu()
        if go is None:
            if gh.abs().max() > 0: worst.append((float("inf"), n, float(gh.abs().max()), 0.0))
            continue
        # oracle grads were clipped in place by clip_grad_norm_: undo
        coef = min(1.0, 5.0 / (float(ro["gnorm"]) + 1e-6))
        go = go / coef
        den = float(go.norm()) + 1e-12
        worst.append((float((gh - go).norm()) / den, n, float(gh.norm()), den))
    worst.sort(reverse=True)
    dl = abs(r["loss"].item() - ro["loss"].item())
    if s <= 4 or s % 10 == 0 or s >= 50:
        print(f"step {s}: loss hip {r['loss'].item():.6f} oracle {ro['loss'].item():.6f} |d|={dl:.2e} gnorm {r['gnorm'].item():.5f}/{float(ro['gnorm']):.5f}")
        for e, n, a, b in worst[:5]:
            print(f"      grad relerr {e:.2e}  {n}  |hip|={a:.3e} |ref|={b:.3e}")
    if resync:   # teacher-force the parameters: copy oracle params into the HIP model after each step
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.data.copy_(tr.s.p[n].detach())
            step.arena.exp_avg.zero_(); step.arena.exp_avg_sq.zero_()
