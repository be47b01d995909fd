import sys, torch, numpy as np
sys.path.insert(0, ".")
from tests.helpers_cfg1 import cfg1_fill
from fqss_amd.data import synth_batch
from fqss_amd.runtime import KDTrainStep
from fqss_amd.smoke import build_pair
from fqss_amd import ops
g = np.load("tests/golden/cfg1_step.npz")
model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
cfg1_fill(fmodel, "T."); cfg1_fill(model, "S.")
x, tgt = synth_batch(2, 8000, seed=0, device="cuda")
cap = {}
mods = dict(model.named_modules())
for name in ("masker.TCN.5.shared_block.3", "masker.TCN.14.shared_block.0", "masker.TCN.9.shared_block.3"):
    def fh(mod, inp, out, name=name):
        cap[name + ".y"] = ops.real(out).detach().clone()
        out.register_hook(lambda gr, name=name: cap.__setitem__(name + ".g", gr.detach().clone()))
    mods[name].register_forward_hook(fh)
step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0)
a = step.arena
a.zero_grad()
est = model(x)
fest = step.teacher(x)
from fqss_amd import kernels as K
out, w, sisdr, gest = K.kd_loss(est.detach(), fest, tgt, 0.1, want_grad=True)
est.backward(gest)
names = list(g["param_names"]); ref = g["s1.grad_norm"]; coef = min(1.0, 5.0 / (float(g["s1.gnorm"]) + 1e-6))
for name in ("masker.TCN.5.shared_block.3", "masker.TCN.14.shared_block.0", "masker.TCN.9.shared_block.3"):
    y, gr = cap[name + ".y"].double(), cap[name + ".g"].double()
    slope = float(mods[name].nl.weight)
    z = torch.where(y > 0, y, y / slope)
    S = float((torch.where(y > 0, torch.zeros_like(z), z) * gr).sum())
    Sabs = float((torch.where(y > 0, torch.zeros_like(z), z) * gr).abs().sum())
    got = float(mods[name].nl.weight.grad)
    print(name, "kernel grad", got, "fp64 recompute", S, "sum|terms|", Sabs, "reference", ref[names.index(name + ".nl.weight")] / coef)
