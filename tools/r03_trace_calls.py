"""which C-ABI entry points does the un-fused ConvTasNet QAT step call (observer + quantizing phase, module-path teacher)?"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fqss_amd import _lib
from fqss_amd.runtime import KDTrainStep
from tests.test_gpu_model import T, _tiny_pair
names = collections.Counter()
real = _lib.call
def call(name, *a):
    names[name] += 1
    return real(name, *a)
_lib.call = call
g = np.load("tests/golden/tiny_step.npz")
model, fmodel = _tiny_pair(g)
step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, coded=False, batched_quantizers=False)
step.teacher.ok = False
x, tgt = T(g["x"]).cuda(), T(g["tgt"]).cuda()
for i in range(53):
    step(x, tgt)
torch.cuda.synchronize()
print(len(names), "entry points")
for k, v in sorted(names.items()):
    print(f"  {k:28s} {v}")
