"""first step at which two FQSS_DETERMINISTIC=1 runs of the tiny training stream differ, and which gradient tensors differ there"""
import os, sys
os.environ.setdefault("FQSS_DETERMINISTIC", "1")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd.data import synth_batch_2band
from fqss_amd.runtime import KDTrainStep
from tests.test_gpu_model import _tiny_pair
g0 = np.load("tests/golden/tiny_step.npz")
gl = np.load("tests/golden/tiny_train_long.npz")
n, B, T, seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(gl["batch"]), int(gl["samples"]), int(gl["seed0"])
graph = "nograph" not in sys.argv
ahead = "noahead" not in sys.argv
runs = []
for r in range(2):
    model, fmodel = _tiny_pair(g0)
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, teacher_ahead=ahead)
    names = {id(p): k for k, p in model.named_parameters()}
    rec = []
    nxt = synth_batch_2band(B, T, seed0, "cuda")
    for i in range(n):
        x, tgt = nxt
        nxt = synth_batch_2band(B, T, seed0 + i + 1, "cuda")
        if graph:
            step.maybe_capture(x, tgt)
        res = step(x, tgt, x_next=nxt[0] if ahead else None)
        rec.append((res["loss"].item(), res["est"].clone(), step.arena.flat_g.clone(), step.arena.flat_p.clone()))
    runs.append((rec, step, names))
for i in range(n):
    a, b = runs[0][0][i], runs[1][0][i]
    same = [a[0] == b[0], torch.equal(a[1], b[1]), torch.equal(a[2], b[2]), torch.equal(a[3], b[3])]
    if not all(same):
        print("first difference at step", i, "(loss, est, grads, params equal:", same, ") graphs:", runs[0][1]._graphs is not None)
        step, names = runs[0][1], runs[0][2]
        for p, o in zip(step.arena.params, step.arena.offsets):
            ga, gb = a[2][o:o + p.numel()], b[2][o:o + p.numel()]
            if not torch.equal(ga, gb):
                print("   grad differs:", names[id(p)], float((ga - gb).abs().max()), float(ga.abs().max()))
        break
else:
    print("no difference over", n, "steps")
