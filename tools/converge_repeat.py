"""Run-to-run spread of the HIP convergence streams of tests/test_gpu_converge.py (the split-K atomics make a run a sample): tail means
(last 50 steps) of N runs per family beside the reference's configurations.  python tools/converge_repeat.py [runs]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd.runtime import KDTrainStep  # noqa: E402
from tests.test_gpu_converge import _run_stream  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def fam(name):
    if name == "convtasnet":
        from tests.test_gpu_model import _tiny_pair
        return np.load(f"{G}/tiny_step.npz"), np.load(f"{G}/tiny_train_long.npz"), _tiny_pair, dict(lr=1e-3, clip=5.0)
    if name == "dptnet":
        from tests.test_gpu_dptnet import _tiny_pair
        return np.load(f"{G}/dpt_tiny_step.npz"), np.load(f"{G}/dpt_train_long.npz"), _tiny_pair, dict(lr=4e-4, clip=5.0)
    from tests.test_gpu_sepformer import _tiny_pair
    return np.load(f"{G}/sep_tiny_step.npz"), np.load(f"{G}/sep_train_long.npz"), _tiny_pair, dict(lr=1.5e-4, clip=5.0, loss="sisdr_pit_per_sample")


for name in ("convtasnet", "dptnet", "sepformer"):
    g0, gl, make, kw = fam(name)
    n, B, T, seed0 = int(gl["n_steps"]), int(gl["batch"]), int(gl["samples"]), int(gl["seed0"])
    ref = gl["sisdr"][:, -50:].mean(1)
    tails, early = [], []
    for r in range(runs):
        model, fmodel = make(g0)
        step = KDTrainStep(model, fmodel, kd_lambda=0.1, teacher_ahead=True, **kw)
        sisdr, loss = _run_stream(step, n, B, T, seed0)
        tails.append(float(sisdr[-50:].mean()))
        early.append(float(np.abs(sisdr[:50] - gl["sisdr"][:, :50].mean(0)).max()))
    print(f"{name:11s} reference tails {np.round(ref, 3)} (mean {ref.mean():.3f}, max - min {ref.max() - ref.min():.3f});  hip tails {np.round(tails, 3)} "
          f"(mean {np.mean(tails):.3f}, max - min {np.max(tails) - np.min(tails):.3f});  observer-phase deviation {np.round(early, 3)}", flush=True)
