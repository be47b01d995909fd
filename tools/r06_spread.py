"""Run-to-run spread of the full-size convergence streams of tests/test_gpu_converge.py (default mode: the bias / depthwise /
frame-weight gradients still add with fp32 atomics, so a run is a sample): N runs each of the lr 1e-3 and lr 1e-4 streams, the
50-step SI-SDR tails, their standard deviation and the studentized-range quantiles a k-run gate can be held to.
    python tools/r06_spread.py [runs]        (GPU box; ~6 s per run)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd.runtime import KDTrainStep  # noqa: E402
from fqss_amd.smoke import build_pair  # noqa: E402
from tests.helpers_cfg1 import cfg1_fill  # noqa: E402
from tests.test_gpu_converge import _run_stream  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for name, fix, lr in (("full-size convtasnet", "cfg1_train_long", 1e-3), ("full-size convtasnet lr 1e-4", "cfg1_train_long_lr1e-4", 1e-4)):
    gl = np.load(f"{G}/{fix}.npz")
    n, B, T, seed0 = int(gl["n_steps"]), int(gl["batch"]), int(gl["samples"]), int(gl["seed0"])
    ref = gl["sisdr"][:, -50:].mean(1)
    tails = []
    for r in range(runs):
        model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
        cfg1_fill(fmodel, "T.")
        cfg1_fill(model, "S.")
        step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=lr, clip=5.0, teacher_ahead=True)
        sisdr, loss = _run_stream(step, n, B, T, seed0)
        tails.append(float(sisdr[-50:].mean()))
    t = np.array(tails)
    print(f"{name}: reference tails {np.round(ref, 3)} (mean {ref.mean():.3f}, max - min {np.ptp(ref):.3f}, sd {ref.std(ddof=1):.3f}); {runs} HIP tails "
          f"{np.round(t, 3)} (mean {t.mean():.3f}, max - min {np.ptp(t):.3f}, sd {t.std(ddof=1):.3f})", flush=True)
