"""Full-size ConvTasNet convergence stream (tests/test_gpu_converge.py) under HIP rounding variants, window means of SI-SDR beside the
reference's six CPU configurations (tests/golden/cfg1_train_long.npz).  python tools/converge_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K  # noqa: E402
from fqss_amd.runtime import KDTrainStep  # noqa: E402
from fqss_amd.smoke import build_pair  # noqa: E402
from tests.helpers_cfg1 import cfg1_fill  # noqa: E402
from tests.test_gpu_converge import _run_stream  # noqa: E402

gl = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg1_train_long.npz"))
n, B, T, seed0 = int(gl["n_steps"]), int(gl["batch"]), int(gl["samples"]), int(gl["seed0"])
wins = [(0, 50), (50, 100), (100, 150), (150, 200), (200, 250), (250, 300)]


def show(name, s):
    print(f"{name:34s}" + "".join(f"  {float(s[a:b].mean()):7.3f}" for a, b in wins), flush=True)


print(" " * 34 + "".join(f"  {a:3d}-{b:3d}" for a, b in wins))
for v in range(gl["sisdr"].shape[0]):
    show(f"reference {gl['variants'][v]}", gl["sisdr"][v])
for name, tiled, ahead in (("hip tiled teacher GEMM, ahead", True, True), ("hip 128-row teacher GEMM, ahead", False, True),
                           ("hip tiled teacher GEMM, in step", True, False)):
    K.TGEMM_TILED = tiled
    model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
    cfg1_fill(fmodel, "T.")
    cfg1_fill(model, "S.")
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, teacher_ahead=ahead)
    sisdr, loss = _run_stream(step, n, B, T, seed0)
    show(name, sisdr)
    np.save(f"gpurun_out/converge_{'t' if tiled else 'v1'}_{'a' if ahead else 'i'}.npy", np.stack([sisdr, loss]))
