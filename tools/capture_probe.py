"""Does the replayed cfg-2 step depend on how the step got to its capture?  A: bench.py's calibration (1 step + 49 forwards + 1 eager
quantizing step) + capture(warmup=2); B: the same + capture(warmup=0); C: a trainer's path -- 51 full observer-phase steps over changing
batches, one eager quantizing step, maybe_capture (warmup 0)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd.data import synth_batch  # noqa: E402
from fqss_amd.runtime import KDTrainStep  # noqa: E402
from fqss_amd.smoke import build_pair  # noqa: E402


def timed(step, X, n=40):
    for it in range(6):
        step(*X[it & 1], x_next=X[(it + 1) & 1][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(n):
        step(*X[it & 1], x_next=X[(it + 1) & 1][0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for variant in sys.argv[1:] or ["A", "B", "C"]:
    model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
    X = [synth_batch(8, 32000, seed=100 * (i + 1), device="cuda") for i in range(2)]
    step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, teacher_ahead=True)
    if variant in ("A", "B"):
        step(*X[0])
        with torch.no_grad():
            for _ in range(49):
                model(X[0][0])
        step(*X[0])
        step.capture(*X[0], warmup=2 if variant == "A" else 0)
    else:
        for i in range(60):
            b = synth_batch(8, 32000, seed=i, device="cuda")
            step.maybe_capture(*b)
            if step._graphs is not None:
                print("C: captured before step", i)
                break
            step(*b)
    print(f"variant {variant}: {timed(step, X):.2f} ms per replayed step", flush=True)
    del step, model, fmodel
    torch.cuda.empty_cache()
