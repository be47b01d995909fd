"""Single-GPU cost of the data-parallel step's schedule (no exchange): cfg-2 step time with 1 vs 4 backward segments (one hipGraph each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fqss_amd.data import synth_batch
from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
from fqss_amd.runtime import KDTrainStep
from fqss_amd.smoke import build_pair
x, tgt = synth_batch(8, 32000, seed=100, device="cuda")
for nb in (1, 4, 1, 4):
    model, fmodel = build_pair("cuda", 0, n_spks=2, kernel_size=16, stride=8)
    step = KDTrainStep(model, fmodel, buckets=nb)
    step(x, tgt)
    for m in model.modules():
        if isinstance(m, GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
    step(x, tgt); step(x, tgt)
    step.capture(x, tgt)
    for _ in range(3):
        step(x, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step(x, tgt)
    torch.cuda.synchronize()
    print(f"buckets {nb}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step ({len(step._graphs[0])} backward graphs)", flush=True)
    del step, model, fmodel
    torch.cuda.empty_cache()
