"""fqss_wq_multi_fwd / _bwd of the real ConvTasNetQ tables, timed alone (graph replays): python tools/bench_tables.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K
from fqss_amd.data import synth_batch
from fqss_amd.runtime import KDTrainStep
from fqss_amd.smoke import build_pair
from fqss_amd.quantization.qat.qat_quant import GradientActivationFakeQuantize
dev = "cuda"
x, tgt = synth_batch(2, 8000, seed=0, device=dev)
model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
step = KDTrainStep(model, fmodel, lr=0.0)
step.use_graph = False
step(x, tgt)
for m in model.modules():
    if isinstance(m, GradientActivationFakeQuantize):
        m.n_iter = m.max_observations
step(x, tgt); step(x, tgt); step(x, tgt)
t = step.tables
assert t is not None
print("descriptors", t.wq_table.shape[0], "channels", t.total_channels)
def timeit(fn, iters=50):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
print("wq_multi_fwd %.1f us" % timeit(lambda: K.wq_multi_fwd(t.wq_table, t.total_channels)))
print("wq_multi_bwd %.1f us" % timeit(lambda: K.wq_multi_bwd(t.wq_table, t.total_channels)))
