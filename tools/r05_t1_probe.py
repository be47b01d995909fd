"""isolated timing of the teacher's T1 GEMM (k_tgemm_k128 / k_tgemm2<0>) at the cfg-2 shape: 96 launches in a hipGraph, operands rotating
    python tools/r05_t1_probe.py            (FQSS_T1_K128=0: the round-4 kernel; FQSS_T1_STAGGER=<cycles>: the start-late experiment)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fqss_amd import kernels as K
dev = torch.device("cuda")
B, Ci, Co, M = 8, 128, 512, 3999
w = torch.randn(Co, Ci, device=dev) * 0.1
planes = K.split3_planes(w)       # (carries the tiled image for 256-row shapes: kernels.tgemm picks the tiled entry point)
bias, slope = torch.randn(Co, device=dev), torch.tensor([0.25], device=dev)
xs = []
for _ in range(8):
    x = K.empty_act((B, Ci, M), dev); x.copy_(torch.randn(B, Ci, M, device=dev)); xs.append(x)
st = K.tstat_buffer(8, B, dev)
def run(i):
    return K.tgemm(planes, xs[i % 8], bias, act=1, slope=slope, stats_out=st[i % 8])
for i in range(3): run(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(96): run(i)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"T1 {e0.elapsed_time(e1) / (5 * 96) * 1e3:.2f} us/launch  (FQSS_T1_K128={os.environ.get('FQSS_T1_K128', '1')}, FQSS_T1_STAGGER={os.environ.get('FQSS_T1_STAGGER', '0')})")
