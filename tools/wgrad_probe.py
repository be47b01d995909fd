"""Correctness probe of fqss_qpw_bwd_w on one shape: python3 tools/wgrad_probe.py B Ci Co M"""
import sys, torch
sys.path.insert(0, ".")
from fqss_amd import kernels as K
torch.manual_seed(0)
B, Ci, Co, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
gz = K.empty_act((B, Co, M), "cuda"); gz.normal_()
xc = K.empty_codes((B, Ci, M), "cuda"); xc.random_(0, 256)
lo, hi = torch.tensor([0.0], device="cuda"), torch.tensor([255.0], device="cuda")
gw = torch.zeros(Co, Ci, device="cuda")
print("launch", flush=True)
K.qpw_bwd_w(gz, xc, lo, hi, gw)
torch.cuda.synchronize()
ref = torch.einsum("bom,bcm->oc", gz.double(), xc.double())
print("max err", (gw.double() - ref).abs().max().item(), "ref max", ref.abs().max().item(), flush=True)
