"""rank r of 2 on ONE GPU (gloo): the cfg-2 step in eager mode with every grouped weight-gradient launch checked against the per-layer kernels"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank = int(sys.argv[1]); world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29711"), FQSS_DIST_BACKEND="gloo")
from fqss_amd import kernels as K
from fqss_amd.data import synth_batch
from fqss_amd.parallel import Comm
from fqss_amd.runtime import KDTrainStep
from fqss_amd.smoke import build_pair
comm = Comm.from_env("cuda")
dev = torch.device("cuda", 0)
nflush = [0]
orig = K.WgradQueue.flush
def checked(self):
    if not self.jobs:
        return
    jobs = list(self.jobs)
    before = [j[5].clone() for j in jobs]
    orig(self)
    torch.cuda.synchronize()
    nflush[0] += 1
    for ji, ((gz1, gz2, xc, lo, hi, gw, ld1, ld2), old) in enumerate(zip(jobs, before)):
        ref = torch.zeros_like(gw)
        if gz2 is None:
            K.qpw_bwd_w(gz1, xc, lo, hi, ref)
        else:
            K.qpw_bwd_w2(gz1, gz2, xc, lo, hi, ref)
        got = gw - old
        err = (got - ref).abs()
        tol = 1e-4 * float(ref.abs().max()) + 1e-12
        if float(err.max()) > tol or not torch.isfinite(got).all():
            bad = (err > tol) | ~torch.isfinite(got)
            rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
            print(f"[rank {rank}] flush {nflush[0]} job {ji}/{len(jobs)} gw {tuple(gw.shape)} M={gz1.shape[2]}: {int(bad.sum())} bad elements, max err {float(err.max()):.3e} "
                  f"(ref max {float(ref.abs().max()):.3e}); rows {rows.min().item()}..{rows.max().item()} cols {cols.min().item()}..{cols.max().item()}; "
                  f"got/ref at worst {float(got.flatten()[err.argmax()]):.4e} / {float(ref.flatten()[err.argmax()]):.4e}", flush=True)
            if float(err.max()) > 1.0:
                g2 = got.reshape(got.shape[0], -1)
                big = (g2.abs() > 1e3) | ~torch.isfinite(g2)
                print("      rows with huge values:", big.any(1).nonzero().flatten().tolist())
                for rr in big.any(1).nonzero().flatten().tolist()[:8]:
                    print("      row", rr, "cols", big[rr].nonzero().flatten().tolist())
                rr = big.any(1).nonzero().flatten().tolist()[0]
                cc = big[rr].nonzero().flatten().tolist()
                print("      values", [f"{float(x):.3e}" for x in g2[rr, cc[0]:cc[0] + 8]], "hex", [hex(int(x)) for x in g2[rr, cc[0]:cc[0] + 8].view(torch.int32) & 0xFFFFFFFF])
K.WgradQueue.flush = checked
model, fmodel = build_pair(dev, 0, n_spks=2, kernel_size=16, stride=8)
x, tgt = synth_batch(8, 32000, seed=100 + rank, device=dev)
step = KDTrainStep(model, fmodel, kd_lambda=0.1, lr=1e-3, clip=5.0, comm=comm, teacher_ahead=True)
step(x, tgt)
with torch.no_grad():
    for _ in range(49):
        model(x)
for it in range(int(os.environ.get("STEPS", "6"))):
    r = step(x, tgt, x_next=x)
    print(f"[rank {rank}] step {it} loss {r['loss'].item():.5f} flushes so far {nflush[0]}", flush=True)
comm.close()
