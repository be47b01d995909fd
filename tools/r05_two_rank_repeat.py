"""run the two-rank full-size bucketed replay (tests/ddp_worker.py bench2) a few times; report non-finite parameters / losses per run"""
import os, subprocess, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for it in range(n):
    d = tempfile.mkdtemp()
    port = str(29600 + it)
    outs = [os.path.join(d, f"rank{r}.pt") for r in range(2)]
    env = dict(os.environ, PYTHONPATH=ROOT)
    ps = [subprocess.Popen([sys.executable, "-m", "tests.ddp_worker", str(r), "2", port, outs[r], "bench2"], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in ps]
    if any(p.returncode for p in ps):
        print("run", it, "FAILED\n", "\n----\n".join(logs)[-3000:])
        continue
    rk = [torch.load(o, weights_only=False) for o in outs]
    fp = rk[0]["flat_p"]
    bad = (~torch.isfinite(fp)).nonzero().flatten()
    print("run", it, "losses", rk[0]["losses"], rk[1]["losses"], "non-finite params:", bad.numel(), bad[:10].tolist(),
          "replicas equal:", torch.equal(torch.nan_to_num(fp), torch.nan_to_num(rk[1]["flat_p"])), flush=True)
