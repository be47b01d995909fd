#!/usr/bin/env python3
"""per-parameter gradient agreement of the tiny HTDemucs step with the reference fixture (GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_htdemucs import T, _models  # noqa: E402
from fqss_amd import ops  # noqa: E402


def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hd_tiny_step.npz"))
    model, fmodel = _models(g)
    mix, src = T(g["mix"]).cuda(), T(g["src"]).cuda()
    with torch.no_grad():
        fest = fmodel(mix)
    est = model(mix)
    from fqss_amd import kernels as K
    loss, task, kd, w, gest = K.hd_kd_loss(T(g["o1.est"]).cuda(), fest, src, torch.ones(2, device="cuda"), 0.1)   # teacher-forced dL/dest
    est.backward(gest)
    print("observer step 1: loss", loss.item(), float(g["o1.loss"]), "est err", float((est.detach().cpu() - T(g["o1.est"])).abs().max()))
    rows = []
    for k, p in model.named_parameters():
        if "o1.grad." + k in g.files:
            want = g["o1.grad." + k]
            got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
            rows.append((np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30), np.linalg.norm(want), k))
    rows.sort(reverse=True)
    for rel, nrm, k in rows[:40]:
        print(f"O1 {rel:9.5f} |g| {nrm:10.3e}  {k}")
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        for _ in range(49):
            model(mix)
    sd = model.state_dict()
    with torch.no_grad():
        for k in g.files:
            if k.startswith("sd."):
                sd[k[3:]].copy_(T(g[k]))
        fest = fmodel(mix)
    est = model(mix)
    loss, task, kd, w = ops.HdKDLoss.apply(est, fest, src, torch.ones(2, device="cuda"), 0.1)
    loss.backward()
    e, ew = est.detach().cpu().numpy(), g["est"]
    print("est max err / scale", np.abs(e - ew).max() / np.abs(ew).max(), "rms", np.sqrt(np.mean((e - ew) ** 2)) / np.abs(ew).max())
    rows = []
    for k, p in model.named_parameters():
        if "grad." + k not in g.files:
            continue
        want, got = g["grad." + k], p.grad.cpu().numpy()
        rel = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
        cos = float((got * want).sum() / max(np.linalg.norm(got) * np.linalg.norm(want), 1e-30))
        rows.append((rel, cos, np.linalg.norm(want), k))
    for rel, cos, nrm, k in rows:
        print(f"{rel:9.4f} cos {cos:7.4f} |g| {nrm:10.3e}  {k}")


if __name__ == "__main__":
    main()
