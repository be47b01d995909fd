#!/usr/bin/env python3
"""Layer fixture for BatchNormQ (qat_layers.py:472-486) from the REAL reference (this container only): BatchNorm1d on [B, C, M] and
BatchNorm2d on [B, C, H, W], training mode (batch statistics, running estimates updated) and eval mode (running statistics), one
observer call then a quantizing forward + backward each, like tools/make_goldens.py::_run_layer.
    python tools/make_goldens_bn.py -> tests/golden/bn_layers.npz"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as MG  # noqa: E402  (installs the shim, imports the reference)

RL, RQ, npy, keyed_randn = MG.RL, MG.RQ, MG.npy, MG.keyed_randn


def run(name, bn, x, d, train):
    with torch.no_grad():
        bn.weight.copy_(1.0 + keyed_randn(name + ".w", tuple(bn.weight.shape), 0.2))
        bn.bias.copy_(keyed_randn(name + ".b", tuple(bn.bias.shape), 0.1))
        bn.running_mean.copy_(keyed_randn(name + ".rm", tuple(bn.running_mean.shape), 0.3))
        bn.running_var.copy_(1.0 + keyed_randn(name + ".rv", tuple(bn.running_var.shape), 0.2).abs())
    L = RL.BatchNormQ(bn, gradient_based=True, act_quant=True)
    for k, v in L.state_dict().items():
        d[f"{name}.sd0.{k}"] = npy(v)
    L.train(train)
    MG.enable_observer(L, True)
    with torch.no_grad():
        y_obs = L(x)
    for m in L.modules():
        if isinstance(m, RQ.GradientActivationFakeQuantize):
            m.n_iter = m.max_observations
    MG._set_act_range(L.activation_fake_quantize, y_obs)
    for k, v in L.state_dict().items():
        d[f"{name}.sd1.{k}"] = npy(v)          # state in front of the quantizing call (running statistics after the observer call)
    xi = x.clone().requires_grad_(True)
    y = L(xi)
    g = keyed_randn(name + ".gout", tuple(y.shape))
    y.backward(g)
    d[f"{name}.in0"], d[f"{name}.gin0"], d[f"{name}.out_obs"], d[f"{name}.out"], d[f"{name}.gout"] = npy(x), npy(xi.grad), npy(y_obs), npy(y), npy(g)
    for k, v in L.state_dict().items():
        d[f"{name}.sd2.{k}"] = npy(v)
    for k, p in L.named_parameters():
        if p.grad is not None:
            d[f"{name}.grad.{k}"] = npy(p.grad)


def main():
    d = {}
    x1 = keyed_randn("bn.x1", (3, 12, 50), 1.3) + 0.4
    x2 = keyed_randn("bn.x2", (2, 6, 9, 11), 0.7) - 0.2
    for train in (True, False):
        t = "train" if train else "eval"
        run(f"bn1d_{t}", nn.BatchNorm1d(12), x1, d, train)
        run(f"bn2d_{t}", nn.BatchNorm2d(6, momentum=0.3, eps=1e-3), x2, d, train)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bn_layers.npz")
    np.savez_compressed(out, **d)
    print("bn_layers:", len(d), "arrays")


if __name__ == "__main__":
    main()
