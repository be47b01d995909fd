"""Pins oracle/sepformer_oracle.py against golden vectors produced by the REAL reference
(tools/make_goldens_sepformer.py).  CPU only."""
import numpy as np
import torch

import oracle.fqss_oracle as O
import oracle.sepformer_oracle as S

torch.set_num_threads(1)
TINY = dict(n_src=2, kernel_size=16, stride=8, chunk_size=10, n_heads=4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _cmp_step(g, p, r, s, est_tol, loss_rel, grad_tol):
    np.testing.assert_allclose(r["fest"].numpy(), g[p + "fest"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["est"].detach().numpy(), g[p + "est"], rtol=0, atol=est_tol)
    assert abs(float(r["loss"].detach()) - float(g[p + "loss"])) < loss_rel * abs(float(g[p + "loss"]))
    assert abs(float(r["kd"].detach()) - float(g[p + "kd"])) < loss_rel * abs(float(g[p + "kd"]))
    nograd = set(g[p + "nograd"].tolist())
    n = 0
    for k, v in s.p.items():
        if p + "grad." + k in g.files:
            w = g[p + "grad." + k]
            assert np.abs(v.grad.numpy() - w).max() <= grad_tol * max(np.abs(w).max(), 1e-4), k
            n += 1
        elif k.endswith(".pos.pe"):
            assert v.grad is None          # the positional table is a buffer of the reference, not a parameter
        else:
            assert k in nograd and (v.grad is None or float(v.grad.abs().max()) == 0.0), k
    assert n > 100


def test_sep_tiny_step1_free_running(golden):
    g = golden("sep_tiny_step")
    sd = {k[4:]: T(g[k]) for k in g.files if k.startswith("sd0.")}
    fsd = {k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}
    s, t = S.StudentSepformerQ(sd, **TINY), S.TeacherSepformer(fsd, stride=8, chunk_size=10, n_heads=4)
    tr = O.Trainer(s, t, lr=1.5e-4)
    r = tr.step(T(g["x"]), T(g["tgt"]))
    assert abs(float(r["gnorm"]) - float(g["s1.gnorm"])) < 2e-4 * float(g["s1.gnorm"])
    _cmp_step(g, "s1.", r, s, 5e-6, 3e-5, 5e-4)


def _forced(g, step):
    sd = {k[len(f"s{step}.post_sd."):]: T(g[k]) for k in g.files if k.startswith(f"s{step}.post_sd.")}
    fsd = {k[4:]: T(g[k]) for k in g.files if k.startswith("fsd.")}
    return S.StudentSepformerQ(sd, **TINY), S.TeacherSepformer(fsd, stride=8, chunk_size=10, n_heads=4)


def test_sep_tiny_step2_from_reference_state(golden):
    """first forward with fake-quantized weights, incl. the residual decoder's own weight quantizer (train_res_dec)"""
    g = golden("sep_tiny_step")
    s, t = _forced(g, 1)
    for q in s.wq.values():
        q.observer = False
    for q in s.aq.values():
        q.n_iter = 1
    r = O.kd_step(s, t, T(g["x"]), T(g["tgt"]))
    r["loss"].backward()
    gn = torch.nn.utils.clip_grad_norm_(s.parameters(), 5.0)
    assert abs(float(gn) - float(g["s2.gnorm"])) < 3e-4 * float(g["s2.gnorm"])
    _cmp_step(g, "s2.", r, s, 1e-5, 5e-5, 1e-3)


def test_sep_tiny_step51_from_reference_state(golden):
    g = golden("sep_tiny_step")
    s, t = _forced(g, 50)
    s.leave_observer_phase()
    r = O.kd_step(s, t, T(g["x"]), T(g["tgt"]))
    assert abs(float(r["loss"].detach()) - float(g["s51.loss"])) < 0.1
    want, got = g["s51.est"], r["est"].detach().numpy()
    assert np.abs(got - want).max() < 0.05 * np.abs(want).max()
