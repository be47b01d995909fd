"""Pins oracle/htdemucs_oracle.py against golden vectors produced by the REAL reference
(tools/make_goldens_htdemucs.py -> tests/golden/hd_tiny_step.npz).  CPU only."""
import numpy as np
import torch

import oracle.htdemucs_oracle as H

torch.set_num_threads(2)
KW = dict(n_src=2, audio_channels=2, nfft=2048, depth=4, t_layers=3, t_heads=2)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _sd(g, pre):
    return {k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)}


def test_float_teacher_matches_the_reference(golden):
    g = golden("hd_tiny_step")
    t = H.HTDemucsOracle(_sd(g, "fsd."), quantized=False, eval_length=441000, **KW)     # fmodel.eval(): padded to 10 s
    with torch.no_grad():
        y = t.forward(T(g["mix"]))
    np.testing.assert_allclose(y.numpy(), g["fest"], rtol=1e-4, atol=1e-5 * np.abs(g["fest"]).max())


def test_student_observer_step_and_quantizing_step(golden):
    g = golden("hd_tiny_step")
    mix, src, fest = T(g["mix"]), T(g["src"]), T(g["fest"])
    s = H.HTDemucsOracle(_sd(g, "sd0."), quantized=True, **KW)
    s.enable_observer(True)
    # observer call 1 (every quantizer passes its input through) with a backward
    est = s.forward(mix)
    loss, task, kd, w = H.solver_loss(est, fest, src)
    np.testing.assert_allclose(est.detach().numpy(), g["o1.est"], rtol=1e-4, atol=1e-4 * np.abs(g["o1.est"]).max())
    np.testing.assert_allclose(loss.item(), float(g["o1.loss"]), rtol=1e-5)
    np.testing.assert_allclose(w.numpy(), g["o1.w"], rtol=1e-4)
    loss.backward()
    n = 0
    for k, p in s.p.items():
        if "o1.grad." + k in g.files and not k.endswith("decoder_bias"):
            want = g["o1.grad." + k]
            if np.linalg.norm(want) > 1e-7:
                rel = np.linalg.norm(p.grad.numpy() - want) / np.linalg.norm(want)
                assert rel < 0.08, (k, rel)           # sign gradient of the L1 loss: samples at fp32 noise from a target flip
                n += 1
    assert n > 300
    for p in s.p.values():
        p.grad = None
    with torch.no_grad():
        for _ in range(49):
            est_obs = s.forward(mix)
    np.testing.assert_allclose(est_obs.numpy(), g["est_obs"], rtol=1e-4, atol=2e-4 * np.abs(g["est_obs"]).max())
    for k in g.files:
        if k.startswith("sd_obs."):
            want = g[k]
            assert np.abs(s.p[k[7:]].detach().numpy() - want).max() <= 5e-3 * max(np.abs(want).max(), 1e-3), k
    # the fixture's tightened ranges, then the quantizing step
    with torch.no_grad():
        for k in g.files:
            if k.startswith("sd."):
                s.p[k[3:]].copy_(T(g[k]))
    est = s.forward(mix)
    loss, task, kd, w = H.solver_loss(est, fest, src)
    ew = g["est"]
    scale = np.abs(ew).max()
    e = est.detach().numpy()
    assert np.abs(e - ew).max() <= 0.05 * scale and np.sqrt(np.mean((e - ew) ** 2)) <= 5e-3 * scale
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-3)
    np.testing.assert_allclose(task.detach().numpy(), g["task"], rtol=2e-3)
    np.testing.assert_allclose(kd.detach().numpy(), g["kd"], rtol=5e-3)
