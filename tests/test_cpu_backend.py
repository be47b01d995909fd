"""cfg 1 of BASELINE.json ("... asteroid env, CPU, batch 2, 1 s ... plumbing, no GPU"; reference train.py:31 `--use_cpu`): the CPU backend
behind the same C ABI (fqss_amd/csrc/cpu/libfqss_cpu.so, selected by `_lib.set_backend("cpu")`), pinned by the SAME reference-generated
fixtures as the HIP kernels -- the bodies of the GPU parity tests run here on host tensors: quantizer goldens bit-exact (G0), every
ConvTasNet LayerQ teacher-forced (G1), the tiny model's 53 QAT steps (G2 at steps 1-2), and the trainer CLI end to end at the cfg-1
size.  The oracle (oracle/) is the checker in those bodies, never the thing computed with."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_backend(monkeypatch):
    from fqss_amd import _lib, smoke
    if not os.path.exists(_lib.CPU_SO_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(_lib.CPU_SO_PATH)])
    _lib.set_backend("cpu")
    # the GPU test bodies say `.cuda()`: on this backend tensors stay on the host
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    real = smoke.build_pair
    monkeypatch.setattr(smoke, "build_pair", lambda device="cpu", seed=0, **kw: real("cpu", seed, **kw))
    yield
    _lib.set_backend("hip")


def test_cpu_library_exports_only_declared_symbols():
    """every symbol of the CPU library is an entry point of include/fqss.h with the SAME name (the host binds both through one table)"""
    from fqss_amd import _lib
    if not os.path.exists(_lib.CPU_SO_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(_lib.CPU_SO_PATH)])
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.CPU_SO_PATH], capture_output=True, text=True, check=True).stdout
    syms = {l.split()[-1] for l in out.splitlines() if " T fqss_" in l}
    assert len(syms) >= 30 and syms <= set(_lib.EXPORTS), syms - set(_lib.EXPORTS)


def test_hip_backend_still_refuses_host_tensors():
    from fqss_amd import _lib
    from fqss_amd.quantization.qat import qat_quant as QQ
    assert _lib.BACKEND == "hip"
    with pytest.raises(_lib.FqssError):
        QQ.GradientActivationFakeQuantize(True)(torch.randn(4, 4))


def test_quantizer_goldens_bit_exact_on_the_cpu_backend(golden, cpu_backend):
    """G0 on the CPU backend: bin index, de-quantised value and gx of the reference's linear_quantize (qat_quant.py:136-147) bit for
    bit, range gradients to summation-order noise; the per-channel weight quantizer incl. ch_out_idx = 1"""
    import numpy as np
    from fqss_amd import kernels as K
    g = golden("fq_act")
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
    for i in range(int(g["n_cases"])):
        x = T(g[f"x{i}"])
        lo, hi = T(g[f"range{i}"][:1]), T(g[f"range{i}"][1:])
        y, idx = K.actq_fwd(x, K.ACT_NONE, None, K.Q_QUANT, lo, hi, None, want_idx=True, dense_idx=True)
        assert np.array_equal(idx.numpy(), g[f"idx{i}"]) and np.array_equal(y.numpy(), g[f"y{i}"])
        gacc = torch.zeros(K.GACC_DOUBLES, dtype=torch.float64)
        gx = K.actq_bwd(x, T(g[f"g{i}"]), K.ACT_NONE, None, K.Q_QUANT, lo, hi, gacc)
        assert np.array_equal(gx.numpy(), g[f"gx{i}"])
        ga = gacc.view(-1, 3).sum(0).numpy()
        np.testing.assert_allclose(ga[0], g[f"gmin{i}"][0], rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(ga[1], g[f"gmax{i}"][0], rtol=2e-5, atol=1e-5)


def test_layer_goldens_on_the_cpu_backend(golden, cpu_backend):
    from tests import test_gpu_model as G
    for name in G.LAYER_NAMES:
        G.test_layer_goldens_teacher_forced(golden, name)


def test_tiny_training_on_the_cpu_backend(golden, cpu_backend):
    from tests import test_gpu_model as G
    G.test_tiny_training_vs_reference_goldens(golden)


def test_cfg1_full_size_steps_on_the_cpu_backend(golden, cpu_backend):
    """BASELINE.json configs[0] at its own size: the FULL ConvTasNetQ (5.1 M parameters), B = 2, T = 8000, steps 1-2 against the
    digests of the reference's run (tests/golden/cfg1_step.npz): loss / KD / task 1e-5, estimate 1e-4, every per-parameter gradient
    norm at the G2 tolerances of the GPU test whose body this is"""
    from tests import test_gpu_model as G
    G.test_full_size_vs_reference_goldens(golden, "cfg1_step", 2, 8000, n_steps=2, device="cpu")


def test_cpu_backend_under_address_sanitizer():
    """the same source under -fsanitize=address,undefined with a driver that calls every entry point on small ragged shapes"""
    d = os.path.join(ROOT, "fqss_amd", "csrc", "cpu")
    p = subprocess.run(["make", "-C", d, "asan"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "selftest ok" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_unbuilt_entry_points_raise_on_the_cpu_backend(cpu_backend):
    from fqss_amd import _lib, kernels as K
    with pytest.raises(_lib.FqssError, match="not built for the cpu backend"):
        K.split3_planes(torch.randn(8, 8))


def test_train_cli_use_cpu(tmp_path):
    """`python -m fqss_amd.train -env asteroid -y configs/convtasnet_2spks_8k_cpu.yaml --use_cpu`: full-size ConvTasNetQ, batch 2 x 1 s"""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "convtasnet_2spks_8k_cpu.yaml")))
    cfg["work_dir"] = str(tmp_path / "run")
    cfg["dataset_cfg"]["steps_per_epoch"] = 2
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, "-m", "fqss_amd.train", "-env", "asteroid", "-y", str(yml), "--use_cpu"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "Training is done!" in p.stdout
    sd = torch.load(tmp_path / "run" / "best_model.pth", weights_only=True)
    assert len(sd) == 948 and all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())
