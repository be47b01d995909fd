"""CPU: the data side's host logic -- batch order against torch's own samplers, the LibriMix datasets of both
envs on a generated WAV + CSV tree through the background reader (on the CPU backend of the C ABI: the same host code, host tensors), the
reference's RNG consumption order, and the trainers' `dataset_cfg.name: librimix` entry points."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from .helpers_librimix import make_librimix_tree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_backend():
    from fqss_amd import _lib
    if not os.path.exists(_lib.CPU_SO_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(_lib.CPU_SO_PATH)])
    _lib.set_backend("cpu")
    yield
    _lib.set_backend("hip")


def test_epoch_batches_follow_torch_samplers():
    """the index order of DataLoader(shuffle=True, drop_last=True) (asteroid_librimix_trainer.py:53-59) and of DistributedSampler"""
    from torch.utils.data import BatchSampler, DistributedSampler, RandomSampler
    from fqss_amd.loader import epoch_batches
    data = list(range(23))
    torch.manual_seed(7)
    ref = list(BatchSampler(RandomSampler(data), batch_size=4, drop_last=True))
    torch.manual_seed(7)
    assert epoch_batches(23, 4, shuffle=True, drop_last=True) == ref and len(ref) == 5
    assert epoch_batches(23, 4, shuffle=False, drop_last=False)[-1] == [20, 21, 22]
    for rank in range(3):
        smp = DistributedSampler(data, num_replicas=3, rank=rank, shuffle=True, seed=5)
        smp.set_epoch(2)
        idx = list(smp)
        got = epoch_batches(23, 2, shuffle=True, drop_last=True, rank=rank, world=3, seed=5, epoch=2)
        assert [i for b in got for i in b] == idx[:len(idx) // 2 * 2]
        smp = DistributedSampler(data, num_replicas=3, rank=rank, shuffle=False)
        assert [i for b in epoch_batches(23, 1, False, False, rank, 3) for i in b] == list(smp)


def test_librimix_through_the_prefetcher_matches_direct_batches(tmp_path, cpu_backend):
    """Prefetcher(LibriMix) yields, batch for batch, what `ds.batch(indices)` returns under the same seeds (the reader thread consumes
    `random` / `np.random` in item order, as the reference's __getitem__ does: start_i, prob_i, snr_i); resampled sources equal the
    oracle's fp64 restatement of the sinc kernel on the PCM samples read from disk"""
    import oracle.fqss_oracle as O
    from fqss_amd.loader import Prefetcher, epoch_batches, with_lookahead
    from fqss_amd.train_env.asteroid_librimix.librimix_dataset import LibriMix, read_wav
    tree = make_librimix_tree(tmp_path, n_train=12, short=0.6)
    aug = {"distribution": "uniform", "param0": -5, "param1": 5, "prob": 0.5}
    ds = LibriMix(tree["train_dir"], task="sep_clean", sample_rate=16000, resample=0.5, n_src=2, segment=1, augmentation_cfg=aug,
                  device="cpu")
    assert len(ds) == 12                                     # the 0.6 s clip is dropped
    torch.manual_seed(3)
    batches = epoch_batches(len(ds), 4, shuffle=True, drop_last=True)
    assert len(batches) == 3
    random.seed(11); np.random.seed(11)
    direct = [ds.batch(b) for b in batches]
    random.seed(11); np.random.seed(11)
    pf = Prefetcher(ds, batches, "cpu", depth=2)
    got = [(x.clone(), t.clone(), None if xn is None else xn.clone()) for x, t, xn in with_lookahead(pf)]
    assert len(got) == 3 and got[-1][2] is None
    for k, ((mx, src), (x, t, xn)) in enumerate(zip(direct, got)):
        assert x.shape == (4, 1, 8000) and t.shape == (4, 2, 8000)
        assert torch.equal(mx, x) and torch.equal(src, t)
        if k + 1 < len(got):
            assert torch.equal(xn, got[k + 1][0])
    # the draws: per item randint(0, length - seg), uniform() < prob, then uniform(-5, 5) when augmenting
    random.seed(11); np.random.seed(11)
    i0 = batches[0][0]
    start = random.randint(0, int(ds._col["length"][i0]) - 16000)
    augmented = np.random.uniform() < 0.5
    clip = torch.from_numpy(read_wav(ds._col["source_1_path"][i0], start, start + 16000))
    ref = O.resample_sinc(clip[None], 16000, 8000)[0]
    np.testing.assert_allclose(direct[0][1][0, 0].numpy(), ref.numpy(), rtol=0, atol=3e-7)
    if not augmented:                                        # the mixture file = s1 + s2 to PCM16 rounding; resampling is linear
        np.testing.assert_allclose(direct[0][0][0, 0].numpy(), direct[0][1][0].sum(0).numpy(), atol=3e-4)
    # an augmented item somewhere in the epoch: the mixture is a1 s1 + a2 s2 at an SNR inside [-5, 5] dB
    mixes, srcs = torch.cat([d[0] for d in direct]), torch.cat([d[1] for d in direct])
    a = torch.linalg.lstsq(srcs.transpose(1, 2), mixes.transpose(1, 2)).solution.squeeze(-1)
    e = lambda v: v.pow(2).mean(-1)
    snr = 10 * torch.log10(e(srcs[:, 0] * a[:, :1]) / e(srcs[:, 1] * a[:, 1:]))
    rescaled = (a - 1).abs().max(1).values > 1e-3
    assert rescaled.any() and not rescaled.all() and bool((snr[rescaled].abs() <= 5.1).all())


def test_reader_errors_reach_the_training_thread(tmp_path, cpu_backend):
    from fqss_amd.loader import Prefetcher
    from fqss_amd.train_env.asteroid_librimix.librimix_dataset import LibriMix
    tree = make_librimix_tree(tmp_path, n_train=4)
    ds = LibriMix(tree["train_dir"], task="sep_clean", sample_rate=16000, resample=0.5, n_src=2, segment=1, device="cpu")
    os.remove(ds._col["source_2_path"][2])
    it = iter(Prefetcher(ds, [[0, 1], [2, 3]], "cpu"))
    next(it)
    with pytest.raises(FileNotFoundError):
        next(it)


def test_speechbrain_dataset_csv_and_augmentation(tmp_path, cpu_backend):
    """prepare_librimix writes the reference's CSV names / columns (prepare_data.py:56-140); SbLibriMix serves whole utterances, and in
    training mode speed-perturbed sources whose sum is the mixture, cut to training_signal_len"""
    import csv
    from fqss_amd.train_env.speechbrain_librimix.prepare_data import SbLibriMix, prepare_librimix
    tree = make_librimix_tree(tmp_path, n_train=5, n_dev=2, seconds=(1.5, 2.0))
    save = tmp_path / "save"
    prepare_librimix(tree["data_folder"], str(save), n_spks=2)
    assert sorted(os.listdir(save)) == ["libri2mix_dev.csv", "libri2mix_test.csv", "libri2mix_train-360.csv"]
    rows = list(csv.DictReader(open(save / "libri2mix_train-360.csv")))
    assert list(rows[0]) == ["ID", "duration", "mix_wav", "mix_wav_format", "mix_wav_opts", "s1_wav", "s1_wav_format", "s1_wav_opts",
                             "s2_wav", "s2_wav_format", "s2_wav_opts", "noise_wav", "noise_wav_format", "noise_wav_opts"]
    assert len(rows) == 5 and rows[0]["mix_wav"].endswith(".wav") and "/mix_clean/" in rows[0]["mix_wav"] and "/s2/" in rows[0]["s2_wav"]
    val = SbLibriMix(str(save / "libri2mix_dev.csv"), device="cpu")
    x, t = val.batch([0])
    assert x.shape[:2] == (1, 1) and t.shape[:2] == (1, 2) and x.shape[-1] == t.shape[-1] >= 12000       # 1.5-2 s at 8 kHz
    np.testing.assert_allclose(x[0, 0].numpy(), t[0].sum(0).numpy(), atol=3e-4)
    tr = SbLibriMix(str(save / "libri2mix_train-360.csv"), train=True, training_signal_len=8000, device="cpu")
    torch.manual_seed(0)
    seen = set()
    for i in range(5):
        x, t = tr.batch([i])
        assert x.shape == (1, 1, 8000) and t.shape == (1, 2, 8000)
        assert torch.equal(x[:, 0], t.sum(1))               # re-mixed from the perturbed sources (:299-300)
        seen.add(round(float(t.abs().sum()), 3))
    assert len(seen) == 5
    x2, t2 = SbLibriMix(str(save / "libri2mix_dev.csv"), device="cpu").batch([0, 1])      # PaddedBatch: zero-padded to the longest
    assert x2.shape[0] == 2 and t2.shape[:2] == (2, 2)


def test_train_cli_use_cpu_on_a_librimix_tree(tmp_path):
    """`python -m fqss_amd.train -env asteroid -y <the reference's dataset_cfg> --use_cpu`: dataset_cfg.name librimix with train_dir /
    valid_dir / task / resample / segment / augmentation (configs/convtasnet_2spks_8k.yaml:27-41) trains from the CSV tree"""
    import yaml
    tree = make_librimix_tree(tmp_path, n_train=6, n_dev=2)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "convtasnet_2spks_8k_cpu.yaml")))
    cfg["work_dir"] = str(tmp_path / "run")
    cfg["dataset_cfg"] = {"name": "librimix", "task": "sep_clean", "train_dir": tree["train_dir"], "valid_dir": tree["valid_dir"],
                          "sample_rate": 16000, "resample": 0.5, "n_src": 2, "noisy": False, "segment": 0.5,
                          "augmentation": {"enable": True, "distribution": "uniform", "param0": -10, "param1": 10}}
    cfg["training_cfg"].update(epochs=1, batch_size=2, num_workers=4, half_lr=True, early_stop=True)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, "-m", "fqss_amd.train", "-env", "asteroid", "-y", str(yml), "--use_cpu"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "Training set size: 6" in p.stdout and "Training is done!" in p.stdout
    assert os.path.exists(tmp_path / "run" / "best_model.pth")
