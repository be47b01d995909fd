"""GPU: the data side connected -- the asteroid / speechbrain trainers and val.py take the
reference's `dataset_cfg` (name: librimix) and the background reader keeps up with the captured step."""
import os
import time

import pytest
import torch
import yaml

from .helpers_librimix import make_librimix_tree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asteroid_env_trains_from_a_librimix_tree_at_the_synthetic_step_time(tmp_path):
    """`train(-env asteroid, -y <configs/convtasnet_2spks_8k.yaml with batch 8 x 4 s>)` on a generated 320-clip LibriMix tree: five epochs
    of 40 steps -- the observer phase and the capture fall into epochs 0-1, epochs 2-4 are pure hipGraph replays fed by the reader thread
    (WAV reads + upload + resampling of batch n+1 on a side stream; its mixture is the teacher's look-ahead input).  Gate: the per-step wall
    time of the replay epochs (pipeline fill at the epoch's start included) is within 5 % of replaying the SAME captured step on
    device-resident batches (what bench.py times; both under this suite's NaN-poisoned carriers, which cost ~2 ms of fill kernels).
    (`loader_wait_s` is reported, not gated: nothing in the loop synchronises with the GPU, so the training thread runs ahead until the
    reader's slot reuse -- which waits for the step that consumed the slot -- pushes back; blocking there is back-pressure.)"""
    from fqss_amd import val as V
    from fqss_amd.train_env.asteroid_librimix import asteroid_librimix_trainer as T
    tree = make_librimix_tree(tmp_path, n_train=320, n_dev=8, seconds=(4.1, 4.6))
    conf = yaml.safe_load(open(os.path.join(ROOT, "configs", "convtasnet_2spks_8k.yaml")))
    conf["work_dir"] = str(tmp_path / "run")
    conf["dataset_cfg"].update(train_dir=tree["train_dir"], valid_dir=tree["valid_dir"], segment=4)
    conf["dataset_cfg"]["augmentation"]["enable"] = True
    conf["training_cfg"].update(epochs=5, batch_size=8)
    conf["testing_cfg"].update(test_dir=tree["test_dir"], n_items=3, segment_samples=16000)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(conf))
    hist = T.train(str(yml), "cuda")
    assert len(hist) == 5 and all(torch.isfinite(torch.tensor([h["loss"], h["val_loss"]])).all() for h in hist)
    assert [h["launch"] for h in hist] == ["eager", "hipGraph replay", "hipGraph replay", "hipGraph replay", "hipGraph replay"]
    step = T.LAST_SYSTEM.stepper
    assert step.teacher_ahead and step._tgraph is not None
    # the same captured step on device-resident batches, alternating two mixtures with the look-ahead announced (bench.py's loop)
    from fqss_amd.data import synth_batch
    X = [synth_batch(8, 32000, seed=s, device="cuda") for s in (1, 2)]
    for it in range(6):
        step(*X[it & 1], x_next=X[(it + 1) & 1][0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(40):
        step(*X[it & 1], x_next=X[(it + 1) & 1][0])
    torch.cuda.synchronize()
    resident_ms = (time.perf_counter() - t0) / 40 * 1e3
    fed_ms = min(hist[3]["train_ms_per_step"], hist[4]["train_ms_per_step"])
    print(f"\nstep fed by the LibriMix reader: {hist[3]['train_ms_per_step']:.2f} / {hist[4]['train_ms_per_step']:.2f} ms per step (epochs 3 / 4); "
          f"on device-resident batches {resident_ms:.2f} ms")
    assert fed_ms <= 1.05 * resident_ms, (fed_ms, resident_ms)
    # val.py on the tree's test folder with the checkpoint the run wrote (val.py:59-92)
    conf["model_cfg"]["model_path"] = os.path.join(conf["work_dir"], "best_model.pth")
    yml.write_text(yaml.safe_dump(conf))
    sisdr, imp = V.val(["-y", str(yml)])
    assert torch.isfinite(torch.tensor([sisdr, imp])).all()


def test_speechbrain_env_trains_from_a_librimix_folder(tmp_path):
    """cfg 4's env on `dataset_cfg.name: librimix` (configs/sepformer_2spks_8k.yaml:27-39 of the reference): prepare_librimix CSVs in
    save_folder, whole utterances, speed perturbation + re-mix + random cut on the device; one utterance shorter than
    training_signal_len arrives as a batch of another shape and runs eagerly between replays"""
    from fqss_amd.train_env.speechbrain_librimix import speechbrain_librimix_trainer as T
    tree = make_librimix_tree(tmp_path, n_train=29, n_dev=2, seconds=(1.3, 1.6), short=0.8)
    conf = open(os.path.join(ROOT, "configs", "sepformer_2spks_8k_synthetic.yaml")).read()
    hp = yaml.load(conf, Loader=T._Loader)
    hp["work_dir"] = str(tmp_path / "run")
    hp["save_folder"] = str(tmp_path / "run" / "save")
    hp["dataset_cfg"] = {"name": "librimix", "task": "sep_clean", "data_folder": tree["data_folder"], "skip_prep": False,
                         "sample_rate": 16000, "resample": 0.5, "noisy": False}
    hp.update(N_epochs=2, batch_size=1, training_signal_len=8000, limit_training_signal_len=True, use_speedperturb=True,
              speedperturb={"perturb_prob": 1.0, "speeds": [95, 100, 105]}, num_spks=2)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(hp))
    hist = T.train(str(yml), 0, False, "cuda")
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor([h["train_loss"], h["valid_si-snr"]])).all() for h in hist)
    assert hist[1]["launch"] == "hipGraph replay"
    assert sorted(os.listdir(hp["save_folder"]))[:3] == ["best_model.pth", "latest_model.pth", "libri2mix_dev.csv"]
