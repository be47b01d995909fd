"""GPU: the `train.py -env asteroid` plugin surface end to end on synthetic mixtures."""
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def test_asteroid_env_trains_and_exports(tmp_path):
    assert torch.cuda.is_available()
    from fqss_amd.train_env.asteroid_librimix import asteroid_librimix_trainer as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = yaml.safe_load(open(os.path.join(root, "configs", "convtasnet_2spks_8k_synthetic.yaml")))
    conf["work_dir"] = str(tmp_path / "run")
    conf["dataset_cfg"].update(segment=0.5, steps_per_epoch=4, val_steps=2)
    conf["training_cfg"].update(epochs=2, batch_size=2)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(conf))
    hist = T.train(str(yml), "cuda")
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["loss"])) for h in hist)
    assert hist[-1]["loss"] < hist[0]["loss"] + 5.0
    sd = torch.load(os.path.join(conf["work_dir"], "best_model.pth"))
    assert len(sd) == 948 and "masker.TCN.0.shared_block.0.activation_fake_quantize.min_range" in sd
    assert os.path.exists(os.path.join(conf["work_dir"], "conf.yml"))
    # (`device == "cpu"` selects the CPU backend of the same C ABI, cfg 1 of BASELINE.json: tests/test_cpu_backend.py)
    # kd_lambda = 0: the teacher-free PIT SI-SDR loss of mysystem.py:153-156 (fqss_pit_sisdr_loss; the teacher is never run)
    conf["work_dir"] = str(tmp_path / "run0")
    conf["training_cfg"].update(kd_lambda=0, epochs=1)
    yml0 = tmp_path / "cfg0.yaml"
    yml0.write_text(yaml.safe_dump(conf))
    hist0 = T.train(str(yml0), "cuda")
    assert len(hist0) == 1 and torch.isfinite(torch.tensor(hist0[0]["loss"]))


def test_asteroid_env_trains_dptnet(tmp_path):
    """cfg 3 through the same env plugin: DPTNet from the YAML, StepLR (`step_lr`) on the stepper's learning rate"""
    from fqss_amd.train_env.asteroid_librimix import asteroid_librimix_trainer as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = yaml.safe_load(open(os.path.join(root, "configs", "dptnet_2spks_8k_synthetic.yaml")))
    conf["work_dir"] = str(tmp_path / "run")
    conf["dataset_cfg"].update(segment=0.25, steps_per_epoch=14, val_steps=1)
    conf["training_cfg"].update(epochs=4)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(conf))
    hist = T.train(str(yml), "cuda")
    assert len(hist) == 4 and all(torch.isfinite(torch.tensor(h["loss"])) for h in hist)
    # 56 steps: the 50-call observer phase ends inside epoch 4, one eager quantizing step, then the captured step replays
    assert [h["launch"] for h in hist] == ["eager", "eager", "eager", "hipGraph replay"]
    assert abs(hist[0]["lr"] - 4e-4) < 1e-12 and abs(hist[2]["lr"] - 4e-4 * 0.98) < 1e-12 and abs(hist[3]["lr"] - 4e-4 * 0.98) < 1e-12
    sd = torch.load(os.path.join(conf["work_dir"], "best_model.pth"))
    assert "separator.DPT.row_transformer.0.transformer.lstm.weight_quantizers_dict.weight_hh_l0.min_range" in sd


def test_speechbrain_env_trains_sepformer(tmp_path):
    """cfg 4: `train(yml, local_rank, distributed_launch, device)` of the speechbrain env on the reference's YAML dialect
    (!ref / !new: tags), one sample per GPU"""
    from fqss_amd.train_env.speechbrain_librimix import speechbrain_librimix_trainer as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "configs", "sepformer_2spks_8k_synthetic.yaml")).read()
    text = text.replace("work_dir: /tmp/fqss_sepformer_synth", f"work_dir: {tmp_path / 'run'}")
    text = text.replace("training_signal_len: 32000", "training_signal_len: 4000").replace("steps_per_epoch: 60", "steps_per_epoch: 3")
    text = text.replace("val_steps: 4", "val_steps: 1")
    yml = tmp_path / "cfg.yaml"
    yml.write_text(text)
    hp = T.load_hparams(str(yml))
    assert hp["num_spks"] == 2 and hp["save_folder"].endswith("run/save") and hp["lr_scheduler"]["patience"] == 3
    hist = T.train(str(yml), 0, False, "cuda")
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["train_loss"])) for h in hist)
    sd = torch.load(os.path.join(str(tmp_path / "run"), "save", "best_model.pth"))
    assert "decoder.residual_error_block.weight_fake_quantize_dec.min_range" in sd and "masker.layers.0.intra_transformer_block.pos.pe" in sd
    # per-GPU batch 2 = n_src: the per-sample objective with the reference's weight broadcast; batch 3 fails like the reference
    two = tmp_path / "two.yaml"
    two.write_text(text.replace("batch_size: 1\n", "batch_size: 2\n", 1).replace(str(tmp_path / "run"), str(tmp_path / "run2")))
    hist2 = T.train(str(two), 0, False, "cuda")
    assert len(hist2) == 2 and all(torch.isfinite(torch.tensor(h["train_loss"])) for h in hist2)
    with pytest.raises(ValueError, match="batch_size must be 1 or n_src"):
        bad = tmp_path / "bad.yaml"
        bad.write_text(text.replace("batch_size: 1\n", "batch_size: 3\n", 1))
        T.train(str(bad), 0, False, "cuda")


def test_htdemucs_env_trains(tmp_path, monkeypatch):
    """cfg 5: `main()` of the htdemucs env with hydra-style overrides on sys.argv, a tiny HTDemucs on synthetic stems: the observer
    phase ends inside epoch 2 and the step is captured; best.th holds the reference's package keys"""
    import sys
    from fqss_amd.train_env.htdemucs_musdbhq import train as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = yaml.safe_load(open(os.path.join(root, "configs", "htdemucs_synthetic.yaml")))
    conf["work_dir"] = str(tmp_path / "run")
    conf["dset"].update(segment=0.05, sources=["a", "b"], steps_per_epoch=28, valid_steps=1)
    conf.update(epochs=2, batch_size=2, weights=[1.0, 1.0])
    conf["htdemucs"] = dict(nfft=2048, channels=8, bottom_channels=16, t_layers=3, t_heads=2)
    yml = tmp_path / "cfg.yaml"
    yml.write_text(yaml.safe_dump(conf))
    monkeypatch.setattr(sys, "argv", ["train.py", "+device=cuda", f"+yml_path={yml}", "optim.lr=0.0002"])
    hist = T.main()
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["train"]["loss"])) and torch.isfinite(torch.tensor(h["valid"]["loss"])) for h in hist)
    assert [h["train"]["launch"] for h in hist] == ["eager", "hipGraph replay"]
    pkg = torch.load(os.path.join(conf["work_dir"], "best.th"))
    assert set(pkg) == {"state", "kwargs", "history"} and pkg["kwargs"]["nfft"] == 2048
    assert "decoder.3.conv_tr.residual_error_block.weight_fake_quantize_dec.min_range" in pkg["state"]
    assert "crosstransformer.layers.1.cross_attn.activation_fake_quantize_head.max_range" in pkg["state"]


def test_val_cli_on_synthetic_mixtures(tmp_path):
    """`val.py -y cfg.yaml` (val.py:184-226): quantized model from the YAML (+ a checkpoint written by the asteroid env), chunked
    inference, SI-SDR and its improvement over the mixture"""
    from fqss_amd import val as V
    from fqss_amd.quantization.qat.models.load_model import create_pretrained_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = yaml.safe_load(open(os.path.join(root, "configs", "convtasnet_2spks_8k_synthetic.yaml")))
    ckpt = tmp_path / "best_model.pth"
    torch.save(create_pretrained_model(dict(conf["model_cfg"], model_path=None)).state_dict(), ckpt)
    conf["model_cfg"]["model_path"] = str(ckpt)
    conf["testing_cfg"] = dict(n_items=2, length_samples=12000, segment_samples=8000, overlap=0.25)
    yml = tmp_path / "val.yaml"
    yml.write_text(yaml.safe_dump(conf))
    sisdr, imp = V.val(["-y", str(yml)])
    assert torch.isfinite(torch.tensor([sisdr, imp])).all()


def test_infer_cli_separates_a_wav_file(tmp_path):
    """`infer.py -y cfg.yaml -a mix.wav --normalize` (infer.py:25-87): wav in, one peak-normalised 16-bit wav per source out, chunked
    inference on the serving path (runtime.InferRunner: codes-only forward as a hipGraph per chunk shape) -- equal to the plain
    eval-mode `process.model_infer` of the same model up to the 16-bit file format"""
    from scipy.io import wavfile
    from fqss_amd import infer as I
    from fqss_amd.data import synth_batch
    from fqss_amd.process import model_infer
    from fqss_amd.quantization.qat.models.load_model import create_pretrained_model, enable_observer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = yaml.safe_load(open(os.path.join(root, "configs", "convtasnet_2spks_8k_synthetic.yaml")))
    ckpt = tmp_path / "best_model.pth"
    torch.save(create_pretrained_model(dict(conf["model_cfg"], model_path=None)).state_dict(), ckpt)
    conf["model_cfg"]["model_path"] = str(ckpt)
    conf["work_dir"] = str(tmp_path / "out")
    conf["testing_cfg"] = dict(segment_samples=8000, overlap=0.25)
    conf["dataset_cfg"]["resample"] = 1                  # (the file below is already at the model's 8 kHz)
    yml = tmp_path / "infer.yaml"
    yml.write_text(yaml.safe_dump(conf))
    mix, _ = synth_batch(1, 20000, seed=77, device="cpu")
    pcm = (mix[0, 0].clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy()
    wav = tmp_path / "mix.wav"
    wavfile.write(str(wav), 8000, pcm)
    paths = I.infer(["-y", str(yml), "-a", str(wav), "--normalize"])
    assert [os.path.basename(p) for p in paths] == ["output0.wav", "output1.wav"]
    # the same through the plain module path
    model = create_pretrained_model(conf["model_cfg"])
    enable_observer(model, False)
    model.to("cuda").eval()
    x = torch.from_numpy(pcm.astype("float32") / 32768.0).reshape(1, -1).cuda()
    m, sd = x.mean(), x.std()
    ref = model_infer(model, (x - m) / sd, n_srcs=2, segment=8000, overlap=0.25, device="cuda") * sd + m
    for k, p in enumerate(paths):
        fs, a = wavfile.read(p)
        assert fs == 8000 and a.dtype.name == "int16" and a.shape == (20000,)
        got = torch.from_numpy(a.astype("float32") / 32768.0)
        want = (ref[k] / ref[k].abs().max()).reshape(-1).cpu()
        assert abs(float(got.abs().max()) - 1.0) <= 1e-3                     # peak-normalised (process.normalize_audio)
        assert float((got - want).abs().max()) <= 2.0 / 32768.0, float((got - want).abs().max())
