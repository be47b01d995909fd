#!/usr/bin/env python3
"""`infer.py -y cfg.yaml -a mixture.wav [--normalize] [--plot]` (reference: infer.py:25-87): separate ONE audio file with the quantized
model of the YAML and write `output<k>.wav` per source into `work_dir`.

Same flags, same order of operations: read the wav (optionally resampled by `dataset_cfg.resample`), optional mean / std
normalisation, `process.model_infer` (whole utterance, or chunks of `testing_cfg.segment_samples` blended by the triangular window),
de-normalisation, peak normalisation of every source (`process.normalize_audio`, process.py:54-55), 16-bit wav files.  The forward
runs on the serving path of this build: `runtime.InferRunner` -- eval mode, codes-only dataflow, one hipGraph per chunk shape -- so
the chunks of a long file after the first cost one graph launch each.  The wav container I/O is the reference's third-party side
(torchaudio, utils.py:25-42: absent here): RIFF files go through `scipy.io.wavfile`; resampling through the build's own polyphase FIR
(fqss_resample_fir).  `--use_cpu` is refused: inference runs on ROCm devices (oracle/ is the CPU checker)."""
import argparse
import os

import numpy as np
import torch
import yaml

from . import kernels as K
from .process import model_infer
from .quantization.qat.models.load_model import create_pretrained_model, enable_observer
from .runtime import InferRunner


def argument_handler(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--yml_path", "-y", type=str, required=True, help="YML configuration file")
    p.add_argument("--audio_path", "-a", type=str, required=True, help="Input audio path to separate")
    p.add_argument("--use_cpu", action="store_true", help="Use cpu")
    p.add_argument("--normalize", action="store_true", help="normalize input/output begore inference")
    p.add_argument("--plot", action="store_true", help="Plot waveform figure")
    return p.parse_args(argv)


def read_audio(path, resample=1, device="cuda"):
    """utils.read_audio (utils.py:25-30): [channels, samples] float32 in [-1, 1) and the (resampled) rate"""
    from scipy.io import wavfile
    fs, a = wavfile.read(path)
    a = np.asarray(a)
    if a.dtype == np.int16:
        x = a.astype(np.float32) / 32768.0
    elif a.dtype == np.int32:
        x = a.astype(np.float32) / 2147483648.0
    elif a.dtype == np.uint8:
        x = (a.astype(np.float32) - 128.0) / 128.0
    else:
        x = a.astype(np.float32)
    x = torch.from_numpy(x.reshape(len(x), -1).T.copy()).to(device)
    if resample != 1:
        x = K.resample(x, int(fs), int(fs * resample))
        fs = int(fs * resample)
    return x, int(fs)


def save_audio(path, waveform, sample_rate):
    """utils.save_audio (utils.py:38-42): 16-bit PCM"""
    from scipy.io import wavfile
    w = waveform.detach().float().cpu()
    assert w.dim() <= 2, "waveform dimensions are too much ! (no more than 2)"
    if w.dim() == 1:
        w = w.unsqueeze(0)
    pcm = (w.clamp(-1.0, 32767.0 / 32768.0) * 32768.0).round().to(torch.int16).numpy().T
    wavfile.write(path, int(sample_rate), pcm[:, 0] if pcm.shape[1] == 1 else pcm)


def normalize_audio(waveform, dim=-1):
    """process.normalize_audio (process.py:54-55)"""
    return waveform / waveform.abs().max(dim=dim, keepdim=True)[0]


def infer(argv=None):
    args = argument_handler(argv)
    if args.use_cpu or not torch.cuda.is_available():
        raise RuntimeError("fqss_amd infers on ROCm devices only (no CPU fallback; oracle/ is the CPU checker)")
    device = "cuda"
    with open(args.yml_path) as f:
        conf = yaml.safe_load(f)
    work_dir = conf["work_dir"]
    os.makedirs(work_dir, exist_ok=True)
    model_cfg = conf["model_cfg"]
    model = create_pretrained_model(model_cfg)
    enable_observer(model, False)
    model.to(device).eval()
    run = InferRunner(model)                    # the serving form: codes-only forward, one hipGraph per chunk shape
    dataset_cfg, testing_cfg = conf["dataset_cfg"], conf.get("testing_cfg", {})
    wav, fs = read_audio(args.audio_path, resample=dataset_cfg.get("resample", 1), device=device)
    ref_mean, ref_std = 0, 1
    if args.normalize:
        ref_mean, ref_std = wav.mean(), wav.std()
        wav = (wav - ref_mean) / ref_std
    sep_wav = model_infer(run, wav, n_srcs=model_cfg.get("n_src", 1), segment=testing_cfg.get("segment_samples", None),
                          overlap=testing_cfg.get("overlap", 0.25), device=device)
    if args.normalize:
        sep_wav = sep_wav * ref_std + ref_mean
    paths = []
    for src in range(model_cfg["n_src"]):
        save_path = os.path.join(work_dir, "output" + str(src) + ".wav")
        save_audio(save_path, normalize_audio(sep_wav[src, ...]), sample_rate=fs)
        print("output" + str(src) + ".wav has been saved to {}".format(save_path))
        paths.append(save_path)
    if args.plot:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        w = sep_wav[0].detach().cpu().reshape(-1, sep_wav.shape[-1])
        fig, axes = plt.subplots(w.shape[0], 1)
        for c, ax in enumerate([axes] if w.shape[0] == 1 else axes):
            ax.plot(torch.arange(w.shape[1]) / fs, w[c], linewidth=1)
            ax.grid(True)
            ax.set_xlabel("Time[sec]")
        save_path = os.path.join(work_dir, "waveform.png")
        fig.savefig(save_path)
        print("Waveform has been saved to {}".format(save_path))
    return paths


if __name__ == "__main__":
    infer()
