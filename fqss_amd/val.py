#!/usr/bin/env python3
"""`val.py -y cfg.yaml` (reference: val.py:17-25, 59-92, 184-226): load the quantized model, run `model_infer` over the evaluation
set and report SI-SDR and its improvement over the unprocessed mixture.  `dataset_cfg.name: librimix` walks `testing_cfg.test_dir`
as val.py:28-92 does (mix_clean | mix_both | mix_single, s1..s3; whole utterances, resampled on the device); `synthetic` evaluates
seeded two-speaker mixtures.  SDR (fast_bss_eval) and STOI (pystoi) are third-party CPU metrics: printed as nan.  MUSDB needs the
`musdb` package (absent): refused."""
import argparse
import glob
import os

import torch
import yaml

from . import kernels as K
from .data import synth_batch
from .process import metric_evaluation, model_infer
from .quantization.qat.models.load_model import create_pretrained_model, enable_observer


def argument_handler():
    p = argparse.ArgumentParser()
    p.add_argument("--yml_path", "-y", type=str, required=True, help="YML configuration file")
    p.add_argument("--use_cpu", action="store_true", help="Use cpu")
    return p.parse_args()


def val_synthetic(model, model_cfg, dataset_cfg, testing_cfg, device):
    n_srcs = model_cfg.get("n_src", 1)
    n, L = testing_cfg.get("n_items", 4), int(testing_cfg.get("length_samples", 32000))
    sisdr = sisdr_imp = 0.0
    for i in range(n):
        mix, clean = synth_batch(1, L, seed=10_000 + i, device=device)
        mix_wav, clean_wavs = mix[0], clean[0]                         # [1, L], [S, L]
        wavs = model_infer(model, mix_wav, n_srcs=n_srcs, segment=testing_cfg.get("segment_samples", None),
                           overlap=testing_cfg.get("overlap", 0.25), device=device, target=clean_wavs)
        s, _, _ = metric_evaluation(wavs, clean_wavs)
        base = K.sisnr_matrix(clean_wavs, mix_wav.expand(n_srcs, -1).contiguous())
        sisdr += s
        sisdr_imp += s - torch.diagonal(base).mean().item()
    return sisdr / n, sisdr_imp / n


def read_librimix(folder, n_spks=1, noisy=False):
    """file lists of a LibriMix test folder (val.py:28-57)"""
    assert 1 <= n_spks <= 3, "Error: Up to 3 sources to seperate!"
    mix_dir = "mix_single" if n_spks == 1 else ("mix_both" if noisy else "mix_clean")
    mix_audio_files = sorted(glob.glob(os.path.join(folder, mix_dir, "*")))
    clean = [sorted(glob.glob(os.path.join(folder, f"s{i + 1}", "*"))) for i in range(n_spks)]
    assert all(len(c) == len(mix_audio_files) for c in clean) and len(mix_audio_files) > 0, "Dataset is missing files!"
    return mix_audio_files, clean


def val_librimix(model, model_cfg, dataset_cfg, testing_cfg, device):
    """val.py:59-92: per utterance read mixture + sources, resample, model_infer, SI-SDR and its improvement over the mixture.  The
    reader thread has utterance i + 1 (WAV reads, upload, resampling) on the device while utterance i is separated."""
    from .loader import Prefetcher
    from .train_env.asteroid_librimix.librimix_dataset import read_wav
    n_srcs = model_cfg.get("n_src", 1)
    mix_files, clean_lists = read_librimix(testing_cfg["test_dir"], n_srcs, dataset_cfg["noisy"])
    limit = testing_cfg.get("n_items")
    n = len(mix_files) if limit is None else min(int(limit), len(mix_files))
    ratio = dataset_cfg.get("resample", 1)

    class _Utterances:                       # one item per batch: (mixture [1, 1, L'], sources [1, S, L'])
        def stage_elems(self, batch_size):
            return 0

        def batch(self, indices, stage=None):
            i = indices[0]
            clips = [read_wav(mix_files[i])] + [read_wav(lst[i]) for lst in clean_lists]
            L = min(len(c) for c in clips)
            import numpy as np
            x = torch.from_numpy(np.stack([c[:L] for c in clips])).to(device)
            if ratio != 1:
                fs = int(dataset_cfg.get("sample_rate", 16000))
                x = K.resample(x, fs, int(fs * ratio))
            return x[:1].unsqueeze(0), x[1:].unsqueeze(0)

    sisdr = sisdr_imp = 0.0
    for i, (mix, clean) in enumerate(Prefetcher(_Utterances(), [[k] for k in range(n)], device, depth=1)):
        mix_wav, clean_wavs = mix[0], clean[0]
        wavs = model_infer(model, mix_wav, n_srcs=n_srcs, segment=testing_cfg.get("segment_samples", None),
                           overlap=testing_cfg.get("overlap", 0.25), device=device, target=clean_wavs)
        s, _, _ = metric_evaluation(wavs, clean_wavs)
        base, _, _ = metric_evaluation(clean_wavs, mix_wav.expand(n_srcs, -1).contiguous())     # val.py:86: the sources as estimates, the mixture as target
        sisdr += s
        sisdr_imp += s - base
        if (i % 500 == 0 and i > 0) or i == 1:
            print("SI-SDR={:0.3f},SI-SDR-imp={:0.3f}".format(sisdr / (i + 1), sisdr_imp / (i + 1)))
    return sisdr / n, sisdr_imp / n


def val(argv=None):
    import sys
    if argv is not None:
        sys.argv[1:] = argv
    args = argument_handler()
    if args.use_cpu or not torch.cuda.is_available():
        raise RuntimeError("fqss_amd evaluates on ROCm devices only (no CPU fallback; oracle/ is the CPU checker)")
    conf = yaml.safe_load(open(args.yml_path))
    model_cfg = conf["model_cfg"]
    model = create_pretrained_model(model_cfg)
    enable_observer(model, False)
    model.to("cuda").eval()
    assert not (not model_cfg["quantization"].get("qat", False) and (model.n_splitter > 1 or model.n_splitter > 1)), \
        "No support for splitter/combiner with non QAT model."
    dataset_cfg, testing_cfg = conf["dataset_cfg"], conf.get("testing_cfg", {})
    if dataset_cfg["name"] == "librimix":
        sisnr, imp = val_librimix(model, model_cfg, dataset_cfg, testing_cfg, "cuda")
    elif dataset_cfg["name"] == "synthetic":
        sisnr, imp = val_synthetic(model, model_cfg, dataset_cfg, testing_cfg, "cuda")
    elif dataset_cfg["name"] == "musdbhq":
        raise NotImplementedError("dataset musdbhq: val.py:95-178 reads MUSDB18-HQ through the third-party `musdb` package (absent here)")
    else:
        assert False, "Dataset {} is not supported!".format(dataset_cfg["name"])
    print("SI-SDR={:0.2f},SI-SDR-imp={:0.2f},SDR={:0.2f},STOI={:0.3f}".format(sisnr, imp, float("nan"), float("nan")))
    return sisnr, imp


if __name__ == "__main__":
    val()
