#!/usr/bin/env python3
"""`val.py -y cfg.yaml` (reference: val.py:17-25, 59-92, 184-226): load the quantized model, run `model_infer` over the evaluation
set and report SI-SDR and its improvement over the unprocessed mixture.  `dataset_cfg.name: synthetic` evaluates seeded
two-speaker mixtures (the LibriMix / MUSDB file readers and the SDR / STOI metrics are the reference's CPU side)."""
import argparse

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
    if dataset_cfg["name"] != "synthetic":
        raise NotImplementedError(f"dataset {dataset_cfg['name']}: the audio file readers are the reference's CPU data side")
    sisnr, imp = val_synthetic(model, model_cfg, dataset_cfg, testing_cfg, "cuda")
    print("SI-SDR={:0.2f},SI-SDR-imp={:0.2f}".format(sisnr, imp))
    return sisnr, imp


if __name__ == "__main__":
    val()
