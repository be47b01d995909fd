"""`train(yml_path, device)` of the asteroid env (reference: asteroid_librimix_trainer.py:140-214),
MI355X edition: same YAML keys (work_dir, model_cfg{,.quantization}, dataset_cfg, training_cfg), same
outputs (conf.yml, latest_model.pth, best_model.pth = student state_dict), data-parallel over the
GPUs of one node when launched with torch.distributed.run (one process per GPU, RCCL).

Data: `dataset_cfg.name: synthetic` generates the seeded 2-speaker mixtures of the measurement
contract.  The LibriMix CSV/soundfile reader is the reference's CPU data side."""
import json
import os

import torch
import yaml

from ...data import synth_batch
from ...parallel import Comm
from ...utils import set_seed
from ..train_utils import create_pretrained_model
from .mysystem import System


def _batches(dataset_cfg, training_cfg, comm, device, split):
    if dataset_cfg.get("name") != "synthetic":
        raise NotImplementedError("only dataset_cfg.name == 'synthetic' is built in; the LibriMix reader is a later §8(f) row")
    sr = int(dataset_cfg.get("sample_rate", 16000) * dataset_cfg.get("resample", 0.5))
    T = int(dataset_cfg.get("segment", 3) * sr)
    n = int(dataset_cfg.get("steps_per_epoch" if split == "train" else "val_steps", 20 if split == "train" else 4))
    B = training_cfg["batch_size"]
    for i in range(n):
        yield synth_batch(B, T, seed=(0 if split == "train" else 10_000) + i * comm.world + comm.rank, device=device)


def train(yml_path, device):
    from ... import _lib
    prev = _lib.BACKEND
    try:
        return _train(yml_path, device)
    finally:
        if _lib.BACKEND != prev:          # `--use_cpu` switched the process to the CPU backend: hand it back as it was found
            _lib.set_backend(prev)


def _train(yml_path, device):
    cpu = device == "cpu"
    if cpu:
        # `--use_cpu` (reference train.py:31; BASELINE.json configs[0]: "CPU, batch 2, 1 s ... plumbing, no GPU"): the same trainer over
        # the CPU backend of the C ABI (csrc/cpu/libfqss_cpu.so: the un-fused per-layer entry points of the ConvTasNet step).  Chosen
        # explicitly here, never as a fallback; oracle/ is not involved.
        from ... import _lib
        _lib.set_backend("cpu")
    with open(yml_path) as f:
        conf = yaml.safe_load(f)
    work_dir, model_cfg, dataset_cfg = conf["work_dir"], conf["model_cfg"], conf["dataset_cfg"]
    training_cfg = conf["training_cfg"]
    set_seed(training_cfg.get("seed", 0))
    comm = Comm.from_env("cpu" if cpu else "cuda")
    dev = torch.device("cpu") if cpu else torch.device("cuda", comm.local_rank)
    if not cpu:
        torch.cuda.set_device(dev)
    elif model_cfg.get("name") != "ConvTasNet":
        raise NotImplementedError("--use_cpu: the CPU backend serves the ConvTasNet step (cfg 1 of BASELINE.json) only")

    model_cfg.update({"model_path": training_cfg.get("pretrained", None)})
    model, fmodel = create_pretrained_model(model_cfg)
    model.to(dev).train()
    fmodel.to(dev).eval()
    if comm.rank == 0:
        os.makedirs(work_dir, exist_ok=True)
        with open(os.path.join(work_dir, "conf.yml"), "w") as out:
            for blk in (model_cfg, dataset_cfg, training_cfg):
                yaml.safe_dump(blk, out)

    opt = training_cfg.get("optim", {})
    # make_optimizer(**training_cfg["optim"]) of the reference (asteroid_librimix_trainer.py:94): the fused clip + Adam kernel serves
    # Adam without weight decay (every shipped config); anything else is refused rather than silently ignored
    if str(opt.get("optimizer", "adam")).lower() != "adam" or float(opt.get("weight_decay", 0) or 0) != 0:
        raise NotImplementedError(f"asteroid env: optim {opt!r}: only adam with weight_decay 0 has a fused kernel")
    betas = tuple(opt.get("betas", (0.9, 0.999)))
    system = System(model, fmodel, training_cfg.get("kd_lambda", 0), lr=opt.get("lr", 1e-3), clip=5.0, comm=comm, betas=betas)
    best, history = float("inf"), []
    # schedulers of train_setup (asteroid_librimix_trainer.py:96-102): StepLR for `step_lr` (DPTNet config), ReduceLROnPlateau
    # (factor 0.5) for `half_lr`; both act once per epoch on the stepper's learning rate
    step_lr, half_lr = training_cfg.get("step_lr"), training_cfg.get("half_lr", False)
    base_lr, plateau_best, plateau_bad = opt.get("lr", 1e-3), float("inf"), 0
    for epoch in range(training_cfg["epochs"]):
        for i, batch in enumerate(_batches(dataset_cfg, training_cfg, comm, dev, "train")):
            system.training_step(batch, i)
        val = torch.stack([system.validation_step(b, i) for i, b in enumerate(_batches(dataset_cfg, training_cfg, comm, dev, "val"))]).mean()
        comm.all_reduce_sum(val)
        val = val.item() / comm.world
        history.append({"epoch": epoch, "loss": system.logged["loss"].item(), "val_loss": val, "lr": system.stepper.lr,
                        "launch": "hipGraph replay" if system.stepper._graphs is not None else "eager"})
        if half_lr:
            if val < plateau_best - 1e-4 * abs(plateau_best):     # torch ReduceLROnPlateau defaults: rel threshold 1e-4
                plateau_best, plateau_bad = val, 0
            else:
                plateau_bad += 1
                if plateau_bad > training_cfg.get("patience", 5):
                    system.stepper.lr, plateau_bad = system.stepper.lr * 0.5, 0
        elif step_lr is not None:
            system.stepper.lr = base_lr * step_lr.get("gamma", 0.98) ** ((epoch + 1) // step_lr.get("step_size", 2))
        if comm.rank == 0:
            print(json.dumps(history[-1]), flush=True)
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            torch.save(sd, os.path.join(work_dir, "latest_model.pth"))
            if val < best:
                best = val
                torch.save(sd, os.path.join(work_dir, "best_model.pth"))
    comm.barrier()
    return history
