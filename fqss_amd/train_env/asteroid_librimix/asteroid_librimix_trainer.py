"""`train(yml_path, device)` of the asteroid env (reference: asteroid_librimix_trainer.py:140-214),
MI355X edition: same YAML keys (work_dir, model_cfg{,.quantization}, dataset_cfg, training_cfg), same
outputs (conf.yml, latest_model.pth, best_model.pth = student state_dict), data-parallel over the
GPUs of one node when launched with torch.distributed.run (one process per GPU, RCCL).

Data: `dataset_cfg.name: librimix` is the reference's own configuration (configs/convtasnet_2spks_8k.yaml:27-41): `prepare_datasets`
builds the LibriMix CSV datasets exactly as asteroid_librimix_trainer.py:26-75 does and feeds them through a background reader
(fqss_amd/loader.py: WAV reads, upload, resampling and SNR mixing of batch n+1 on a side stream while step n replays).
`dataset_cfg.name: synthetic` generates the seeded 2-speaker mixtures of the measurement contract."""
import json
import os

import torch
import yaml

from ...data import synth_batch
from ...loader import Prefetcher, epoch_batches, with_lookahead
from ...parallel import Comm
from ...utils import set_seed
from ..train_utils import create_pretrained_model
from .mysystem import System


LAST_SYSTEM = None


def prepare_datasets(dataset_cfg, training_cfg, device):
    """train / validation LibriMix sets with the reference's arguments (asteroid_librimix_trainer.py:26-52)"""
    from .librimix_dataset import LibriMix
    augmentation_cfg = dataset_cfg.get("augmentation", None)
    if augmentation_cfg and not augmentation_cfg.get("enable", False):
        augmentation_cfg = None
    common = dict(task=dataset_cfg["task"], sample_rate=dataset_cfg["sample_rate"], resample=dataset_cfg.get("resample", 1),
                  n_src=dataset_cfg["n_src"], segment=dataset_cfg["segment"], device=device)
    train_set = LibriMix(csv_dir=dataset_cfg["train_dir"], augmentation_cfg=augmentation_cfg, **common)
    val_set = LibriMix(csv_dir=dataset_cfg["valid_dir"], **common)
    print("Training set size: {}".format(len(train_set)))
    print("Validation set size: {}".format(len(val_set)))
    return train_set, val_set


class _Data:
    """the two loaders of prepare_datasets (:53-67): shuffled / sequential batches of training_cfg.batch_size, drop_last; under DDP
    every rank reads its DistributedSampler share (Lightning's replacement sampler).  `train(epoch)` yields (x, tgt, x_next)."""

    def __init__(self, dataset_cfg, training_cfg, comm, device):
        self.cfg, self.tc, self.comm, self.device = dataset_cfg, training_cfg, comm, device
        self.synthetic = dataset_cfg.get("name") == "synthetic"
        self.loader_wait_s = 0.0
        if self.synthetic:
            return
        if dataset_cfg.get("name") != "librimix":
            raise NotImplementedError(f"asteroid env: dataset_cfg.name {dataset_cfg.get('name')!r} (librimix | synthetic)")
        self.train_set, self.val_set = prepare_datasets(dataset_cfg, training_cfg, device)

    def _synthetic(self, split):
        ds, comm = self.cfg, self.comm
        sr = int(ds.get("sample_rate", 16000) * ds.get("resample", 0.5))
        T = int(ds.get("segment", 3) * sr)
        n = int(ds.get("steps_per_epoch" if split == "train" else "val_steps", 20 if split == "train" else 4))
        for i in range(n):
            yield synth_batch(self.tc["batch_size"], T, seed=(0 if split == "train" else 10_000) + i * comm.world + comm.rank,
                              device=self.device)

    def _files(self, split, epoch):
        ds = self.train_set if split == "train" else self.val_set
        batches = epoch_batches(len(ds), self.tc["batch_size"], shuffle=split == "train", drop_last=True, rank=self.comm.rank,
                                world=self.comm.world, seed=self.tc.get("seed", 0), epoch=epoch)
        limit = self.cfg.get("steps_per_epoch" if split == "train" else "val_steps")      # optional cap (smoke runs), not a reference key
        if limit is not None:
            batches = batches[:int(limit)]
        pf = Prefetcher(ds, batches, self.device, depth=int(self.tc.get("prefetch_depth", 2)))
        yield from pf
        self.loader_wait_s += pf.wait_s

    def train(self, epoch):
        return with_lookahead(self._synthetic("train") if self.synthetic else self._files("train", epoch))

    def val(self, epoch):
        return self._synthetic("val") if self.synthetic else self._files("val", epoch)


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
    data = _Data(dataset_cfg, training_cfg, comm, dev)
    # the loader's next mixture is on the device a step early: the frozen teacher runs on it beside the current step (KDTrainStep
    # teacher_ahead: the path bench.py measures; HIP backend only)
    ahead = (not cpu) and training_cfg.get("kd_lambda", 0) > 0 and bool(training_cfg.get("teacher_ahead", True))
    system = System(model, fmodel, training_cfg.get("kd_lambda", 0), lr=opt.get("lr", 1e-3), clip=5.0, comm=comm, betas=betas,
                    teacher_ahead=ahead)
    global LAST_SYSTEM
    LAST_SYSTEM = system                # tools / tests: the stepper of the run that just finished (its graphs, its model)
    best, history, since_best = float("inf"), [], 0
    # schedulers of train_setup (asteroid_librimix_trainer.py:96-102): StepLR for `step_lr` (DPTNet config), ReduceLROnPlateau
    # (factor 0.5) for `half_lr`; both act once per epoch on the stepper's learning rate
    step_lr, half_lr = training_cfg.get("step_lr"), training_cfg.get("half_lr", False)
    base_lr, plateau_best, plateau_bad = opt.get("lr", 1e-3), float("inf"), 0
    for epoch in range(training_cfg["epochs"]):
        import time
        t_epoch, n_steps = time.perf_counter(), 0
        for i, (x, tgt, x_next) in enumerate(data.train(epoch)):
            system.training_step((x, tgt), i, x_next=x_next)
            n_steps += 1
        if not cpu:
            torch.cuda.synchronize()
        t_epoch = time.perf_counter() - t_epoch
        val = torch.stack([system.validation_step(b, i) for i, b in enumerate(data.val(epoch))]).mean()
        comm.all_reduce_sum(val)
        val = val.item() / comm.world
        history.append({"epoch": epoch, "loss": system.logged["loss"].item(), "val_loss": val, "lr": system.stepper.lr,
                        "train_ms_per_step": 1e3 * t_epoch / max(1, n_steps), "loader_wait_s": data.loader_wait_s,
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
                torch.save(sd, os.path.join(work_dir, "best_model.pth"))
        # EarlyStopping(monitor="val_loss", mode="min", patience=30) of train_setup (:117-118); every rank sees the same reduced `val`
        since_best = 0 if val < best else since_best + 1
        best = min(best, val)
        if training_cfg.get("early_stop", False) and since_best >= 30:
            if comm.rank == 0:
                print(f"Early stopping: val_loss has not improved for {since_best} epochs")
            break
    comm.barrier()
    return history
