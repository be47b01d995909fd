"""`train(yml_path, local_rank, distributed_launch, device)` of the speechbrain env (reference:
train_env/speechbrain_librimix/speechbrain_librimix_trainer.py:575-700 and Separation.fit_batch :121-200), MI355X edition,
without the speechbrain / hyperpyyaml dependencies.

Same YAML keys as the reference's configs/sepformer_2spks_8k.yaml (work_dir, model_cfg{,.quantization}, dataset_cfg, N_epochs,
batch_size, lr, clip_grad_norm, loss_upper_lim, training_signal_len, seed, kd_lambda, threshold_byloss, threshold, lr_scheduler);
`!ref <key>` references are resolved, `!new:` / `!name:` object tags (loggers, augmenters, speechbrain callables) are read as
plain mappings and ignored.  Step semantics kept:
  * KD objective per sample, then the mean over the batch (:99-115): `KDTrainStep(loss="sisdr_pit_per_sample")`, the per-sample form of
    the fused KD-loss kernel (fqss_kd_loss_per_sample).  Per-GPU batch 1 (shipped) or 2: the reference broadcasts the KD weights as
    `[1, n_src, n_src] * [1, B]`, which only exists for B in (1, n_src) -- larger batches are refused here as they fail there;
  * `threshold_byloss`: `loss[loss > th]` -- samples at or below the threshold leave the batch mean; an empty selection leaves the
    loss as it is (:141-149), which at batch 1 makes the option a no-op; the kernel applies it on the device;
  * a step whose loss is not below `loss_upper_lim` (NaN / inf) is skipped and counted (:157-176); under data parallelism the
    decision is taken collectively so that every rank applies or skips the same update;
  * clip_grad_norm, Adam(lr), `ReduceLROnPlateau(factor, patience, dont_halve_until_epoch)` on the validation loss (restated
    from speechbrain 0.5.14's published behaviour: third-party, parity unpinned).
Data: `dataset_cfg.name: librimix` is the reference's configuration (configs/sepformer_2spks_8k.yaml:27-39): `prepare_librimix` writes
the CSVs into `save_folder`, `SbLibriMix` (prepare_data.py) serves whole utterances with the train-time speed perturbation / re-mix /
random cut of `compute_forward` on the device, a batch ahead of the step (fqss_amd/loader.py); `synthetic`: seeded 2-speaker mixtures."""
import json
import os
import re

import torch
import yaml

from ...data import synth_batch
from ...loader import Prefetcher, epoch_batches
from ...parallel import Comm
from ...quantization.qat.models.load_model import create_model, quantize_model
from ...runtime import KDTrainStep
from ...utils import set_seed
from ..asteroid_librimix.wsdr import si_sdr


class _Loader(yaml.SafeLoader):
    pass


def _tagged(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        return loader.construct_mapping(node, deep=True)
    if isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    return loader.construct_scalar(node)


_Loader.add_multi_constructor("!new:", _tagged)
_Loader.add_multi_constructor("!name:", _tagged)
_Loader.add_multi_constructor("!apply:", _tagged)
_Loader.add_constructor("!ref", lambda loader, node: "!ref " + loader.construct_scalar(node))
_Loader.add_constructor("!PLACEHOLDER", lambda loader, node: None)
_REF = re.compile(r"<([A-Za-z_0-9\[\]]+)>")


def _lookup(conf, path):
    cur = conf
    for tok in re.findall(r"[A-Za-z_0-9]+", path):
        cur = cur[tok]
    return cur


def _resolve(conf, v, depth=0):
    if isinstance(v, dict):
        return {k: _resolve(conf, x, depth) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve(conf, x, depth) for x in v]
    if isinstance(v, str) and v.startswith("!ref ") and depth < 8:
        body = v[5:].strip()
        m = _REF.fullmatch(body)
        if m:
            return _resolve(conf, _lookup(conf, m.group(1)), depth + 1)
        return _REF.sub(lambda mm: str(_resolve(conf, _lookup(conf, mm.group(1)), depth + 1)), body)
    return v


def load_hparams(yml_path):
    with open(yml_path) as f:
        conf = yaml.load(f, Loader=_Loader)
    return _resolve(conf, conf)


class ReduceLROnPlateau:
    """speechbrain.nnet.schedulers.ReduceLROnPlateau restated: halve after `patience` epochs without improvement, never
    before `dont_halve_until_epoch`"""

    def __init__(self, factor=0.5, patience=2, dont_halve_until_epoch=65, lr_min=1e-8):
        self.factor, self.patience, self.dont_halve_until_epoch, self.lr_min = factor, patience, dont_halve_until_epoch, lr_min
        self.patience_counter, self.losses, self.anchor = 0, [], 99999

    def __call__(self, lr, epoch, loss):
        new = lr
        if epoch > self.dont_halve_until_epoch:
            if loss <= self.anchor:
                self.patience_counter, self.anchor = 0, loss
            elif self.patience_counter < self.patience:
                self.patience_counter += 1
            else:
                new, self.patience_counter = max(lr * self.factor, self.lr_min), 0
        else:
            self.anchor = min(self.anchor, loss)
        self.losses.append(loss)
        return new


class _Data:
    """train / valid batches per epoch.  librimix: `dataio_prep` + `fit()`'s loaders (:482-573, 660-668; dataloader_opts: batch_size,
    no shuffle key -> speechbrain's train loader keeps file order unless `shuffle` is set; DistributedSampler shares under DDP)"""

    def __init__(self, hp, comm, device):
        self.hp, self.comm, self.device = hp, comm, device
        ds = hp["dataset_cfg"]
        self.synthetic = ds.get("name") == "synthetic"
        self.loader_wait_s = 0.0
        if self.synthetic:
            return
        if ds.get("name") != "librimix":
            raise NotImplementedError(f"speechbrain env: dataset_cfg.name {ds.get('name')!r} (librimix | synthetic)")
        from .prepare_data import SbLibriMix, prepare_librimix
        n_spks, noisy = int(hp.get("num_spks", hp["model_cfg"].get("n_src", 2))), bool(ds.get("noisy", False))
        save = hp.get("save_folder") or os.path.join(hp["work_dir"], "save")
        if comm.rank == 0:                      # run_on_main(prepare_librimix, ...) (:601-610)
            prepare_librimix(ds["data_folder"], save, n_spks=n_spks, skip_prep=ds.get("skip_prep", False), librimix_addnoise=noisy)
        comm.barrier()
        sp = hp.get("speedperturb") or {}
        common = dict(n_src=n_spks, sample_rate=ds.get("sample_rate", 16000), resample=ds.get("resample", 1), noisy=noisy, device=device,
                      data_root=ds["data_folder"])
        self.sets = {
            "train": SbLibriMix(os.path.join(save, f"libri{n_spks}mix_train-360.csv"), train=True, speeds=sp.get("speeds", (95, 100, 105)),
                                use_speedperturb=bool(hp.get("use_speedperturb", False)), perturb_prob=sp.get("perturb_prob", 1.0),
                                limit_training_signal_len=bool(hp.get("limit_training_signal_len", False)),
                                training_signal_len=int(hp.get("training_signal_len", 32000)), **common),
            "val": SbLibriMix(os.path.join(save, f"libri{n_spks}mix_dev.csv"), **common)}
        for k in ("use_wavedrop", "use_rand_shift"):
            if hp.get(k):
                raise NotImplementedError(f"speechbrain env: {k} (off in every shipped config) has no device kernel")

    def batches(self, split, epoch):
        hp, comm = self.hp, self.comm
        ds = hp["dataset_cfg"]
        if self.synthetic:
            T = int(hp.get("training_signal_len", 32000))
            n = int(ds.get("steps_per_epoch" if split == "train" else "val_steps", 20 if split == "train" else 4))
            for i in range(n):
                yield synth_batch(int(hp["batch_size"]), T, seed=(0 if split == "train" else 10_000) + i * comm.world + comm.rank,
                                  device=self.device)
            return
        opts = hp.get("dataloader_opts") or {}
        bs = int(opts.get("batch_size", hp["batch_size"]))
        batches = epoch_batches(len(self.sets[split]), bs, shuffle=bool(opts.get("shuffle", False)) and split == "train", drop_last=False,
                                rank=comm.rank, world=comm.world, seed=hp.get("seed", 0), epoch=epoch)
        limit = ds.get("steps_per_epoch" if split == "train" else "val_steps")        # optional cap (smoke runs), not a reference key
        if limit is not None:
            batches = batches[:int(limit)]
        pf = Prefetcher(self.sets[split], batches, self.device, depth=int(hp.get("prefetch_depth", 2)))
        yield from pf
        self.loader_wait_s += pf.wait_s


def train(yml_path, local_rank=0, distributed_launch=False, device="cuda"):
    if device == "cpu":
        raise RuntimeError("fqss_amd is MI355X-only (no CPU fallback); the CPU checker lives in oracle/")
    hp = load_hparams(yml_path)
    if int(hp.get("batch_size", 1)) not in (1, 2):
        raise ValueError("speechbrain env: per-GPU batch_size must be 1 or n_src = 2 (the reference's KD weights broadcast "
                         "[1, n_src, n_src] * [1, B] and raise for any other batch)")
    set_seed(hp.get("seed", 0))
    comm = Comm.from_env("cuda")
    dev = torch.device("cuda", comm.local_rank)
    torch.cuda.set_device(dev)
    model_cfg = hp["model_cfg"]
    model = create_model(model_cfg)
    kd_lambda = float(hp.get("kd_lambda", 0))
    if kd_lambda <= 0:
        raise NotImplementedError("speechbrain env: the QAT path is the KD step (kd_lambda > 0)")
    import copy
    fmodel = copy.deepcopy(model).to(dev).eval()            # float teacher BEFORE quantization (:625-629)
    model = quantize_model(model, model_cfg["quantization"]).to(dev).train()
    work_dir = hp["work_dir"]
    if comm.rank == 0:
        os.makedirs(os.path.join(work_dir, "save"), exist_ok=True)
    lr = float(hp.get("lr", 1.5e-4))
    clip = float(hp.get("clip_grad_norm", 5))
    upper = float(hp.get("loss_upper_lim", 999999))
    threshold = float(hp["threshold"]) if hp.get("threshold_byloss") and "threshold" in hp else None
    step = KDTrainStep(model, fmodel, kd_lambda=kd_lambda, lr=lr, clip=clip if clip >= 0 else 1e30, comm=comm,
                       loss="sisdr_pit_per_sample", loss_threshold=threshold)
    sch = hp.get("lr_scheduler") or {}
    sched = ReduceLROnPlateau(sch.get("factor", 0.5), sch.get("patience", 2), sch.get("dont_halve_until_epoch", 65))
    history, best, nonfinite = [], float("inf"), 0
    flag = torch.zeros(1, device=dev)
    data = _Data(hp, comm, dev)
    for epoch in range(1, int(hp["N_epochs"]) + 1):
        losses = []
        for x, tgt in data.batches("train", epoch):
            step.maybe_capture(x, tgt)        # quantizing phase: both halves of the step replay as hipGraphs
            # (an utterance shorter than training_signal_len gives a batch of another shape: that step runs eagerly)
            graphed = step._graphs is not None and x.shape == step._sx.shape
            step._maybe_sync_ranges()
            # fwd + loss + bwd with the gradient exchange of the data-parallel ranks (both forms all-reduce the flat gradient)
            r = step.replay_fwd_bwd(x, tgt) if graphed else step._step_eager(x, tgt)
            # (the hard threshold of easy items, threshold_byloss, is applied inside the loss kernel)
            flag.copy_((r["loss"] < upper).float().reshape(1))       # False for NaN / inf as well
            if comm.world > 1:
                comm.all_reduce_sum(flag)
            if flag.item() >= comm.world:        # the reference syncs here too (loss.detach().cpu(), :199)
                if graphed:
                    step.replay_optimize()
                else:
                    step._optimize()
                    step._eager_q += 1 if step.tables is not None else 0
                losses.append(r["loss"].item())
            else:
                nonfinite += 1
                print(f"Warning: Loss is {r['loss'].item()}, skipping this sample! nonfinite_count={nonfinite}")
        with torch.no_grad():
            val = torch.stack([-si_sdr(model(x), tgt).mean() for x, tgt in data.batches("val", epoch)]).mean().reshape(1)
        comm.all_reduce_sum(val)
        val = val.item() / comm.world
        new_lr = sched(step.lr, epoch, val)
        rec = {"epoch": epoch, "lr": step.lr, "train_loss": sum(losses) / max(1, len(losses)), "valid_si-snr": val,
               "launch": "hipGraph replay" if step._graphs is not None else "eager"}
        step.lr = new_lr
        history.append(rec)
        if comm.rank == 0:
            print(json.dumps(rec), flush=True)
            with open(os.path.join(work_dir, "train_log.txt"), "a") as f:
                f.write(json.dumps(rec) + "\n")
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            torch.save(sd, os.path.join(work_dir, "save", "latest_model.pth"))
            if val < best:
                best = val
                torch.save(sd, os.path.join(work_dir, "save", "best_model.pth"))
    comm.barrier()
    return history
