"""`main()` of the htdemucs env (reference: train_env/htdemucs_musdbhq/train.py:163-262 `get_solver` / `main`, solver.py:38-429
`Solver`), MI355X edition, without the hydra / dora / demucs dependencies.

Plugin surface kept: `main()` takes no arguments and reads `key=value` / `+key=value` overrides from `sys.argv[1:]` the way the
reference's hydra entry point does (train.py:44-46 passes `+device=<device>`); the configuration is the reference's YAML layout
(`model_cfg{,.quantization}`, `dset`, `epochs`, `batch_size`, `optim{lr, beta2, loss, clip_grad, weight_decay}`, `weights`,
`kd_lambda`, `seed`, `test.metric`).  `+yml_path=<file>` selects the YAML (default: configs/htdemucs_synthetic.yaml); the
reference always reads configs/htdemucs.yaml through hydra.  Step semantics kept (solver.py:314-429):
  * train: mix = sum of the stems; estimate = student(mix); teacher under no_grad; loss = source-weighted
    (1 - kd_lambda) L1(est, src) + kd_lambda w L1(est, teacher), w = exp((sdr_teacher - sdr_student) / 10) per (sample, source);
    optional clip_grad; Adam(lr, betas (momentum, beta2)); global batch divided over the ranks (train.py:200-201), gradients
    averaged over ranks;
  * valid: eval mode on the validation stems, `reco` = source-weighted L1; the best state by `test.metric: loss` is written to
    `<work_dir>/best.th` as {"state", "kwargs", "history"} (the keys load_model.py:38-45, 76-102 read back).
Data: `dset.name: synthetic` (seeded Gaussian stems); MUSDB readers, demucs' augmentations (shift / flip / scale / remix /
repitch), EMA copies, `apply_model` split inference and the SDR evaluation are the reference's CPU / third-party data side."""
import json
import os
import sys

import torch
import yaml

from ...parallel import Comm, local_device
from ...quantization.qat.models.load_model import quantize_model
from ...quantization.qat.models.htdemucsq import HTDemucsQ
from ...runtime import KDTrainStep
from ...utils import set_seed
from ... import kernels as K

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def _set_path(conf, dotted, value):
    cur = conf
    keys = dotted.split(".")
    for k in keys[:-1]:
        cur = cur.setdefault(k, {})
    cur[keys[-1]] = value


def load_config(argv):
    """hydra-style overrides: `a.b=1`, `+a.b=1`; values are parsed as YAML scalars"""
    over = {}
    for tok in argv:
        if "=" not in tok:
            raise ValueError(f"override {tok!r}: expected key=value")
        k, v = tok.lstrip("+").split("=", 1)
        over[k] = yaml.safe_load(v)
    path = over.pop("yml_path", None) or os.path.join(ROOT, "configs", "htdemucs_synthetic.yaml")
    conf = yaml.safe_load(open(path))
    for k, v in over.items():
        _set_path(conf, k, v)
    return conf


def synth_stems(B, S, C, T, seed, device):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, S, C, T, generator=g) * 0.1).to(device)


class Solver:
    def __init__(self, conf, model, fmodel, comm, device):
        self.conf, self.model, self.fmodel, self.comm, self.device = conf, model, fmodel, comm, device
        opt = conf["optim"]
        if opt.get("optim", "adam") != "adam" or opt.get("loss", "l1") != "l1" or opt.get("weight_decay", 0):
            raise NotImplementedError("htdemucs env: Adam without weight decay on the L1 loss is the shipped configuration")
        if any(conf.get("ema", {}).get(k) for k in ("epoch", "batch")):
            raise NotImplementedError("htdemucs env: EMA copies are not built")
        self.weights = torch.tensor(conf.get("weights", [1.0] * model.n_srcs), device=device, dtype=torch.float32)
        self.step = KDTrainStep(model, fmodel, kd_lambda=float(conf.get("kd_lambda", 0.1)), lr=float(opt["lr"]),
                                clip=float(opt.get("clip_grad") or 0.0), comm=comm, loss="l1_sdr", source_weights=self.weights,
                                betas=(float(opt.get("momentum", 0.9)), float(opt.get("beta2", 0.999))))
        self.history = []
        self.best_state, self.best_loss = None, float("inf")

    def _batch(self, epoch, idx, train):
        d = self.conf["dset"]
        B = self.conf["batch_size"] // self.comm.world if train else 1
        T = int(round(d["segment"] * d["samplerate"]))
        seed = self.conf.get("seed", 42) * 1000003 + (epoch * 4096 + idx) * self.comm.world + self.comm.rank + (0 if train else 1 << 30)
        return synth_stems(B, self.model.n_srcs, d["channels"], T, seed, self.device)

    def _run_one_epoch(self, epoch, train=True):
        d = self.conf["dset"]
        n = d["steps_per_epoch"] if train else d["valid_steps"]
        tot = torch.zeros(1, device=self.device, dtype=torch.float64)
        self.model.train(train)
        for idx in range(n):
            sources = self._batch(epoch, idx, train)
            mix = sources.sum(dim=1)
            if train:
                self.step.maybe_capture(mix, sources)
                tot += self.step(mix, sources)["loss"].double()
            else:
                with torch.no_grad():
                    est = self.model(mix)
                    loss, task, _, _, _ = K.hd_kd_loss(est, est, sources, self.weights, 0.0, want_grad=False)    # kd_lambda 0: the L1 task loss
                tot += loss.double()
        self.comm.all_reduce_sum(tot)
        out = {"loss": tot.item() / (n * self.comm.world), "reco": tot.item() / (n * self.comm.world)}
        if train:
            out["launch"] = "hipGraph replay" if self.step._graphs is not None else "eager"
        return out

    def train(self):
        for epoch in range(self.conf["epochs"]):
            m = {"train": self._run_one_epoch(epoch), "valid": self._run_one_epoch(epoch, train=False)}
            self.model.train()
            key = self.conf.get("test", {}).get("metric", "loss")
            if key != "loss":
                raise NotImplementedError("htdemucs env: best-model selection by nsdr needs demucs' evaluation (third party)")
            if m["valid"]["loss"] <= self.best_loss:
                self.best_loss = m["valid"]["loss"]
                self.best_state = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
            m["valid"]["best"] = self.best_loss
            self.history.append(m)
            if self.comm.rank == 0:
                print(f"epoch {epoch + 1}: train loss {m['train']['loss']:.5f}  valid loss {m['valid']['loss']:.5f}  best {self.best_loss:.5f}",
                      flush=True)
                os.makedirs(self.conf["work_dir"], exist_ok=True)
                torch.save({"state": self.best_state, "kwargs": self.model._init_kwargs, "history": self.history},
                           os.path.join(self.conf["work_dir"], "best.th"))
                with open(os.path.join(self.conf["work_dir"], "history.json"), "w") as f:
                    json.dump(self.history, f)
        return self.history


def get_solver(conf):
    import copy
    device = conf.get("device", "cuda")
    if device != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("the htdemucs env of this build trains on ROCm devices only (no CPU fallback; oracle/ is the CPU checker)")
    comm = Comm.from_env("cuda")
    torch.cuda.set_device(local_device(comm.local_rank))
    dev = torch.device("cuda", torch.cuda.current_device())
    set_seed(conf.get("seed", 42))
    mc = conf["model_cfg"]
    kwargs = dict(sources=list(conf["dset"]["sources"]), audio_channels=conf["dset"]["channels"], samplerate=conf["dset"]["samplerate"],
                  segment=conf["dset"]["segment"])
    kwargs.update(conf.get("htdemucs") or {})
    model = HTDemucsQ(**kwargs)
    model._init_kwargs = kwargs
    if mc.get("model_path"):
        sd = torch.load(mc["model_path"], map_location="cpu", weights_only=False)   # trusted local package ({"state", "kwargs"})
        model.load_state_dict(sd.get("state", sd), strict=True)
    fmodel = copy.deepcopy(model).to(dev).eval() if float(conf.get("kd_lambda", 0.1)) > 0 else None
    if fmodel is None:
        raise NotImplementedError("htdemucs env: kd_lambda = 0 (no teacher) is not built; the shipped value is 0.1")
    model = quantize_model(model, mc["quantization"]).to(dev).train()
    model._init_kwargs = kwargs
    assert conf["batch_size"] % comm.world == 0
    return Solver(conf, model, fmodel, comm, dev)


def main():
    conf = load_config(sys.argv[1:])
    solver = get_solver(conf)
    hist = solver.train()
    solver.comm.close()
    return hist
