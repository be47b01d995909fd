"""Training system of the asteroid env without the PyTorch-Lightning dependency.

Keeps the step semantics of the reference's System (train_env/asteroid_librimix/mysystem.py):
`common_step(batch, batch_nb, train)` -> (loss, kd_loss) with the SDR-weighted KD objective when
`train and kd_lambda > 0`, the plain PIT SI-SDR loss otherwise; `training_step` / `validation_step`.
The fast path used by the trainer is fqss_amd.runtime.KDTrainStep (flat-arena Adam, hipGraph)."""
import torch

from ...runtime import KDTrainStep
from .wsdr import KDObjective, si_sdr


class System:
    default_monitor = "val_loss"

    def __init__(self, model, fmodel, kd_lambda, lr=1e-3, clip=5.0, comm=None, betas=(0.9, 0.999), teacher_ahead=False):
        # kd_lambda = 0: the reference trains on the plain PIT SI-SDR loss without the teacher (mysystem.py:153-156); KDTrainStep then
        # runs fqss_pit_sisdr_loss and never calls the teacher
        self.model = model
        self.fmodel = fmodel
        self.kd_lambda = kd_lambda
        self.objective = KDObjective(kd_lambda)
        self.stepper = KDTrainStep(model, fmodel, kd_lambda=kd_lambda, lr=lr, clip=clip, comm=comm, betas=betas,
                                   teacher_ahead=teacher_ahead)
        self.logged = {}

    def forward(self, *a, **k):
        return self.model(*a, **k)

    __call__ = forward

    def common_step(self, batch, batch_nb, train=True):
        inputs, targets = batch
        est = self(inputs)
        if train and self.kd_lambda > 0:
            with torch.no_grad():
                fest = self.fmodel(inputs)
            loss, kd_db, _, _ = self.objective(est, fest, targets)
            return loss, kd_db
        return -si_sdr(est, targets).mean(), 0

    def training_step(self, batch, batch_nb, x_next=None):
        """one optimiser step through the fused runtime; logs `loss` / `kd_loss` like the reference.  x_next: the mixture of the
        NEXT call (a prefetching loader has it on the device already): the teacher starts on it beside this step"""
        self.stepper.maybe_capture(*batch)      # quantizing phase: the step replays as hipGraphs from here on
        r = self.stepper(*batch, x_next=x_next if self.stepper.teacher_ahead else None)
        self.logged.update(loss=r["loss"], kd_loss=r["kd_loss"])
        return r["loss"]

    def validation_step(self, batch, batch_nb):
        with torch.no_grad():
            loss, _ = self.common_step(batch, batch_nb, train=False)
        self.logged.update(val_loss=loss)
        return loss
