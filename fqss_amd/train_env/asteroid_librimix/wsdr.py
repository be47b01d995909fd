"""SDR losses of the asteroid env on MI355X (reference: train_env/asteroid_librimix/wsdr.py:46-95).

`PairwiseWSDR("sisdr")` + PIT over 2 speakers + the SDR-weighted KD objective are ONE fused HIP
kernel sequence (csrc/train_ops.hip, fqss_kd_loss); this module keeps the reference's names for the
pieces a training script touches."""
from ... import kernels as K
from ... import ops


class KDObjective:
    """loss, kd_loss_dB = KDObjective(kd_lambda)(est, fest, targets)   (mysystem.py:124-151)"""

    def __init__(self, kd_lambda=0.1):
        self.kd_lambda = kd_lambda

    def __call__(self, est, fest, targets):
        loss, kd_db, w, sisdr = ops.KDLoss.apply(est, fest, targets, self.kd_lambda)
        return loss, kd_db, w, sisdr


def si_sdr(est, targets):
    """mean best-permutation SI-SDR in dB of a [B,2,T] estimate (no gradient)"""
    out, w, sisdr, _ = K.kd_loss(est.detach(), targets, targets, 0.0, want_grad=False)
    return sisdr
