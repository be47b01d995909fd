"""SDR losses of the asteroid env on MI355X (reference: train_env/asteroid_librimix/wsdr.py:46-95).

`PairwiseWSDR("sisdr")` + PIT over 2 speakers + the SDR-weighted KD objective are ONE fused HIP
kernel sequence (csrc/train_ops.hip, fqss_kd_loss); this module keeps the reference's names for the
pieces a training script touches."""
import torch

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


def _pair_moments(est, targets, zero_mean):
    """per sample: energy of estimate i, of target j, their dot product -- [B,2,1], [B,1,2], [B,2,2] fp64, from ONE streaming pass
    of the moment kernel over the waveforms (everything after it is arithmetic on 24 numbers per sample)"""
    T = est.shape[-1]
    m = K.kd_moments(est.detach(), targets.detach(), targets.detach())
    se, st = m[:, 0:2], m[:, 4:6]
    ee, tt, et = m[:, 6:8], m[:, 10:12], m[:, 12:16].reshape(-1, 2, 2)
    if zero_mean:
        ee = ee - se * se / T
        tt = tt - st * st / T
        et = et - se[:, :, None] * st[:, None, :] / T
    return ee[:, :, None], tt[:, None, :], et


class PairwiseWSDR:
    """PairwiseWSDR(sdr_type)(est_targets, targets, weights=None) -> [batch, n_src, n_src] (reference wsdr.py:46-95; entry [b, i, j]
    pairs estimate i with target j).  Evaluation form (no autograd: the training objective and its gradient are the fused
    fqss_kd_loss); n_src = 2, the moment kernel's size."""

    def __init__(self, sdr_type, zero_mean=True, take_log=True, EPS=1e-8):
        assert sdr_type in ["snr", "sisdr", "sdsdr"]
        self.sdr_type, self.zero_mean, self.take_log, self.EPS = sdr_type, zero_mean, take_log, EPS

    def forward(self, est_targets, targets, weights=None):
        if targets.size() != est_targets.size() or targets.ndim != 3:
            raise TypeError(f"Inputs must be of shape [batch, n_src, time], got {targets.size()} and {est_targets.size()} instead")
        ee, tt, et = _pair_moments(est_targets, targets, self.zero_mean)
        if self.sdr_type in ("sisdr", "sdsdr"):
            a = et / (tt + self.EPS)               # projection coefficient of estimate i on target j
            proj2 = a * a * tt
        else:
            a, proj2 = None, tt.expand_as(et)
        if self.sdr_type in ("sdsdr", "snr"):
            noise2 = ee - 2.0 * et + tt
        else:
            noise2 = ee - 2.0 * a * et + proj2
        sdr = proj2 / (noise2 + self.EPS)
        if weights is not None:
            sdr = sdr * weights[:, None, None]
        out = 10.0 * torch.log10(sdr + self.EPS) if self.take_log else -sdr
        return out.float()

    __call__ = forward


class SDR:
    """SDR(sdr_type)(est_targets, targets, weights=None) -> scalar (reference wsdr.py:10-43): estimate i against target i, mean over
    batch and sources before the optional log"""

    def __init__(self, sdr_type, zero_mean=True, take_log=True, EPS=1e-8):
        assert sdr_type in ["sisdr", "sdr"]
        self.sdr_type, self.zero_mean, self.take_log, self.EPS = sdr_type, zero_mean, take_log, EPS

    def forward(self, est_targets, targets, weights=None):
        assert targets.size() == est_targets.size()
        ee, tt, et = _pair_moments(est_targets, targets, self.zero_mean)
        ee, tt, et = ee[:, :, 0], tt[:, 0, :], torch.diagonal(et, dim1=1, dim2=2)
        a = et / (tt + self.EPS)
        proj2 = a * a * tt
        noise2 = ee - 2.0 * a * et + proj2 if self.sdr_type == "sisdr" else ee - 2.0 * et + tt
        sdr = proj2 / (noise2 + self.EPS)
        if weights is not None:
            # the reference multiplies its [B, n_src] ratios by weights[:, None, None] (wsdr.py:37): that BROADCASTS to [B, B, n_src]
            # -- every sample's ratio meets every sample's weight -- and the mean runs over all of it; reproduced as written
            sdr = sdr[None, :, :] * weights[:, None, None]
        sdr = torch.mean(sdr)
        return (10.0 * torch.log10(sdr + self.EPS) if self.take_log else sdr).float()

    __call__ = forward


# aliases (wsdr.py:98-102)
sisdr = SDR("sisdr", take_log=False)
sdr = SDR("sdr", take_log=False)
pairwise_wsisdr = PairwiseWSDR("sisdr", take_log=False)
pairwise_wsdsdr = PairwiseWSDR("sdsdr", take_log=False)
