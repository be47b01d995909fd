"""8-bit input splitter / output combiner on MI355X (reference: process.py:10-52).

`preprocess` = global max-normalise + MSB/LSB floor-quantised channels (HIP: fqss_minmax + fqss_splitter2),
`postprocess` = `x0 + x1 * 2^-8` (HIP: fqss_axpby).  The evaluation helpers of the reference's process.py
(torchmetrics SI-SNR/SDR/STOI, chunked overlap-add inference) are the "next" rows of SURVEY.md §8(f).
"""
import torch

from . import kernels as K
from . import ops


def quantize(x, threshold=1.0, n_bits=8, sign=True):
    """floor quantizer of the splitter (process.py:10-14); served by the splitter kernel only"""
    raise NotImplementedError("process.quantize is fused inside fqss_splitter2; call preprocess(n_splitter=2)")


def preprocess(x, n_splitter=1, n_bits=8, sign=True, normalize=True):
    if x.dim() == 2:
        x = x.unsqueeze(1)
    if n_splitter <= 1:
        return x
    if n_splitter != 2 or n_bits != 8 or not sign:
        raise NotImplementedError("the splitter kernel serves n_splitter=2, 8 bit, signed")
    if normalize and x.shape[1] == 1 and x.dim() == 3:
        return ops.splitter2(x)
    # multi-channel / multi-dimensional inputs (HTDemucs: [B, A, Fr, T] spectrogram, [B, A, T] waveform): the flattened
    # [B, 2, A*...] result is torch.cat([msb, lsb], dim=1)
    with torch.no_grad():
        y = K.splitter2(x, normalize=normalize)
    return y.view(x.shape[0], 2 * x.shape[1], *x.shape[2:])


def postprocess(x, n_combiner=1, n_bits=8, sign=True):
    """x: [n_combiner, batch, sources, audio_channels, T]"""
    if n_combiner == 1:
        y = x.squeeze(0)
    elif n_combiner == 2 and n_bits == 8 and sign:
        y = ops.Combine2.apply(x[0], x[1])
    else:
        raise NotImplementedError("the combiner kernel serves n_combiner in {1,2}, 8 bit, signed")
    if y.dim() <= 4 and y.shape[-2] == 1:
        y = y.squeeze(-2)
    return y


# ---------------------------------------------------------------------------------------------------------------------------
# evaluation side: process.py:105-194 of the reference on the device
# ---------------------------------------------------------------------------------------------------------------------------
def si_snr(est, ref):
    """SI-SNR in dB between matching rows of est / ref [S, L] (torchmetrics' ScaleInvariantSignalNoiseRatio, third party:
    zero-mean SI-SDR with eps = 2^-23, restated; process.py:119, 137)"""
    return torch.diagonal(K.sisnr_matrix(est.reshape(-1, est.shape[-1]), ref.reshape(-1, ref.shape[-1])))


def swap_channel_order(sep_tensor, clean_tensor):
    """process.swap_channel_order (process.py:105-125): every estimate goes to the target it matches best (sign flipped when it
    moves); the decision is taken on the device (fqss_sisnr_matrix), the copy is the weighted overlap-add's with weight 1"""
    n_src = clean_tensor.shape[0]
    if n_src == 1:
        return sep_tensor
    L = sep_tensor.shape[-1]
    sep2 = sep_tensor.reshape(n_src, -1).contiguous()
    _, mp = K.sisnr_matrix(sep2, clean_tensor.reshape(n_src, -1), want_map=True)
    idx = mp[:, 0].long()
    out = sep2[idx] * mp[:, 1:2].to(sep2.dtype)        # (a gather of S rows and a sign: no arithmetic on the samples)
    return out.reshape(sep_tensor.shape)


def model_infer(model, mix, n_srcs=1, segment=None, overlap=0.25, device="cuda", target=None):
    """process.model_infer (process.py:156-194): whole-utterance inference, or chunks of `segment` samples hopped by
    (1 - overlap) * segment, re-ordered per chunk against `target` and blended with the triangular window.  mix [channels, length];
    returns [n_srcs, (channels,) length] on the device (the reference returns a CPU tensor)."""
    if str(device) == "cpu":
        raise RuntimeError("fqss_amd runs on ROCm devices only (oracle/ is the CPU checker)")
    mix = mix.to(device)
    if not segment:
        with torch.no_grad():
            out = model(mix.unsqueeze(0)).detach()[0]
        pad = mix.size(-1) - out.size(-1)
        return torch.nn.functional.pad(out, (0, pad)) if pad > 0 else out
    channels, length = mix.shape
    num_srcs = model.n_srcs if hasattr(model, "n_srcs") else n_srcs
    out = torch.zeros((num_srcs, channels, length) if channels > 1 else (num_srcs, length), device=mix.device)
    sum_weight = torch.zeros(length, device=mix.device)
    stride = int((1 - overlap) * segment)
    if target is not None:
        target = target.to(device)
    for start in range(0, length, stride):
        stop = min(start + segment, length)
        n = stop - start
        chunk = mix[..., start:stop]
        if n < segment:
            padded = torch.zeros(channels, segment, device=mix.device)
            padded[:, :n].copy_(chunk)
            chunk = padded
        chunk_out = model_infer(model, chunk, device=device)[..., :n].contiguous()
        mp = None
        if target is not None and num_srcs > 1:
            _, mp = K.sisnr_matrix(chunk_out.reshape(num_srcs, -1), target[..., start:start + n].reshape(num_srcs, -1), want_map=True)
        K.infer_ola(chunk_out, mp, out, sum_weight, start, n, segment)
    K.infer_normalize(out, sum_weight)
    return out


def metric_evaluation(sep_waveform, clean_waveforms, sample_rate=16000):
    """mean over sources of the best-match SI-SNR (process.py:127-154).  SDR (fast_bss_eval) and STOI (pystoi) are third-party CPU
    metrics outside the hot path: reported as NaN"""
    db = K.sisnr_matrix(sep_waveform.reshape(clean_waveforms.shape[0], -1), clean_waveforms.reshape(clean_waveforms.shape[0], -1))
    return db.max(dim=1).values.mean().item(), float("nan"), float("nan")


# ---------------------------------------------------------------------------------------------------------------------------
# data side: the SNR augmentation of the LibriMix dataset on the device, batched
# ---------------------------------------------------------------------------------------------------------------------------
def generate_2mix_snr(signal1, signal2, snr, clip=True):
    """process.generate_2mix_snr (process.py:77-91) for one pair [T] or a batch [B, T]; snr: a number or a [B] tensor (dB)"""
    one = signal1.dim() == 1
    a, b = signal1.reshape(-1, signal1.shape[-1]), signal2.reshape(-1, signal2.shape[-1])
    s = snr if torch.is_tensor(snr) else torch.full((a.shape[0],), float(snr), device=a.device)
    out = K.snr_mix(a, b, s.to(a.device, torch.float32), 0, clip)
    return out[0] if one else out


def generate_3mix_snr(signal1, signal2, signal3, snr1_23, snr2_3):
    return generate_2mix_snr(signal1, generate_2mix_snr(signal2, signal3, snr2_3), snr1_23)


def generate_mix_noise(sig, noise, snr):
    """process.generate_mix_noise (process.py:98-103)"""
    one = sig.dim() == 1
    a, b = sig.reshape(-1, sig.shape[-1]), noise.reshape(-1, noise.shape[-1])
    s = snr if torch.is_tensor(snr) else torch.full((a.shape[0],), float(snr), device=a.device)
    out = K.snr_mix(a, b, s.to(a.device, torch.float32), 1, True)
    return out[0] if one else out
