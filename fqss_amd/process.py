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
