"""Import shim for the upstream reference (THIS container only; never runs on the GPU box).

The reference at /root/reference imports third-party packages that are not installed here
(torchaudio, torchmetrics, demucs, julius, openunmix) at module load time
(process.py:3-4, utils.py:2, quantization/qat/models/load_model.py:6 -> htdemucsq.py:19-20).
We register empty stand-in modules so the *reference's own* code for the hot path
(quantization/qat/*, process.py, train_env/asteroid_librimix/wsdr.py) imports unchanged.
Nothing from the reference is copied; it is only imported to generate golden vectors.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if "/root/reference" in sys.path:
        return

    class _Dummy:  # metric classes are constructed lazily, never on the hot path
        def __init__(self, *a, **k):
            pass

    _stub("torchaudio")
    tm = _stub("torchmetrics", ScaleInvariantSignalNoiseRatio=_Dummy, SignalDistortionRatio=_Dummy)
    tma = _stub("torchmetrics.audio")
    tms = _stub("torchmetrics.audio.stoi", ShortTimeObjectiveIntelligibility=_Dummy)
    tm.audio = tma
    tma.stoi = tms
    _stub("julius")

    def capture_init(init):
        import functools

        @functools.wraps(init)
        def __init__(self, *args, **kwargs):
            self._init_args_kwargs = (args, kwargs)
            init(self, *args, **kwargs)

        return __init__

    d = _stub("demucs")
    d.states = _stub("demucs.states", capture_init=capture_init)
    d.spec = _stub("demucs.spec", spectro=None, ispectro=None)
    d.utils = _stub("demucs.utils", center_trim=None, unfold=None)
    ou = _stub("openunmix")
    ou.filtering = _stub("openunmix.filtering", wiener=None)
    sys.path.insert(0, REFERENCE_ROOT)
