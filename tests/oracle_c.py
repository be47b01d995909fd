"""ctypes binding of oracle/libfq_oracle.so (C restatement of the quantizer core). Test-only."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_SO = os.path.join(_DIR, "libfq_oracle.so")


def _load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_DIR, "fq_core.c")):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return C.CDLL(_SO)


_lib = _load()
_f = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def act_fwd(x, lo, hi):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    idx = np.empty(x.shape, np.uint8)
    _lib.fqo_act_fwd(_p(x, C.c_float), C.c_int64(x.size), C.c_float(lo), C.c_float(hi), _p(y, C.c_float), _p(idx, C.c_uint8))
    return y, idx


def act_bwd(x, g, lo, hi):
    x = np.ascontiguousarray(x, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    gx = np.empty_like(x)
    gmin, gmax = C.c_double(), C.c_double()
    _lib.fqo_act_bwd(_p(x, C.c_float), _p(g, C.c_float), C.c_int64(x.size), C.c_float(lo), C.c_float(hi),
                     _p(gx, C.c_float), C.byref(gmin), C.byref(gmax))
    return gx, gmin.value, gmax.value


def _w_layout(shape, axis):
    outer = int(np.prod(shape[:axis])) if axis > 0 else 1
    ch = shape[axis]
    inner = int(np.prod(shape[axis + 1:]))
    return outer, ch, inner


def w_fwd(w, lo, hi, axis):
    w = np.ascontiguousarray(w, np.float32)
    o, c, i = _w_layout(w.shape, axis)
    lo = np.ascontiguousarray(lo, np.float32).ravel()
    hi = np.ascontiguousarray(hi, np.float32).ravel()
    y = np.empty_like(w)
    idx = np.empty(w.shape, np.int8)
    _lib.fqo_w_fwd(_p(w, C.c_float), C.c_int64(o), C.c_int64(c), C.c_int64(i), _p(lo, C.c_float), _p(hi, C.c_float),
                   _p(y, C.c_float), _p(idx, C.c_int8))
    return y, idx


def w_bwd(w, g, lo, hi, axis):
    w = np.ascontiguousarray(w, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    o, c, i = _w_layout(w.shape, axis)
    lo = np.ascontiguousarray(lo, np.float32).ravel()
    hi = np.ascontiguousarray(hi, np.float32).ravel()
    gw = np.empty_like(w)
    gmin = np.empty(c, np.float32)
    gmax = np.empty(c, np.float32)
    _lib.fqo_w_bwd(_p(w, C.c_float), _p(g, C.c_float), C.c_int64(o), C.c_int64(c), C.c_int64(i), _p(lo, C.c_float),
                   _p(hi, C.c_float), _p(gw, C.c_float), _p(gmin, C.c_float), _p(gmax, C.c_float))
    return gw, gmin, gmax


def splitter2(x):
    x = np.ascontiguousarray(x, np.float32)
    B, T = x.shape[0], x.shape[-1]
    out = np.empty((B, 2, T), np.float32)
    _lib.fqo_splitter2(_p(x, C.c_float), C.c_int64(B), C.c_int64(T), _p(out, C.c_float))
    return out


def combine2(x0, x1):
    x0 = np.ascontiguousarray(x0, np.float32)
    x1 = np.ascontiguousarray(x1, np.float32)
    y = np.empty_like(x0)
    _lib.fqo_combine2(_p(x0, C.c_float), _p(x1, C.c_float), C.c_int64(x0.size), _p(y, C.c_float))
    return y
