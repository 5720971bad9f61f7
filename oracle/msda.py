"""TEST INFRASTRUCTURE ONLY — Python handle on the plain-C K1 oracle (oracle/msda_ref.c).

Pinned (tests/test_oracle_msda.py) against golden vectors produced by the reference's own
``ms_deform_attn_core_pytorch`` (ops/functions/ms_deform_attn_func.py:52-72) on the fixture of
ops/test.py:24-39 and on model-shaped cases (tests/golden/msda_*.npz, oracle/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def msda_forward(value, shapes, lsi, loc, attw):
    """value [B,S,M,D], shapes [L,2] i64, lsi [L] i64, loc [B,Lq,M,L,P,2], attw [B,Lq,M,L,P] -> [B,Lq,M*D]."""
    value = np.ascontiguousarray(value)
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attw = np.ascontiguousarray(attw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((B, Lq, M * D), dtype=dt)
    fn = getattr(_load(), "oracle_msda_forward_f32" if dt == np.float32 else "oracle_msda_forward_f64")
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    fn(p(value), p(shapes), p(lsi), p(loc), p(attw), p(out), B, S, M, D, L, Lq, P)
    return out
