"""ctypes access to oracle/liboracle.so (TEST INFRASTRUCTURE: the checker, never the thing measured)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        L = ctypes.CDLL(LIB)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        L.fxo_batch.argtypes = [ctypes.c_int, ctypes.c_char_p, i64, vp, i64, i64, vp, vp, vp, ctypes.c_int]
        L.fxo_batch.restype = None
        _lib = L
    return _lib


def batch(op, pattern, rows, nthreads=1):
    """op 0 = .in., 1 = .match., 2 = regex (spans).  rows: uint8 ndarray [n, L].  Every row pays the per-call compile,
    as the reference's elemental operators do."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, L = rows.shape
    flags = np.zeros(n, dtype=np.uint8)
    frm = np.zeros(n, dtype=np.int32)
    to = np.zeros(n, dtype=np.int32)
    vp = ctypes.c_void_p
    lib().fxo_batch(op, pattern, len(pattern), rows.ctypes.data_as(vp), n, L, flags.ctypes.data_as(vp),
                    frm.ctypes.data_as(vp), to.ctypes.data_as(vp), nthreads)
    return flags, frm, to
