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


_pool = None
_pool_size = 0


def _workers(nthreads):
    """One persistent pool of host threads (ctypes releases the GIL during the call)."""
    global _pool, _pool_size
    if _pool is None or _pool_size < nthreads:
        from concurrent.futures import ThreadPoolExecutor
        if _pool is not None:
            _pool.shutdown(wait=True)
        _pool = ThreadPoolExecutor(max_workers=nthreads)
        _pool_size = nthreads
    return _pool


def batch(op, pattern, rows, nthreads=1):
    """op 0 = .in., 1 = .match., 2 = regex (spans).  rows: uint8 ndarray [n, L].  Every row pays the per-call compile,
    as the reference's elemental operators do.  nthreads > 1: the rows are cut into chunks that a persistent pool of host threads
    works off, each through a single-threaded fxo_batch call.  (fxo_batch's own OpenMP region is fast only while libgomp's workers
    are still spinning from the last call; between calls a test does device work, the spinning workers and the host's other
    threads oversubscribe the cores, and a ONE-row call then takes 30-64 ms here on 8 threads -- 0.1 s per call with 256 on the
    GPU box, where the suite makes thousands of small oracle calls: 77 s of test_tiny_rows_every_length_and_batch_end alone.
    Pool workers sleep on a queue instead.)"""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, L = rows.shape
    flags = np.zeros(n, dtype=np.uint8)
    frm = np.zeros(n, dtype=np.int32)
    to = np.zeros(n, dtype=np.int32)
    fn = lib().fxo_batch
    rp, fp, ap, bp = rows.ctypes.data, flags.ctypes.data, frm.ctypes.data, to.ctypes.data
    if nthreads <= 1 or n * max(L, 16) < 16384:
        fn(op, pattern, len(pattern), rp, n, L, fp, ap, bp, 1)
        return flags, frm, to
    chunk = max(8, min(512, n // (4 * nthreads) + 1))   # (several chunks per thread: rows cost unevenly -- a no-match row is the restart loop's worst case)

    def work(i0):
        m = min(chunk, n - i0)
        fn(op, pattern, len(pattern), rp + i0 * L, m, L, fp + i0, ap + 4 * i0, bp + 4 * i0, 1)

    for _ in _workers(nthreads).map(work, range(0, n, chunk)):
        pass
    return flags, frm, to
