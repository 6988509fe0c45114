"""Host-side mirror of the reference's `use forgex` surface for the batch match path.

Reference interface mirrored here (reference src/forgex.F90:24-54):
    is_valid_regex(pattern)                                  :58-71
    pattern .in. str      -> in_(pattern, strs)              :74-160
    pattern .match. str   -> match(pattern, strs)            :163-231
    call regex(pattern, text, res, length, from, to, status, err_msg) -> regex(pattern, text)   :235-347
    regex_f(pattern, text)                                   :351-358
Same argument meaning and error behaviour: an invalid pattern makes `.in.`/`.match.` False for every element and
makes regex return ('', 0, -9999, -9999, status, message).  Strings are BYTES (Fortran default character); `str`
arguments are encoded as UTF-8.  Batches are what the elemental operators take as rank-1 arrays: every element of
one batch has the same length (a Fortran `character(L) :: s(n)`); a Python list of unequal lengths is grouped by
length.  All matching runs on the GPU through libforgex_amd.so -- there is no CPU path.
"""
import ctypes

import numpy as np

from . import _lib

INVALID_CHAR_INDEX = -9999


def _b(x):
    return x.encode("utf-8") if isinstance(x, str) else bytes(x)


def strerror(status):
    return _lib.lib().fxamd_strerror(int(status)).decode()


class Program:
    """One compiled pattern (= one `tree%build` + `automaton%init` of the reference, done once per batch)."""

    def __init__(self, pattern, op):
        L = _lib.lib()
        pat = _b(pattern)
        self.pattern, self.op = pat, op
        h = ctypes.c_void_p()
        st = ctypes.c_int32(0)
        rc = L.fxamd_compile(pat, len(pat), op, ctypes.byref(h), ctypes.byref(st))
        if rc != 0:
            raise RuntimeError("fxamd_compile failed: %d" % rc)
        self._h = h
        self.status = st.value

    @classmethod
    def from_blob(cls, blob, op):
        L = _lib.lib()
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        rc = L.fxamd_program_from_blob(buf, len(blob), ctypes.byref(h))
        if rc != 0:
            raise ValueError("fxamd_program_from_blob failed: %d" % rc)
        self._h, self.op, self.pattern = h, op, None
        self.status = L.fxamd_program_status(h)
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().fxamd_program_free(h)
            except Exception:
                pass
            self._h = None

    @property
    def valid(self):
        return self.status == 0 or self.status >= 100

    @property
    def supported(self):
        return self.status < 100

    def info(self):
        a = (ctypes.c_int32 * 8)()
        _lib.lib().fxamd_program_info(self._h, a)
        keys = ["mode", "flags", "nA", "nR", "n_classes", "status", "total_bytes", "n_bounds"]
        return dict(zip(keys, list(a)))

    def blob(self):
        L = _lib.lib()
        n = L.fxamd_program_blob_size(self._h)
        buf = (ctypes.c_char * n)()
        rc = L.fxamd_program_blob(self._h, buf, n)
        if rc != 0:
            raise RuntimeError("fxamd_program_blob failed: %d" % rc)
        return bytes(buf)

    def last_path(self):
        return _lib.lib().fxamd_last_path(self._h)

    # ---- device-resident batch: torch uint8 CUDA tensor [n, L] --------------------------------------------
    def match_device(self, rows, spans=True, out=None):
        import torch
        if not rows.is_cuda or rows.dtype != torch.uint8 or rows.dim() != 2 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous uint8 CUDA tensor of shape [n, row_len]")
        n, rl = rows.shape
        if out is None:
            flags = torch.empty(n, dtype=torch.uint8, device=rows.device)
            frm = torch.empty(n, dtype=torch.int32, device=rows.device) if spans else None
            to = torch.empty(n, dtype=torch.int32, device=rows.device) if spans else None
        else:
            flags, frm, to = out
        stream = torch.cuda.current_stream(rows.device).cuda_stream
        with torch.cuda.device(rows.device):
            rc = _lib.lib().fxamd_match_batch_device(self._h, rows.data_ptr() if n else None, n, rl, flags.data_ptr(),
                                                     frm.data_ptr() if frm is not None else None,
                                                     to.data_ptr() if to is not None else None, stream)
        if rc == _lib.E_UNSUPPORTED:
            raise NotImplementedError("pattern %r is valid but not supported by the device path (status %d: %s)" % (
                self.pattern, self.status, strerror(self.status)))
        if rc != 0:
            raise RuntimeError("fxamd_match_batch_device failed: %d (hip error %d)" % (rc, _lib.lib().fxamd_last_hip_error()))
        return flags, frm, to

    def match_device_packed(self, rows, spans=True, out=None):
        """PACKED results of a device-resident batch (what a multi-GPU host gathers): one uint8 CUDA tensor holding 1 bit per row and,
        with spans, from / to narrowed to the row length (layout: packed_layout).  Rows of up to 256 bytes are packed by the search
        kernel itself."""
        import torch
        if not rows.is_cuda or rows.dtype != torch.uint8 or rows.dim() != 2 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous uint8 CUDA tensor of shape [n, row_len]")
        n, rl = rows.shape
        spans = bool(spans) and self.op == _lib.OP_SEARCH
        total = packed_layout(n, rl, spans)[2]
        packed = out if out is not None else torch.empty(max(total, 16), dtype=torch.uint8, device=rows.device)
        stream = torch.cuda.current_stream(rows.device).cuda_stream
        with torch.cuda.device(rows.device):
            rc = _lib.lib().fxamd_match_batch_device_packed(self._h, rows.data_ptr() if n else None, n, rl, 1 if spans else 0, packed.data_ptr(), stream)
        if rc == _lib.E_UNSUPPORTED:
            raise NotImplementedError("pattern %r is valid but not supported by the device path (status %d)" % (self.pattern, self.status))
        if rc != 0:
            raise RuntimeError("fxamd_match_batch_device_packed failed: %d (hip error %d)" % (rc, _lib.lib().fxamd_last_hip_error()))
        return packed

    # ---- host batch: numpy uint8 [n, L] -----------------------------------------------------------------------
    def match_host(self, rows, spans=True):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n, rl = rows.shape
        flags = np.zeros(n, dtype=np.uint8)
        frm = np.zeros(n, dtype=np.int32) if spans else None   # (`.match.` programs leave them untouched: 0 / 0)
        to = np.zeros(n, dtype=np.int32) if spans else None
        vp = ctypes.c_void_p
        rc = _lib.lib().fxamd_match_batch_host(self._h, rows.ctypes.data_as(vp), n, rl, flags.ctypes.data_as(vp),
                                               frm.ctypes.data_as(vp) if spans else None, to.ctypes.data_as(vp) if spans else None)
        if rc == _lib.E_UNSUPPORTED:
            raise NotImplementedError("pattern %r is valid but not supported by the device path (status %d)" % (self.pattern, self.status))
        if rc != 0:
            raise RuntimeError("fxamd_match_batch_host failed: %d (hip error %d)" % (rc, _lib.lib().fxamd_last_hip_error()))
        return flags, frm, to


class pinned:
    """`with pinned(rows): prog.match_host(rows)` -- the numpy array pinned in place for the block (fxamd_host_register): the host
    entry then reads it by DMA instead of staging pageable memory."""

    def __init__(self, arr):
        self.arr = np.ascontiguousarray(arr)
        if self.arr is not arr:
            raise ValueError("pinned() needs a C-contiguous array (it is pinned in place)")

    def __enter__(self):
        rc = _lib.lib().fxamd_host_register(self.arr.ctypes.data_as(ctypes.c_void_p), self.arr.nbytes)
        if rc != 0:
            raise RuntimeError("fxamd_host_register failed: %d (hip error %d)" % (rc, _lib.lib().fxamd_last_hip_error()))
        return self.arr

    def __exit__(self, exc_type, exc, tb):
        rc = _lib.lib().fxamd_host_unregister(self.arr.ctypes.data_as(ctypes.c_void_p))
        if rc != 0 and exc_type is None:   # a failed unpin leaves the memory page-locked: say so (unless an exception is already on its way)
            raise RuntimeError("fxamd_host_unregister failed: %d (hip error %d)" % (rc, _lib.lib().fxamd_last_hip_error()))
        return False


def packed_layout(n, row_len, spans=True):
    """(off_from, off_to, total_bytes, span_bytes) of a packed result image (fxamd_packed_layout)."""
    a, b, t = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    w = ctypes.c_int32(0)
    rc = _lib.lib().fxamd_packed_layout(int(n), int(row_len), 1 if spans else 0, ctypes.byref(a), ctypes.byref(b), ctypes.byref(t), ctypes.byref(w))
    if rc != 0:
        raise ValueError("fxamd_packed_layout failed: %d" % rc)
    return a.value, b.value, t.value, w.value


def unpack_results(packed, n, row_len, spans=True):
    """packed image (uint8 CUDA tensor) -> (flags uint8[n], from int32[n], to int32[n]) on the same device."""
    import torch
    flags = torch.empty(n, dtype=torch.uint8, device=packed.device)
    frm = torch.empty(n, dtype=torch.int32, device=packed.device) if spans else None
    to = torch.empty(n, dtype=torch.int32, device=packed.device) if spans else None
    stream = torch.cuda.current_stream(packed.device).cuda_stream
    with torch.cuda.device(packed.device):
        rc = _lib.lib().fxamd_unpack_results(packed.data_ptr(), n, row_len, 1 if spans else 0, flags.data_ptr(), frm.data_ptr() if spans else None,
                                             to.data_ptr() if spans else None, stream)
    if rc != 0:
        raise RuntimeError("fxamd_unpack_results failed: %d" % rc)
    return flags, frm, to


def match_many(programs, rows, spans=True):
    """m compiled programs against the same device-resident rows (torch uint8 CUDA tensor [n, L]): flags [m, n] (and from / to)."""
    import torch
    if not rows.is_cuda or rows.dtype != torch.uint8 or rows.dim() != 2 or not rows.is_contiguous():
        raise ValueError("rows must be a contiguous uint8 CUDA tensor of shape [n, row_len]")
    n, rl = rows.shape
    m = len(programs)
    flags = torch.empty((m, n), dtype=torch.uint8, device=rows.device)
    frm = torch.zeros((m, n), dtype=torch.int32, device=rows.device) if spans else None
    to = torch.zeros((m, n), dtype=torch.int32, device=rows.device) if spans else None
    handles = (ctypes.c_void_p * m)(*[p._h for p in programs])
    stream = torch.cuda.current_stream(rows.device).cuda_stream
    with torch.cuda.device(rows.device):
        rc = _lib.lib().fxamd_match_multi_device(handles, m, rows.data_ptr() if n else None, n, rl, flags.data_ptr(),
                                                 frm.data_ptr() if spans else None, to.data_ptr() if spans else None, stream)
    if rc != 0:
        raise RuntimeError("fxamd_match_multi_device failed: %d (hip error %d)" % (rc, _lib.lib().fxamd_last_hip_error()))
    return flags, frm, to


def _as_batches(strs):
    """-> list of (indices, ndarray[n, L]) with uniform L, plus scalar flag."""
    if isinstance(strs, (bytes, bytearray, str)):
        b = _b(strs)
        return [(np.array([0]), np.frombuffer(b, dtype=np.uint8).reshape(1, len(b)))], True, 1
    if isinstance(strs, np.ndarray) and strs.dtype == np.uint8 and strs.ndim == 2:
        return [(np.arange(strs.shape[0]), strs)], False, strs.shape[0]
    items = [_b(s) for s in strs]
    by_len = {}
    for i, s in enumerate(items):
        by_len.setdefault(len(s), []).append(i)
    out = []
    for ln, idx in by_len.items():
        arr = np.frombuffer(b"".join(items[i] for i in idx), dtype=np.uint8).reshape(len(idx), ln)
        out.append((np.array(idx), arr))
    return out, False, len(items)


def is_valid_regex(pattern):
    return Program(pattern, _lib.OP_SEARCH).valid


def _flags(pattern, strs, op):
    prog = Program(pattern, op)
    batches, scalar, n = _as_batches(strs)
    res = np.zeros(n, dtype=bool)
    if prog.status == 0:
        for idx, arr in batches:
            f, _, _ = prog.match_host(arr, spans=False)
            res[idx] = f != 0
    elif prog.status >= 100:
        raise NotImplementedError("pattern %r: %s" % (pattern, strerror(prog.status)))
    return bool(res[0]) if scalar else res


def in_(pattern, strs):
    """`pattern .in. strs` (elemental over strs)."""
    return _flags(pattern, strs, _lib.OP_SEARCH)


def match(pattern, strs):
    """`pattern .match. strs` (elemental over strs)."""
    return _flags(pattern, strs, _lib.OP_MATCH)


def regex(pattern, text):
    """`call regex(pattern, text, res, length, from, to, status, err_msg)` -> (res, length, from, to, status, err_msg).
    `text` may be one string or a batch; for a batch every field is a list/array."""
    prog = Program(pattern, _lib.OP_SEARCH)
    batches, scalar, n = _as_batches(text)
    msg = strerror(prog.status)
    if prog.status != 0 and prog.status < 100:   # forgex.F90:266-274
        if scalar:
            return b"", 0, INVALID_CHAR_INDEX, INVALID_CHAR_INDEX, prog.status, msg
        return ([b""] * n, np.zeros(n, np.int32), np.full(n, INVALID_CHAR_INDEX, np.int32),
                np.full(n, INVALID_CHAR_INDEX, np.int32), prog.status, msg)
    if prog.status >= 100:
        raise NotImplementedError("pattern %r: %s" % (pattern, msg))
    frm = np.zeros(n, np.int32)
    to = np.zeros(n, np.int32)
    res = [b""] * n
    for idx, arr in batches:
        f, a, b = prog.match_host(arr, spans=True)
        frm[idx], to[idx] = a, b
        for k, i in enumerate(idx):
            if a[k] > 0 and b[k] > 0:
                res[i] = arr[k, a[k] - 1:b[k]].tobytes()
    length = np.where((frm > 0) & (to > 0), to - frm + 1, 0).astype(np.int32)
    if scalar:
        return res[0], int(length[0]), int(frm[0]), int(to[0]), 0, msg
    return res, length, frm, to, 0, msg


def regex_f(pattern, text):
    """`regex_f(pattern, text)` -> matched substring ('' when none or when the pattern is invalid)."""
    return regex(pattern, text)[0]


class Batch:
    """Rows resident in HBM across calls and patterns, for hosts without a device runtime (what the Fortran module's type(fx_batch)
    binds: fxamd_batch_* of include/forgex_amd.h).  `rows`: a C-contiguous uint8 numpy array [n, row_len] (uploaded once) or a torch
    CUDA tensor (wrapped, not copied).  run() enqueues one or more programs and leaves the results on the device; fetch() / count()
    bring back one result set / the number of matching rows."""

    def __init__(self, rows):
        L = _lib.lib()
        h = ctypes.c_void_p()
        self._keep = None
        if isinstance(rows, np.ndarray):
            if rows.dtype != np.uint8 or rows.ndim != 2 or not rows.flags.c_contiguous:
                raise ValueError("rows must be a C-contiguous uint8 array [n, row_len]")
            n, rl = rows.shape
            rc = L.fxamd_batch_upload(rows.ctypes.data_as(ctypes.c_void_p), n, rl, ctypes.byref(h))
        else:
            if not rows.is_cuda or rows.dim() != 2 or not rows.is_contiguous():
                raise ValueError("rows must be a contiguous uint8 CUDA tensor [n, row_len]")
            n, rl = rows.shape
            self._keep = rows
            rc = L.fxamd_batch_wrap(ctypes.c_void_p(rows.data_ptr()), n, rl, ctypes.byref(h))
            if rc == 0:   # the tensor's producer (torch's current stream on its device) goes before the batch's private stream
                import torch
                rc2 = L.fxamd_batch_after(h, ctypes.c_void_p(torch.cuda.current_stream(rows.device).cuda_stream))
                if rc2 != 0:   # (an unordered private stream would be a data race with the producer, not an error anybody sees: ADVICE r05)
                    L.fxamd_batch_free(h)
                    raise RuntimeError("fxamd_batch_after failed: %d" % rc2)
        if rc != 0:
            raise RuntimeError("fxamd_batch_upload / _wrap failed: %d" % rc)
        self._h, self.n, self.row_len = h, int(n), int(rl)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().fxamd_batch_free(h)
            except Exception:
                pass
            self._h = None

    def run(self, progs, spans=True):
        progs = [progs] if isinstance(progs, Program) else list(progs)
        arr = (ctypes.c_void_p * len(progs))(*[p._h for p in progs])
        if self._keep is not None:   # wrapped tensor: whatever torch's current stream did to it since goes first
            import torch
            rc = _lib.lib().fxamd_batch_after(self._h, ctypes.c_void_p(torch.cuda.current_stream(self._keep.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("fxamd_batch_after failed: %d" % rc)
        rc = _lib.lib().fxamd_batch_run(arr, len(progs), self._h, 1 if spans else 0)
        self._spans = bool(spans) and all(p.op == _lib.OP_SEARCH for p in progs)
        if rc != 0:
            raise RuntimeError("fxamd_batch_run failed: %d" % rc)

    def sync(self):
        rc = _lib.lib().fxamd_batch_sync(self._h)
        if rc != 0:
            raise RuntimeError("fxamd_batch_sync failed: %d" % rc)

    def fetch(self, which=0, spans=None):
        if spans is None:   # what the last run produced
            spans = getattr(self, "_spans", True)
        flags = np.empty(self.n, np.uint8)
        frm = np.empty(self.n, np.int32) if spans else None
        to = np.empty(self.n, np.int32) if spans else None
        vp = ctypes.c_void_p
        rc = _lib.lib().fxamd_batch_fetch(self._h, which, flags.ctypes.data_as(vp), frm.ctypes.data_as(vp) if spans else None,
                                          to.ctypes.data_as(vp) if spans else None)
        if rc != 0:
            raise RuntimeError("fxamd_batch_fetch failed: %d" % rc)
        return flags, frm, to

    def count(self, which=0):
        c = ctypes.c_int64(0)
        rc = _lib.lib().fxamd_batch_count(self._h, which, ctypes.byref(c))
        if rc != 0:
            raise RuntimeError("fxamd_batch_count failed: %d" % rc)
        return int(c.value)
