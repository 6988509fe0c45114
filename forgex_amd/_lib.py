"""ctypes binding of libforgex_amd.so (include/forgex_amd.h).  The library is built in-tree by
`forgex_amd.build()` / `__graft_entry__.build()`; importing a match entry point without it fails loudly --
there is no Python or CPU fallback for matching."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FXAMD_LIB") or os.path.join(_HERE, "libforgex_amd.so")   # FXAMD_LIB: kernel-experiment builds only
CSRC = os.path.join(_HERE, "csrc")

OP_SEARCH, OP_MATCH = 0, 1
E_UNSUPPORTED = -5

_lib = None


def build(force=False):
    """Compile the HIP kernels + C ABI for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("forgex_amd: %s is missing -- run forgex_amd.build() (hipcc, gfx950). "
                           "There is no CPU fallback for the match path." % LIB_PATH)
    try:
        # When torch is in the process its bundled HIP runtime must be the one that initialises the GPU: load it
        # (and touch the device) BEFORE libforgex_amd.so pulls in a second copy of libamdhip64.
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32p, i64 = c.c_void_p, c.POINTER(c.c_int32), c.c_int64
    L.fxamd_compile.argtypes = [c.c_char_p, i64, c.c_int, c.POINTER(vp), i32p]
    L.fxamd_compile.restype = c.c_int
    L.fxamd_compile_nfa.argtypes = [c.c_int32, c.c_int32, c.c_int32, i64, vp, vp, vp, vp, vp, c.c_char_p, i64, c.c_char_p, i64,
                                    c.c_char_p, i64, c.c_int, c.POINTER(vp), i32p]
    L.fxamd_compile_nfa.restype = c.c_int
    L.fxamd_program_free.argtypes = [vp]
    L.fxamd_program_free.restype = None
    L.fxamd_program_status.argtypes = [vp]
    L.fxamd_program_status.restype = c.c_int32
    L.fxamd_program_blob_size.argtypes = [vp]
    L.fxamd_program_blob_size.restype = i64
    L.fxamd_program_blob.argtypes = [vp, vp, i64]
    L.fxamd_program_blob.restype = c.c_int
    L.fxamd_program_from_blob.argtypes = [vp, i64, c.POINTER(vp)]
    L.fxamd_program_from_blob.restype = c.c_int
    L.fxamd_program_info.argtypes = [vp, i32p]
    L.fxamd_program_info.restype = c.c_int
    L.fxamd_strerror.argtypes = [c.c_int32]
    L.fxamd_strerror.restype = c.c_char_p
    L.fxamd_strerror_copy.argtypes = [c.c_int32, vp, i64]
    L.fxamd_strerror_copy.restype = i64
    L.fxamd_program_upload.argtypes = [vp]
    L.fxamd_program_upload.restype = c.c_int
    L.fxamd_match_batch_device.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp]
    L.fxamd_match_batch_device.restype = c.c_int
    L.fxamd_packed_layout.argtypes = [i64, i64, c.c_int, c.POINTER(i64), c.POINTER(i64), c.POINTER(i64), i32p]
    L.fxamd_packed_layout.restype = c.c_int
    L.fxamd_match_batch_device_packed.argtypes = [vp, vp, i64, i64, c.c_int, vp, vp]
    L.fxamd_match_batch_device_packed.restype = c.c_int
    L.fxamd_unpack_results.argtypes = [vp, i64, i64, c.c_int, vp, vp, vp, vp]
    L.fxamd_unpack_results.restype = c.c_int
    L.fxamd_program_reserve.argtypes = [vp, i64, vp]
    L.fxamd_program_reserve.restype = c.c_int
    L.fxamd_match_multi_device.argtypes = [vp, c.c_int32, vp, i64, i64, vp, vp, vp, vp]
    L.fxamd_match_multi_device.restype = c.c_int
    L.fxamd_match_batch_host.argtypes = [vp, vp, i64, i64, vp, vp, vp]
    L.fxamd_match_batch_host.restype = c.c_int
    L.fxamd_launch_fast_only.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp]
    L.fxamd_launch_fast_only.restype = c.c_int
    L.fxamd_last_path.argtypes = [vp]
    L.fxamd_last_path.restype = c.c_int
    L.fxamd_last_hip_error.argtypes = []
    L.fxamd_last_hip_error.restype = c.c_int
    L.fxamd_device_count.argtypes = []
    L.fxamd_device_count.restype = c.c_int
    L.fxamd_reload_env.argtypes = []
    L.fxamd_reload_env.restype = None
    L.fxamd_cache_trim.argtypes = []
    L.fxamd_cache_trim.restype = i64
    L.fxamd_batch_upload.argtypes = [vp, i64, i64, c.POINTER(vp)]
    L.fxamd_batch_wrap.argtypes = [vp, i64, i64, c.POINTER(vp)]
    L.fxamd_batch_free.argtypes = [vp]
    L.fxamd_batch_free.restype = None
    L.fxamd_batch_info.argtypes = [vp, c.POINTER(i64), c.POINTER(i64)]
    L.fxamd_batch_run.argtypes = [c.POINTER(vp), c.c_int32, vp, c.c_int]
    L.fxamd_batch_sync.argtypes = [vp]
    L.fxamd_batch_after.argtypes = [vp, vp]
    L.fxamd_batch_fetch.argtypes = [vp, c.c_int32, vp, vp, vp]
    L.fxamd_batch_count.argtypes = [vp, c.c_int32, c.POINTER(i64)]
    L.fxamd_batch_results.argtypes = [vp, c.POINTER(vp), c.POINTER(vp), c.POINTER(vp), i32p, c.POINTER(vp)]
    for name in ("fxamd_batch_upload", "fxamd_batch_wrap", "fxamd_batch_info", "fxamd_batch_run", "fxamd_batch_sync", "fxamd_batch_after", "fxamd_batch_fetch",
                 "fxamd_batch_count", "fxamd_batch_results"):
        getattr(L, name).restype = c.c_int
    L.fxamd_host_register.argtypes = [vp, i64]
    L.fxamd_host_register.restype = c.c_int
    L.fxamd_host_unregister.argtypes = [vp]
    L.fxamd_host_unregister.restype = c.c_int
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "fxamd_compile", "fxamd_compile_nfa", "fxamd_program_free", "fxamd_program_status", "fxamd_program_blob_size",
    "fxamd_program_blob", "fxamd_program_from_blob", "fxamd_program_info", "fxamd_strerror", "fxamd_strerror_copy", "fxamd_program_upload", "fxamd_program_reserve",
    "fxamd_match_batch_device", "fxamd_packed_layout", "fxamd_match_batch_device_packed", "fxamd_unpack_results", "fxamd_match_multi_device", "fxamd_match_batch_host", "fxamd_last_path", "fxamd_last_hip_error", "fxamd_device_count",
    "fxamd_host_register", "fxamd_host_unregister", "fxamd_f_compile", "fxamd_f_program_free", "fxamd_f_strerror_copy", "fxamd_f_match_batch_host",
    "fxamd_cache_trim", "fxamd_batch_upload", "fxamd_batch_wrap", "fxamd_batch_free", "fxamd_batch_info", "fxamd_batch_run", "fxamd_batch_sync", "fxamd_batch_after", "fxamd_batch_fetch",
    "fxamd_batch_count", "fxamd_batch_results", "fxamd_f_batch_upload", "fxamd_f_batch_wrap", "fxamd_f_batch_free", "fxamd_f_batch_run",
    "fxamd_f_batch_sync", "fxamd_f_batch_fetch", "fxamd_f_batch_count",
]
BENCH_SYMBOLS = ["fxamd_launch_fast_only", "fxamd_reload_env"]   # include/forgex_amd_bench.h: measurement hooks, not part of the boundary
