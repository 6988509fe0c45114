#!/usr/bin/env python3
"""A/B of two library builds on the few entry points every version has (compile, match_batch_device, free), bound with ctypes
directly: python tools/exp_lib_ab.py <libA.so> <libB.so>.  Config-3 bytes viewed as rows of several lengths."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from forgex_amd import synth

torch.cuda.init()
dev = torch.device("cuda")
base = synth.batch("cfg3", 0, 4_000_000, dev).reshape(-1)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.fxamd_compile.argtypes = [ctypes.c_char_p, i64, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)]
    L.fxamd_match_batch_device.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp]
    L.fxamd_program_free.argtypes = [vp]
    for pat in (rb"[a-z]+\d+", rb"\d{3}-\d{4}"):
        for rl in (256, 400, 512, 1008, 1024, 4096):
            n = base.numel() // rl
            rows = base[: n * rl]
            h = vp()
            st = ctypes.c_int32(0)
            assert L.fxamd_compile(pat, len(pat), 0, ctypes.byref(h), ctypes.byref(st)) == 0
            f = torch.empty(n, dtype=torch.uint8, device=dev)
            a = torch.empty(n, dtype=torch.int32, device=dev)
            b = torch.empty(n, dtype=torch.int32, device=dev)

            def step():
                rc = L.fxamd_match_batch_device(h, rows.data_ptr(), n, rl, f.data_ptr(), a.data_ptr(), b.data_ptr(), None)
                assert rc == 0, rc
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100
            print("%-44s %-12s L %5d  %.3f ms  %5.0f GB/s  matches %d" % (os.path.basename(os.path.dirname(path)) + "/" + os.path.basename(path), pat.decode(), rl, dt * 1e3,
                                                                         n * rl / dt / 1e9, int(f.sum())), flush=True)
            L.fxamd_program_free(h)
