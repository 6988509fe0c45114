#!/usr/bin/env python3
"""Per-phase cycle shares of fx_search_fast (debug build `make -C forgex_amd/csrc stamp`; run on the GPU box).

    FXAMD_LIB=forgex_amd/libforgex_amd_stamp.so python tools/stamp_phases.py [cfg3]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FXAMD_LIB", os.path.join(ROOT, "forgex_amd", "libforgex_amd_stamp.so"))
import torch
import forgex_amd
from forgex_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n, L = synth.SHAPES[cfg]
n = min(n, 10_000_000)
dev = torch.device("cuda", 0)
rows = synth.batch(cfg, 0, n, dev)
prog = forgex_amd.Program(synth.PATTERNS[cfg], forgex_amd.OP_SEARCH)
lib = forgex_amd.lib()
lib.fxamd_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
for spans in (True, False):
    prog.match_device(rows, spans=spans)
    torch.cuda.synchronize()
    lib.fxamd_debug_stamps(buf)
    for _ in range(60):   # (the clocks settle over the first ~30 back-to-back launches)
        prog.match_device(rows, spans=spans)
    torch.cuda.synchronize()
    lib.fxamd_debug_stamps(buf)
    for _ in range(20):
        prog.match_device(rows, spans=spans)
    torch.cuda.synchronize()
    lib.fxamd_debug_stamps(buf)
    v = list(buf)[:8]
    tot = float(sum(v)) or 1.0
    names = ["wait loads + store_tile", "issue prefetch (+decode/pad)", "backward loop", "re-walk + leading NUL", "forward first 32", "forward tail loop",
             "output + loop end", "loop head"]
    print("spans" if spans else "flags only")
    for nm, x in zip(names, v):
        print("  %-30s %6.2f %%   %8.0f cycles per tile" % (nm, 100.0 * x / tot, x / (20.0 * ((n + 63) // 64))))
