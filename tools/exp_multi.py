#!/usr/bin/env python3
"""m patterns over one batch: fx_search_multi (one pass over the rows) against one pipeline per pattern (FXAMD_NO_MULTI=1 in a
second process), and against a single pattern.  Usage: python tools/exp_multi.py [cfg] [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import forgex_amd as fx
from forgex_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else synth.SHAPES[cfg][0]
rows = synth.batch(cfg, 0, n, torch.device("cuda"))
sets = {
    "6 mixed (test_many_patterns)": [rb"[a-z]+\d+", rb"\d{3}-\d{4}", rb"zz+", rb"[0-9]$", b"needle", rb"(ab|cd)+\d"],
    "6 on the 8-state tables": [rb"[a-z]+\d+", rb"[0-9]$", b"needle", rb"(ab|cd)+\d", rb"x[yz]+\d", rb"[a-f]+ [g-z]"],
    "2 on the 8-state tables": [rb"[a-z]+\d+", rb"(ab|cd)+\d"],
    # (round 6: automata of 9..16 states can share the pass too -- FXAMD_MULTI_W16=1; by default one pipeline each)
    "6 with three on the nibble tables": [rb"[a-z]+\d+", rb"\d{3}-\d{4}", rb"[a-z0-9._]+@[a-z0-9]+\.[a-z]+", rb"\d\d:\d\d:\d\d", b"needle", rb"(ab|cd)+\d"],
    "3 on the nibble tables": [rb"\d{3}-\d{4}", rb"[a-z0-9._]+@[a-z0-9]+\.[a-z]+", rb"\d\d:\d\d:\d\d"],
}
if cfg == "cfg4":   # UTF-8 rows: the shared pass scans them with the patterns' byte-level tables (FXAMD_MULTI_NO_BYTES=1: defers them instead)
    sets["6 UTF-8 patterns"] = [synth.PATTERNS["cfg4"].encode(), "[ぁ-ん]+".encode(), "[α-ω][ぁ-ん]".encode(), "ん[α-ω]+".encode(), rb"[a-z]+",
                                 "(α|β|γ)[ぁ-ん].".encode()]
    sets["5 UTF-8 patterns whose tables decode"] = [synth.PATTERNS["cfg4"].encode(), "[ぁ-ん]+".encode(), "[α-ω][ぁ-ん]".encode(), rb"[a-z]+", "(α|β|γ)[ぁ-ん].".encode()]


def rate(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


single = fx.Program(synth.PATTERNS[cfg].encode() if cfg == "cfg4" else rb"[a-z]+\d+", fx.OP_SEARCH)
out1 = single.match_device(rows)
t1 = rate(lambda: single.match_device(rows, out=out1))
print("%s %d rows: single pattern %.3f ms (path %d)  [FXAMD_NO_MULTI=%s FXAMD_MULTI_NO_BYTES=%s FXAMD_MULTI_W16=%s]" % (
    cfg, n, t1 * 1e3, single.last_path(), os.environ.get("FXAMD_NO_MULTI", ""), os.environ.get("FXAMD_MULTI_NO_BYTES", ""), os.environ.get("FXAMD_MULTI_W16", "")), flush=True)
for name, pats in sets.items():
    progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
    tm = rate(lambda: fx.match_many(progs, rows), reps=10)
    paths = [p.last_path() for p in progs]
    tf = rate(lambda: fx.match_many(progs, rows, spans=False), reps=10)
    print("  %-38s spans %.3f ms = %.2f x single   flags only %.3f ms   paths %s" % (name, tm * 1e3, tm / t1, tf * 1e3, paths), flush=True)
