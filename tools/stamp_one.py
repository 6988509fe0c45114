#!/usr/bin/env python3
"""Per-phase cycle shares of fx_search_one (debug build `make -C forgex_amd/csrc stamp-one [STAMP_OBJ=12_2]`: the object <chunks>_<part> the
config's kernel lives in -- config 4: 12_2, config 2: 4_3; run on the GPU box).

    FXAMD_LIB=forgex_amd/libforgex_amd_stamp_one_12_2.so python tools/stamp_one.py [cfg4] [--flags-only] [--md]

Every wave accumulates s_memtime deltas per phase (LDS slots, lane 0) and adds them to its own row of a device buffer at its end (plain stores).
Printed: each phase's ticks per wave (average over waves and launches), its share of the waves' lifetime, the spread of the waves' lifetimes, and
the launch's duration by HIP events next to it.  s_memtime counts shader-clock ticks; the tick length is taken from the longest wave's lifetime
against the launch's duration (one round of resident blocks: the longest wave spans the launch but for its dispatch).  The stamps cost a few percent
(s_memtime + a wait for outstanding LDS operations per stamp, one explicit wait for the tile's loads); `FXAMD_REF_US` prints the product's time.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FXAMD_LIB", os.path.join(ROOT, "forgex_amd", "libforgex_amd_stamp_one_12_2.so"))
import torch
import forgex_amd
from forgex_amd import synth

# --fast: the phases of fx_search_fast (`make stamp-fast STAMP_FAST_OBJ=16_1`: long rows; 8_1: the half-row kernel of 256-byte rows) instead of fx_search_one's;
# `shape:<name>`: a workload of tools/bench_shapes.py (config bytes viewed at another row length) instead of a BASELINE config
FAST = "--fast" in sys.argv
MAX_WAVES, SLOTS = (65536, 12) if FAST else (16384, 20)
LIFE, COUNT, TILES = (10, 11, 8) if FAST else (18, 19, 12)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = args[0] if args else "cfg4"
spans = "--flags-only" not in sys.argv
md = "--md" in sys.argv
dev = torch.device("cuda", 0)
if cfg.startswith("shape:"):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_shapes
    desc, op, pats, src, view, n, packed, spans = bench_shapes.SHAPES[cfg[6:]]
    assert op == "search" and len(pats) == 1 and isinstance(view, int)
    _, L0 = synth.SHAPES[src]
    L = view
    flat = synth.batch(src, 0, (n * L + L0 - 1) // L0, dev).reshape(-1)
    rows = flat[: n * L].reshape(n, L)
    pattern = pats[0]
else:
    n, L = synth.SHAPES[cfg]
    n = min(n, 12_500_000)
    rows = synth.batch(cfg, 0, n, dev)
    pattern = synth.PATTERNS[cfg]
prog = forgex_amd.Program(pattern, forgex_amd.OP_SEARCH)
lib = forgex_amd.lib()
reader = lib.fxamd_debug_stamps_fast if FAST else lib.fxamd_debug_stamps_one
reader.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
buf = np.zeros(MAX_WAVES * SLOTS, dtype=np.uint64)


def read():
    assert reader(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    return buf.reshape(MAX_WAVES, SLOTS).astype(np.float64)


prog.match_device(rows, spans=spans)
torch.cuda.synchronize()
for _ in range(60):   # (the clocks settle over the first back-to-back launches)
    prog.match_device(rows, spans=spans)
torch.cuda.synchronize()
read()
REPS = 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    prog.match_device(rows, spans=spans)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000.0 / REPS
v = read()
live = v[:, COUNT] > 0
w = v[live] / v[live][:, COUNT:COUNT + 1]            # per wave, per launch
waves = int(live.sum())
if FAST:
    names = {7: "loop head", 0: "wait for the segment's loads + staging registers -> LDS", 1: "issue of the next segment's loads (+ decode / pad)", 2: "backward loop (all segments of the tile)",
             3: "exact start (re-walk) + leading NUL", 4: "forward pass: first 32 symbols", 5: "forward pass: longer matches", 6: "results + loop end"}
    order = [7, 0, 1, 2, 3, 4, 5, 6]
else:
    names = {0: "start-up (header, tables -> LDS, barrier)", 1: "wait for the tile's loads", 14: "staging registers -> LDS", 2: "issue of the next tile's loads",
             3: "class-level scan (ASCII tile)", 4: "speculative forward walk + results", 5: "gather of queued rows (global -> LDS)", 6: "byte-level scan in place",
             7: "byte-level scan of gathered rows", 8: "in-LDS decode", 9: "scan of decoded rows", 10: "general row procedure", 11: "end (compaction flush, loop exit)"}
    order = [0, 1, 14, 2, 3, 4, 6, 5, 7, 8, 9, 10, 11]
life = w[:, LIFE]
tiles, gath = w[:, TILES].mean(), (0.0 if FAST else w[:, 13].mean())
tick_us = us / life.max()
if "--ghz" in sys.argv or FAST:   # (grids of several rounds of blocks: no wave spans the launch -- a shader clock is assumed instead: config 4's one-round launch measured 2.04 GHz)
    tick_us = 1e-3 / float(sys.argv[sys.argv.index("--ghz") + 1] if "--ghz" in sys.argv else 2.04)
print("%s %s: path %d, %.2f us per launch (HIP events over %d launches; the product's kernel: %s us), %d waves%s, %.1f tiles and %.2f gathered passes per wave" %
      (cfg, "spans" if spans else "flags only", prog.last_path(), us, REPS, os.environ.get("FXAMD_REF_US", "?"), waves, " (or more: the buffer holds %d)" % MAX_WAVES if waves == MAX_WAVES else "", tiles, gath))
print("wave lifetime in ticks: mean %.0f, median %.0f, 5 %% %.0f, 95 %% %.0f, longest %.0f -> 1 tick = %.5f us if the longest wave spans the launch (%.2f GHz)" %
      (life.mean(), np.median(life), np.percentile(life, 5), np.percentile(life, 95), life.max(), tick_us, 1e-3 / tick_us))
tot = sum(w[:, i].mean() for i in order)
if md:
    print("\n| phase | share of a wave's lifetime | ticks per wave | us per wave | per tile, us |")
    print("|---|---|---|---|---|")
for i in order:
    x = w[:, i].mean()
    if x == 0:
        continue
    if md:
        print("| %s | %.1f %% | %.0f | %.2f | %.3f |" % (names[i], 100.0 * x / tot, x, x * tick_us, x * tick_us / max(tiles, 1.0)))
    else:
        print("  %-44s %6.2f %%   %8.0f ticks per wave  %7.2f us  %6.3f us per tile" % (names[i], 100.0 * x / tot, x, x * tick_us, x * tick_us / max(tiles, 1.0)))
print("sum of the phases %.2f us; mean wave lifetime %.2f us = %.0f %% of the launch's %.2f us (the rest: dispatch of the blocks, waves that end before the longest one)" %
      (tot * tick_us, life.mean() * tick_us, 100.0 * life.mean() * tick_us / us, us))
