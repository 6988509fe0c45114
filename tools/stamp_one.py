#!/usr/bin/env python3
"""Per-phase cycle shares of fx_search_one (debug build `make -C forgex_amd/csrc stamp-one [STAMP_OBJ=12_2]`; run on the GPU box).

    FXAMD_LIB=forgex_amd/libforgex_amd_stamp_one.so python tools/stamp_one.py [cfg4] [--flags-only] [--md]

Every wave accumulates s_memtime deltas per phase (scalar registers); lane 0 adds them to a device array at the wave's end.  Printed: each
phase's share of the waves' summed lifetimes, cycles per tile, and what the kernel's wall time (HIP events) is made of -- the average
and the longest wave lifetime against the launch's duration.  The stamps themselves cost a few percent (s_memtime + s_waitcnt per
stamp); the un-stamped library's time for the same call is printed next to it when FXAMD_REF_LIB names it.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FXAMD_LIB", os.path.join(ROOT, "forgex_amd", "libforgex_amd_stamp_one.so"))
import torch
import forgex_amd
from forgex_amd import synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = args[0] if args else "cfg4"
spans = "--flags-only" not in sys.argv
md = "--md" in sys.argv
n, L = synth.SHAPES[cfg]
n = min(n, 12_500_000)
dev = torch.device("cuda", 0)
rows = synth.batch(cfg, 0, n, dev)
prog = forgex_amd.Program(synth.PATTERNS[cfg], forgex_amd.OP_SEARCH)
lib = forgex_amd.lib()
lib.fxamd_debug_stamps_one.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 20)()

prog.match_device(rows, spans=spans)
torch.cuda.synchronize()
for _ in range(60):   # (the clocks settle over the first back-to-back launches)
    prog.match_device(rows, spans=spans)
torch.cuda.synchronize()
assert lib.fxamd_debug_stamps_one(buf) == 0
REPS = 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    prog.match_device(rows, spans=spans)
e1.record()
torch.cuda.synchronize()
assert lib.fxamd_debug_stamps_one(buf) == 0
us = e0.elapsed_time(e1) * 1000.0 / REPS
v = [float(x) for x in buf]
names = ["start-up (header, tables -> LDS, barrier)", "wait for the tile's loads + LDS store", "issue of the next tile's loads", "class-level scan (ASCII tile)",
         "speculative forward walk + results", "gather of queued rows (issue)", "byte-level scan in place", "byte-level scan of gathered rows", "in-LDS decode",
         "scan of decoded rows", "general row procedure", "end (compaction flush, loop exit)"]
tiles, gath, life_sum, life_max, waves = v[12] / REPS, v[13] / REPS, v[14], v[15], v[16] / REPS
tot = sum(v[:12]) or 1.0
# s_memtime counts at 100 MHz on gfx9 (the constant "REFCLK"); convert with the measured wall time of the longest wave when that is plausible
print("%s %s: path %d, %.2f us per launch (HIP events over %d launches), %d waves, %.1f tiles and %.2f gathered passes per wave" %
      (cfg, "spans" if spans else "flags only", prog.last_path(), us, REPS, waves, tiles / waves, gath / waves))
avg_life = life_sum / (REPS * waves)
print("wave lifetime: average %.0f ticks, longest %.0f ticks; launch = %.2f us -> 1 tick = %.4f us if the longest wave spans the launch" %
      (avg_life, life_max, us, us / life_max))
tick_us = us / life_max
if md:
    print("\n| phase | share of the waves' lifetime | ticks per wave | us per wave (at %.4f us per tick) |" % tick_us)
    print("|---|---|---|---|")
for nm, x in zip(names, v[:12]):
    per_wave = x / (REPS * waves)
    if md:
        print("| %s | %.1f %% | %.0f | %.2f |" % (nm, 100.0 * x / tot, per_wave, per_wave * tick_us))
    else:
        print("  %-44s %6.2f %%   %8.0f ticks per wave  %7.2f us" % (nm, 100.0 * x / tot, per_wave, per_wave * tick_us))
print("average wave lifetime %.2f us of the launch's %.2f us (%.0f %%): the rest is waves that start late or end early (ramp-up, drain, the longest tail)" %
      (avg_life * tick_us, us, 100.0 * avg_life * tick_us / us))
ref = os.environ.get("FXAMD_REF_US")
if ref:
    print("un-stamped library, same call: %s us" % ref)
