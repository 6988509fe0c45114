#!/usr/bin/env python3
"""Where config 4's time goes: the same batch with and without its structurally invalid rows (exception queues + merged decode pass
at the end of every block), and pure-ASCII rows of the same shape through the same program."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import forgex_amd as fx
from forgex_amd import synth

dev = torch.device("cuda")
n, L = synth.SHAPES["cfg4"]
rows = synth.batch("cfg4", 0, n, dev)
idx = torch.arange(n, dtype=torch.int64, device=dev)
r = synth._rowhash(idx, synth.SEEDS["cfg4"], 0)
corrupt = (synth._lsr(r, 20) % 100) == 0
clean = rows.clone()
src = torch.nonzero(~corrupt)[: int(corrupt.sum())].flatten()
clean[corrupt] = rows[src]
ascii_rows = synth.batch("cfg2", 0, n * L // 64, dev).reshape(n, L).contiguous()


def rate(prog, x, spans=True, reps=200):
    out = prog.match_device(x, spans=spans)
    for _ in range(30):
        prog.match_device(x, spans=spans, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        prog.match_device(x, spans=spans, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


p = fx.Program(synth.PATTERNS["cfg4"], fx.OP_SEARCH)
for name, x in (("cfg4 as generated (%.2f %% invalid rows)" % (100.0 * float(corrupt.float().mean())), rows), ("invalid rows replaced by valid ones", clean),
                ("pure ASCII rows, same shape", ascii_rows)):
    for spans in (True, False):
        dt = rate(p, x, spans)
        print("%-52s %-6s path %2d  %.1f us  %.0f GB/s" % (name, "spans" if spans else "flags", p.last_path(), dt * 1e6, x.numel() / dt / 1e9), flush=True)
