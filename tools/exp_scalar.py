#!/usr/bin/env python3
"""Cost of ONE scalar call through the host-buffer entry (what `pattern .in. text` of the Fortran module does per element: compile,
match one row, free) -- with the compile cache and the pooled host pipes, and with FXAMD_NO_CACHE=1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forgex_amd as fx

texts = [b"ab12  cd345", b"no digits here", b"   z9", b"xxfoobarbaz"]
for label, pats in (("same pattern", [rb"[a-z]+\d+"] * 400), ("16 patterns in turn", [("p%d[a-z]+\\d+" % (i % 16)).encode() for i in range(400)]),
                    ("all different", [("q%d[a-z]+\\d+" % i).encode() for i in range(400)])):
    fx.in_(pats[0], texts[0])
    t0 = time.perf_counter()
    for i, p in enumerate(pats):
        fx.in_(p, texts[i % 4])
    dt = (time.perf_counter() - t0) / len(pats)
    print("%-22s %.1f us per scalar call  [FXAMD_NO_CACHE=%s]" % (label, dt * 1e6, os.environ.get("FXAMD_NO_CACHE", "")), flush=True)
