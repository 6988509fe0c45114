import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import forgex_amd
from forgex_amd import synth
dev = torch.device("cuda")
rows = synth.batch("cfg3", 0, 2_000_000, dev).clone()
n, L = rows.shape
# 20 % of the rows get a latin-1 style byte (structurally invalid UTF-8), another 20 % a valid 2-byte character
g = torch.Generator(device="cpu").manual_seed(1)
idx = torch.randperm(n, generator=g)
bad, good = idx[: n // 5].to(dev), idx[n // 5: 2 * n // 5].to(dev)
rows[bad, 17] = 0xE9
rows[good, 40] = 0xC3
rows[good, 41] = 0xA9
for pat in ("[a-z]+\\d+", "id=\\d+", "ab.*\\d;", "\\d{3}-\\d{4}"):
    p = forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
    p.match_device(rows); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): p.match_device(rows)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%-14s path %d flags %x  %.3f ms  %.0f GB/s" % (pat, p.last_path(), p.info()["flags"], dt * 1e3, rows.numel() / dt / 1e9), flush=True)
