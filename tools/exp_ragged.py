import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import forgex_amd
from forgex_amd import synth
dev = torch.device("cuda")
rows = synth.batch("cfg3", 0, 8_000_000, dev)
for L in (80, 100, 255, 132, 8, 256):
    nb = rows.numel() // L * L
    r = rows.view(-1)[:nb].view(-1, L).contiguous()
    for pat in ("[a-z]+\\d+", "\\d{3}-\\d{4}"):
        p = forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
        p.match_device(r); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): p.match_device(r)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("L %3d %-12s path %d  %.3f ms  %.0f GB/s" % (L, pat, p.last_path(), dt * 1e3, r.numel() / dt / 1e9), flush=True)
