import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import forgex_amd
from forgex_amd import synth
dev = torch.device("cuda")
rows = synth.batch("cfg3", 0, 4_000_000, dev)
for L in (256, 512, 1024, 4096, 400, 1008):
    nb = rows.numel() // L * L
    r = rows.view(-1)[:nb].view(-1, L)
    for pat in ("[a-z]+\\d+", "\\d{3}-\\d{4}"):
        p = forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
        p.match_device(r); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): p.match_device(r)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(L, pat, "path", p.last_path(), "%.3f ms  %.0f GB/s" % (dt * 1e3, r.numel() / dt / 1e9), flush=True)
