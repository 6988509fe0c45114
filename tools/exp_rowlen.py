import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import forgex_amd
from forgex_amd import synth
dev = torch.device("cuda")
rows = synth.batch("cfg3", 0, 8_000_000, dev)   # 2 GB of config-3 bytes, viewed at several row lengths
for L in (16, 32, 64, 128, 256, 100, 255, 8):
    nb = rows.numel() // L * L
    r = rows.view(-1)[:nb].view(-1, L)
    if L in (100, 255, 8):
        r = r.contiguous()
    for spans in (True, False):
        p = forgex_amd.Program("[a-z]+\\d+", forgex_amd.OP_SEARCH)
        p.match_device(r, spans=spans); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): p.match_device(r, spans=spans)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("L %3d %s path %d  %.3f ms  %.0f GB/s  (%.1f M rows)" % (L, "spans" if spans else "flags", p.last_path(), dt * 1e3, r.numel() / dt / 1e9, r.shape[0] / 1e6), flush=True)
