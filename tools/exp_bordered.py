# Bordered prefix literals (FXP_F_OVERLAP_SINK): tile-kernel rate vs the general kernel (forced with a misaligned base),
# with and without rows that contain an overlap witness.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import forgex_amd
from forgex_amd import synth
dev = torch.device("cuda")
n = 2_000_000
base = synth.batch("cfg3", 0, n, dev).clone()
def rate(p, rows):
    p.match_device(rows); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): p.match_device(rows)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5
for pat, wit, lit in (("--[a-z]+", b"---", b"--ab"), ("aa[bc]", b"aaa", b"aab"), ("zz\\d+", b"zzz", b"zz12"), ("abab\\d", b"ababab", b"abab7")):
    for frac in (0.0, 0.01, 0.2):
        rows = base.clone()
        g = torch.Generator(device="cpu").manual_seed(3)
        idx = torch.randperm(n, generator=g)
        hit = idx[: n // 2].to(dev)
        rows[hit, 30:30 + len(lit)] = torch.tensor(list(lit), dtype=torch.uint8, device=dev)
        k = int(n * frac)
        if k:
            w = idx[n // 2: n // 2 + k].to(dev)
            rows[w, 100:100 + len(wit)] = torch.tensor(list(wit), dtype=torch.uint8, device=dev)
        p = forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
        dt = rate(p, rows); path = p.last_path()
        # general kernel: same bytes at a base that is not 16-byte aligned
        flat = torch.empty(rows.numel() + 16, dtype=torch.uint8, device=dev)
        mis = flat[1:1 + rows.numel()].view(rows.shape); mis.copy_(rows)
        q = forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
        dg = rate(q, mis)
        print("%-10s witness rows %4.0f%%  tile path %d %.3f ms %.0f GB/s | general path %d %.3f ms %.0f GB/s" % (
            pat, frac * 100, path, dt * 1e3, rows.numel() / dt / 1e9, q.last_path(), dg * 1e3, rows.numel() / dg / 1e9), flush=True)
