# FXP_F_PREFIX_CHECK / FXP_F_SUFFIX_CHECK programs (round 6): the one-launch tile kernel with the per-row check of the start / of the match's end against the general
# kernel (FXAMD_FORCE_GENERAL=1 in a child process: the environment is read once), with none / 1 % / 20 % of the rows failing the check (they are finished by the
# general row procedure inside the launch).  2 M x 256 B and 8 M x 64 B rows of config-3 text.
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import forgex_amd
    from forgex_amd import synth
    dev = torch.device("cuda")
    pat, lit, bad, L, frac = sys.argv[2], sys.argv[3].encode().decode("unicode_escape").encode("latin-1"), sys.argv[4].encode().decode("unicode_escape").encode("latin-1"), int(sys.argv[5]), float(sys.argv[6])
    n = (2_000_000 * 256) // L
    rows = synth.batch("cfg3", 0, (n * L + 255) // 256, dev).reshape(-1)[: n * L].reshape(n, L).clone()
    g = torch.Generator(device="cpu").manual_seed(3)
    idx = torch.randperm(n, generator=g)
    hit = idx[: n // 4].to(dev)
    rows[hit, 20:20 + len(lit)] = torch.tensor(list(lit), dtype=torch.uint8, device=dev)
    k = int(n * frac)
    if k:
        w = idx[n // 4: n // 4 + k].to(dev)
        rows[w, 8:8 + len(bad)] = torch.tensor(list(bad), dtype=torch.uint8, device=dev)
    p = forgex_amd.Program(pat, forgex_amd.OP_SEARCH)
    f, a, b = p.match_device(rows)
    torch.cuda.synchronize()
    for _ in range(3):
        p.match_device(rows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        p.match_device(rows)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%d %.4f %.0f %d %d" % (p.last_path(), dt * 1e3, rows.numel() / dt / 1e9, int(f.sum()), int(a.sum() % 1000003)))
    sys.exit(0)
# pattern, a text that matches from a prefix occurrence, a text that matches but NOT from a candidate (the check fails: general procedure)
CASES = [(r"(}[abc]){2}\d*c{2,}", "}a}b12cc", "x}a}a}bcc"), (r"(\t{3}[a-z]){2}", "\\t\\t\\ta\\t\\t\\tb", "\\t\\t\\t\\ta\\t\\t\\tb"), (r"A{1,2}bb", "Abbx", "Abb")]
for pat, lit, bad in CASES:
    for L in (256, 64):
        for frac in (0.0, 0.01, 0.2):
            out = {}
            for arm, env in (("tile", {}), ("general", {"FXAMD_FORCE_GENERAL": "1"})):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", pat, lit, bad, str(L), str(frac)], env=dict(os.environ, **env), capture_output=True, text=True)
                out[arm] = r.stdout.split() if r.returncode == 0 else ["?", "0", "0", "-1", r.stderr[-200:]]
            same = out["tile"][3:] == out["general"][3:]
            print("%-22s L %3d  rows failing the check %4.0f %%  tile path %s %s ms %s GB/s | general path %s %s ms %s GB/s | same results %s" % (
                pat, L, frac * 100, out["tile"][0], out["tile"][1], out["tile"][2], out["general"][0], out["general"][1], out["general"][2], same), flush=True)
