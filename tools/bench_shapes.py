#!/usr/bin/env python3
"""The workloads behind DESIGN.md 4.1d's table other than the four `.in.` configs bench.py covers: `.match.`, rows longer than
256 bytes, literal search, many patterns in one pass, packed results -- one named SHAPE per run, so that the same command can sit
behind `rocprofv3 --kernel-trace --stats` / `--pmc` (tools/profile_shapes.sh) and every row of that table has a kernel trace and
PMC traffic under profiles/.

    python tools/bench_shapes.py --shape match_cfg3 [--steps 50 --warmup 20]
    python tools/bench_shapes.py --list

One JSON line: step time (HIP events over the timed steps), input GB/s, algorithmic bytes per step (rows * (row_len + out bytes
per row)) and the fraction of the 8 TB/s HBM peak they amount to, last_path.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name -> (description, op, patterns, source config, row_len view (None = the config's), rows, packed, spans)
SHAPES = {
    "match_cfg3":   ("`.match.` `[a-z ]+\\d*[a-z ]*` over config-3 rows (8-state tables, one verdict byte per row)", "match", [r"[a-z ]+\d*[a-z ]*"], "cfg3", None, 10_000_000, False, False),
    "match_chain_cfg3": ("`.match.` `[a-z ]{6}[a-z ]*\\d{0,3}[a-z ]{6}[a-z ]*` (23 states: chain tables) over config-3 rows", "match", [r"[a-z ]{6}[a-z ]*\d{0,3}[a-z ]{6}[a-z ]*"], "cfg3", None, 10_000_000, False, False),
    "match_chain_128": ("`.match.` of the 23-state pattern (chain tables) over config 5's shard (12.5 M x 128 B)", "match", [r"[a-z ]{6}[a-z ]*\d{0,3}[a-z ]{6}[a-z ]*"], "cfg5", None, 12_500_000, False, False),
    "match_cfg1x":  ("`.match.` `\\d{3}-\\d{4}` (config 1's pattern, nibble tables) over 64M x 8 B rows of config 1's generator", "match", [r"\d{3}-\d{4}"], "cfg1", None, 64_000_000, False, False),
    "match_cfg5":   ("`.match.` `[a-z ]+\\d*[a-z ]*` over config 5's shard (12.5 M x 128 B, 8-state tables)", "match", [r"[a-z ]+\d*[a-z ]*"], "cfg5", None, 12_500_000, False, False),
    "match_utf8":   ("`.match.` `[α-ωぁ-ん ]+` over config-4 rows (byte-level tables)", "match", ["[α-ωぁ-ん ]+"], "cfg4", None, 1 << 20, False, False),
    "in_flags_cfg1x": ("`.in.` verdict (flags only) `\\d{3}-\\d{4}` over 64M x 8 B rows of config 1's generator (fx_search_tiny)", "search", [r"\d{3}-\d{4}"], "cfg1", None, 64_000_000, False, False),
    "match_ragged_200": ("`.match.` over config-3 bytes viewed as rows of 200 B (ragged rows of the one-launch kernel)", "match", [r"[a-z ]+\d*[a-z ]*[a-z 0-9]*"], "cfg3", 200, 12_800_000, False, False),
    "match_long_1024": ("`.match.` `[a-z ]+\\d*[a-z ]*[a-z 0-9]*` over config-3 bytes viewed as 2.5M x 1024 B rows (segment loop of fx_match_fast)", "match", [r"[a-z ]+\d*[a-z ]*[a-z 0-9]*"], "cfg3", 1024, 2_500_000, False, False),
    "match_long_chain_1024": ("`.match.` of a 23-state pattern (chain tables) over config-3 bytes viewed as 2.5M x 1024 B rows", "match", [r"[a-z ]{6}[a-z ]*\d{0,3}[a-z ]{6}[a-z ]*"], "cfg3", 1024, 2_500_000, False, False),
    "long_1024":    ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as 2.5M x 1024 B rows", "search", [r"[a-z]+\d+"], "cfg3", 1024, 2_500_000, False, True),
    "long_chain_1024": ("`[a-z]{6}\\d{1,3}[a-z ]{6}` (17 states: chain tables) `.in.` + spans, config-3 bytes viewed as 2.5M x 1024 B rows", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg3", 1024, 2_500_000, False, True),
    "long_4096":    ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as 625k x 4096 B rows", "search", [r"[a-z]+\d+"], "cfg3", 4096, 625_000, False, True),
    "long_400":     ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as 6.4M x 400 B rows (not a multiple of 16)", "search", [r"[a-z]+\d+"], "cfg3", 400, 6_400_000, False, True),
    "nibble_cfg3":  ("`\\d{3}-\\d{4}` `.in.` + spans over config-3 rows (9..16 states: nibble tables)", "search", [r"\d{3}-\d{4}"], "cfg3", None, 10_000_000, False, True),
    "chain_cfg3":   ("an e-mail pattern `.in.` + spans over config-3 rows (> 16 states: chain tables)", "search", [r"[a-z0-9]+@[a-z0-9]+\.[a-z]{2,4}"], "cfg3", None, 10_000_000, False, True),
    "chain17_cfg3": ("`[a-z]{6}\\d{1,3}[a-z ]{6}` (17 states: chain tables only) `.in.` + spans over config-3 rows: half of the rows match", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg3", None, 10_000_000, False, True),
    "chain17_200":  ("the 17-state pattern over config-3 bytes viewed as rows of 200 B (chain tables on ragged rows of the one-launch kernel)", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg3", 200, 12_800_000, False, True),
    "chain17_128":  ("the 17-state pattern over config 5's shard (12.5 M x 128 B)", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg5", None, 12_500_000, False, True),
    "nibble_cfg3_flags": ("`\\d{3}-\\d{4}` `.in.` verdict (flags only) over config-3 rows (nibble tables)", "search", [r"\d{3}-\d{4}"], "cfg3", None, 10_000_000, False, False),
    "chain17_cfg3_flags": ("the 17-state pattern, `.in.` verdict (flags only) over config-3 rows (chain tables, half rows)", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg3", None, 10_000_000, False, False),
    "nibble_128":   ("`\\d{3}-\\d{4}` (nibble tables) `.in.` + spans over config 5's shard (12.5 M x 128 B)", "search", [r"\d{3}-\d{4}"], "cfg5", None, 12_500_000, False, True),
    "chain17_64":   ("the 17-state pattern over config-3 bytes viewed as 40 M rows of 64 B", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg3", 64, 40_000_000, False, True),
    "chain17_192":  ("the 17-state pattern over config-3 bytes viewed as rows of 192 B", "search", [r"[a-z]{6}\d{1,3}[a-z ]{6}"], "cfg3", 192, 13_333_333, False, True),
    "literal_cfg2": ("literal `foobar` `.in.` + spans over 16M x 64 B rows of config 2's generator (raw-byte INDEX on the tile kernel)", "search", ["foobar"], "cfg2", None, 16 << 20, False, True),
    "multi6_cfg3":  ("six 8-state patterns over config-3 rows in ONE pass (fx_search_multi)", "search",
                     [r"[a-z]+\d+", r"\d+[a-z]", r"[a-z]+ \d", r"q[a-z]*\d", r"\d\d+", r"[a-z]\d[a-z]"], "cfg3", None, 10_000_000, False, True),
    "packed_cfg5":  ("config 5's shard with PACKED results (1 bit + 2 x uint8 per row, written by the search kernel)", "search", [r"[a-z]+\d+"], "cfg5", None, 12_500_000, True, True),
    "packed_cfg3":  ("config 3 with PACKED results (half-row pipeline + fx_pack)", "search", [r"[a-z]+\d+"], "cfg3", None, 10_000_000, True, True),
    # config 4's text at other row lengths (whole characters: the first L // 5 five-byte slots of a config-4 row, blank-padded): the same automata at
    # other occupancies (rows of 64 / 96 / 128 / 192 bytes: 4 / 3 / 3 / 2 blocks of the one-launch kernel per CU)
    "utf8_64":      ("config 4's pattern and text in rows of 64 B", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 64), 3 << 20, False, True),
    "utf8_96":      ("config 4's pattern and text in rows of 96 B", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 96), 2 << 20, False, True),
    "utf8_128":     ("config 4's pattern and text in rows of 128 B", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 128), 3 << 19, False, True),
    "utf8_192":     ("config 4 itself (rows of 192 B)", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 192), 1 << 20, False, True),
    "utf8_192_clean": ("config 4 without its corrupted rows (each replaced by its predecessor): what the exception path costs", "search", ["[α-ωぁ-ん]+"], "cfg4", ("clean", 192), 1 << 20, False, True),
    "utf8_192_flags": ("config 4, flags only", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 192), 1 << 20, False, False),
    "utf8_128_flags": ("config 4's text in rows of 128 B, flags only", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 128), 3 << 19, False, False),
    "utf8_64_flags": ("config 4's text in rows of 64 B, flags only", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 64), 3 << 20, False, False),
    "ragged_255":   ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as rows of 255 B (ragged loader)", "search", [r"[a-z]+\d+"], "cfg3", 255, 10_000_000, False, True),
    # row lengths Fortran programs declare (character(80), (100), (132), (200)): 2.56 GB of config-3 bytes viewed at that length
    "ragged_80":    ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as rows of 80 B", "search", [r"[a-z]+\d+"], "cfg3", 80, 32_000_000, False, True),
    "ragged_100":   ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as rows of 100 B", "search", [r"[a-z]+\d+"], "cfg3", 100, 25_600_000, False, True),
    "ragged_132":   ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as rows of 132 B", "search", [r"[a-z]+\d+"], "cfg3", 132, 19_393_939, False, True),
    "ragged_200":   ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as rows of 200 B", "search", [r"[a-z]+\d+"], "cfg3", 200, 12_800_000, False, True),
    # rows of 64 / 32 / 16 bytes with spans (the span kernel, fx_span.hpp): config 5's bytes viewed at that length (one row in 4 / 8 / 16 matches)
    "rows_64":      ("`[a-z]+\\d+` `.in.` + spans, config-5 bytes viewed as 25 M rows of 64 B", "search", [r"[a-z]+\d+"], "cfg5", 64, 25_000_000, False, True),
    "rows_32":      ("`[a-z]+\\d+` `.in.` + spans, config-5 bytes viewed as 50 M rows of 32 B", "search", [r"[a-z]+\d+"], "cfg5", 32, 50_000_000, False, True),
    "rows_16":      ("`[a-z]+\\d+` `.in.` + spans, config-5 bytes viewed as 100 M rows of 16 B", "search", [r"[a-z]+\d+"], "cfg5", 16, 100_000_000, False, True),
    "nibble_rows_32": ("`\\d{3}-\\d{4}` (nibble tables) `.in.` + spans, config-5 bytes viewed as 50 M rows of 32 B", "search", [r"\d{3}-\d{4}"], "cfg5", 32, 50_000_000, False, True),
    "nibble_rows_16": ("`[a-z]{2}\\d{2,3}[a-z ]?x?` (nibble tables) `.in.` + spans, config-5 bytes viewed as 100 M rows of 16 B", "search", [r"[a-z]{2}\d{2,3}[a-z ]?x?"], "cfg5", 16, 100_000_000, False, True),
    "ragged_20":    ("`[a-z]+\\d+` `.in.` + spans, config-3 bytes viewed as rows of 20 B", "search", [r"[a-z]+\d+"], "cfg3", 20, 64_000_000, False, True),
    # short rows of a length that does not divide 64 (character(10), (12), (20), (24)): the tiny-row kernels' ragged spans
    "match_rows_12": ("`.match.` `[a-z ]+\\d*[a-z ]*` over config-3 bytes viewed as 100M rows of 12 B (fx_match_tiny, ragged spans)", "match", [r"[a-z ]+\d*[a-z ]*"], "cfg3", 12, 100_000_000, False, False),
    "in_flags_rows_20": ("`.in.` verdict `[a-z]+\\d+` over config-3 bytes viewed as 64M rows of 20 B (fx_search_tiny, ragged spans)", "search", [r"[a-z]+\d+"], "cfg3", 20, 64_000_000, False, False),
    "in_flags_rows_10": ("`.in.` verdict `[a-z]+\\d+` over config-3 bytes viewed as 128M rows of 10 B", "search", [r"[a-z]+\d+"], "cfg3", 10, 128_000_000, False, False),
    "ragged_255_flags": ("the same at 255 B, flags only", "search", [r"[a-z]+\d+"], "cfg3", 255, 10_000_000, False, False),
    # config 4's pattern and text at ragged row lengths (byte-level tables on ragged rows: round 4)
    "utf8_100":     ("config 4's pattern and text in rows of 100 B", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 100), 2 << 20, False, True),
    "utf8_132":     ("config 4's pattern and text in rows of 132 B", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 132), 3 << 19, False, True),
    "utf8_256":     ("config 4's pattern and text in rows of 256 B (255 B of text + a blank): the half-row first pass defers every tile to the gated follow-up", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 256), 1 << 20, False, True),
    "utf8_256_any": ("`[a-z ]+` (a pattern that matches ASCII text too) over config 4's text in rows of 256 B: the half-row pipeline on a batch that is mostly UTF-8 (FX_ADAPT_CALLS)", "search", ["[a-z ]+"], "cfg4", ("cut", 256), 1 << 20, False, True),
    "utf8_255":     ("config 4's pattern and text in rows of 255 B (190 B of text, blank-padded)", "search", ["[α-ωぁ-ん]+"], "cfg4", ("cut", 255), 1 << 20, False, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="match_cfg3")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=0, help="another row count than the shape's (small-batch experiments)")
    args = ap.parse_args()
    if args.list:
        for k, v in SHAPES.items():
            print("%-14s %s" % (k, v[0]))
        return
    desc, op, pats, cfg, view, n, packed, spans = SHAPES[args.shape]
    if args.rows > 0:
        n = args.rows
    import torch
    import forgex_amd
    from forgex_amd import synth
    dev = torch.device("cuda", 0)
    _, L0 = synth.SHAPES[cfg]
    if view is None:
        L = L0
        rows = synth.batch(cfg, 0, n, dev)
    elif isinstance(view, tuple):
        L = view[1]
        keep = (L // 5) * 5 if L < L0 else L0
        full = synth.batch(cfg, 0, n, dev)
        rows = torch.full((n, L), 32, dtype=torch.uint8, device=dev)
        rows[:, :keep] = full[:, :keep]
        del full
        if view[0] == "clean":
            idx = torch.arange(n, dtype=torch.int64, device=dev)
            corrupt = (synth._lsr(synth._rowhash(idx, synth.SEEDS[cfg], 0), 20) % 100) == 0
            for _ in range(4):   # (runs of corrupted rows: a few passes)
                src = torch.where(corrupt, (idx - 1).clamp(min=0), idx)
                rows = rows[src]
                corrupt = corrupt[src] & corrupt
    else:
        L = view
        n0 = (n * L + L0 - 1) // L0
        flat = synth.batch(cfg, 0, n0, dev).reshape(-1)
        rows = flat[: n * L].reshape(n, L)
    opc = forgex_amd.OP_MATCH if op == "match" else forgex_amd.OP_SEARCH
    progs = [forgex_amd.Program(p, opc) for p in pats]
    assert all(p.status == 0 for p in progs)
    m = len(progs)
    if m > 1:
        def step():
            return forgex_amd.match_many(progs, rows, spans=spans)
        out_bytes = m * (9 if spans else 1)
    elif packed:
        buf = progs[0].match_device_packed(rows, spans=spans)

        def step():
            return progs[0].match_device_packed(rows, spans=spans, out=buf)
        w = forgex_amd.packed_layout(n, L, spans)[3]
        out_bytes = 0.125 + 2 * w
    else:
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        frm = torch.empty(n, dtype=torch.int32, device=dev) if spans else None
        to = torch.empty(n, dtype=torch.int32, device=dev) if spans else None

        def step():
            return progs[0].match_device(rows, spans=spans, out=(flags, frm, to))
        out_bytes = 9 if spans else 1
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.steps):
        res = step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    if m > 1:
        matches = int((res[0] != 0).sum().item())
    elif packed:
        matches = None
    else:
        matches = int((flags != 0).sum().item())
    alg = n * (L + out_bytes)
    print(json.dumps({"shape": args.shape, "what": desc, "rows": n, "row_len": L, "patterns": pats, "op": op, "ms_per_step": ms,
                      "input_gbs": n * L / (ms * 1e-3) / 1e9, "algorithmic_bytes_per_step": alg,
                      "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / 8000.0, "last_path": [p.last_path() for p in progs], "matches": matches,
                      "steps": args.steps, "warmup": args.warmup}))


if __name__ == "__main__":
    main()
