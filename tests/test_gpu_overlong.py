"""GPU parity on rows full of bytes >= 0x80 -- multi-byte characters, broken sequences, 0xFE / 0xFF, and OVERLONG encodings, which the reference
DECODES (src/essential/utf8_m.f90:338-430 `ichar_utf8` is arithmetic: `C0 80` is U+0000, so `^` matches behind it; `C1 A1` is `a`) -- at every row
length the first passes take: 256 (half-row pipeline), 2..128 (span kernel), 257.. (segment walker), through the C ABI against the oracle.

Round 5 tried to walk ASCII-minded searches (`[a-z]+\\d+`: every code point >= 0x80 of one class that kills the forward automaton and resets the
reverse one) on tables indexed by the RAW byte, with no deferral and no follow-up launch.  It is NOT sound and was removed: an overlong sequence is
bytes >= 0x80 that decode to a code point BELOW 0x80 (found by this test: `^[a-z]+` over a row holding `C0 80`), and telling those apart needs the byte
pairs `C0|C1 xx`, `E0 80|81`, `F0 80 80|81` -- more memory than an 8-state reverse automaton has.  What it would have bought on pure-ASCII batches, measured
before the removal (gpurun call r05_c32): config 3 0.474 -> 0.470 ms, config 5 0.339 -> 0.334 ms, 1024-byte rows 0.4955 -> 0.4878 ms."""
import os

import numpy as np
import pytest

import oracle_lib
from test_gpu_span import _rows

pytestmark = pytest.mark.gpu
NT = os.cpu_count() or 1

PATS = [r"[a-z]+\d+", r"\d+$", r"^[a-z]+", r"\d{2,3}[a-f]?", r"[A-Za-z_][A-Za-z0-9_]*=", r"(ab|cd)+\d", r"[a-z]+@[a-z]+", r"x*$", r"\w+x", r"[a-f]+ [g-z]",
        r"[α-ω]+\d", r".", r"[^a-z]", r"[a-z]*\s+\d", r"/+", r"a+"]


@pytest.fixture(scope="module")
def fx(built):
    import torch
    import forgex_amd
    assert torch.cuda.is_available()
    return forgex_amd


def _mixed_rows(L, n, seed):
    """test_gpu_span's rows with bytes >= 0x80 in every second one, plus rows that are ALL such bytes, rows of 0xFE / 0xFF, a match on either
    side of a multi-byte character / a broken sequence, and overlong encodings of NUL, `a`, `/`, `1` in two, three and four bytes"""
    rows = _rows(L, n, seed, 0.5)
    if L >= 16:
        rows[1, :] = 0xFE
        rows[2, :] = 0xFF
        rows[3, :] = np.frombuffer(("あ" * L).encode()[:L], dtype=np.uint8)
        pieces = [b"ab12\xe3\x81\x82cd345", b"ab1\xffcd34", b"abc\xfe123", b"\xceab12\xb1", b"a1\x80\x80\x80b2", b"123-4567\xe3", b"\xc0\xafxy7",
                  b"q \xc0\x80eik w", b"b\xc1\xa1\xc1\xa1\xc0\xb1 x", b"\xe0\x80\x80abc 9", b"zz\xe0\x81\xa1\xe0\x80\xb1\xe0\x80\xaf/", b"\xf0\x80\x80\x80ab\xf0\x80\x81\xa1\xf0\x80\x80\xb1",
                  b"ab\xc1\xa1\xc0\xb12", b"\xc0\x80\xc0\x80a1"]
        for i, pc in enumerate(pieces):
            r = np.full(L, 32, dtype=np.uint8)
            pc = pc[:L]
            r[:len(pc)] = np.frombuffer(pc, dtype=np.uint8)
            rows[4 + i] = r
            r2 = np.full(L, ord("q"), dtype=np.uint8)
            r2[L - len(pc):] = np.frombuffer(pc, dtype=np.uint8)
            rows[4 + len(pieces) + i] = r2
    return rows


@pytest.mark.parametrize("L", [256, 128, 64, 32, 16, 100, 20, 6, 257, 300, 400, 1024, 2049])
def test_rows_of_high_bytes_and_overlong_sequences_vs_oracle(fx, L):
    import torch
    n = 64 * 37 + 5
    rows = _mixed_rows(L, n, 7000 + L)
    dev_rows = torch.from_numpy(rows).cuda()
    for pat in PATS:
        of, oa, ob = oracle_lib.batch(2, pat.encode(), rows, NT)
        prog = fx.Program(pat, fx.OP_SEARCH)
        f, a, b = prog.match_device(dev_rows)
        torch.cuda.synchronize()
        path = prog.last_path()
        f, a, b = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
        bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
        assert bad.size == 0, (pat, L, path, int(bad[0]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]),
                               rows[bad[0]][:64].tobytes())
        ff, _, _ = prog.match_device(dev_rows, spans=False)   # the verdict alone (other kernels at most lengths): the same flags
        torch.cuda.synchronize()
        assert np.array_equal(ff.cpu().numpy(), of), (pat, L, "flags only")
        om, _, _ = oracle_lib.batch(1, pat.encode(), rows, NT)
        pm = fx.Program(pat, fx.OP_MATCH)
        fm, _, _ = pm.match_device(dev_rows, spans=False)
        torch.cuda.synchronize()
        assert np.array_equal(fm.cpu().numpy(), om), (pat, L, "match", pm.last_path())


def test_packed_results_and_batch_ends(fx):
    """packed results (the first pass writes them itself) and batches that end inside a tile, on the same kind of rows"""
    import torch
    for L in (256, 128, 20):
        for n in (1, 63, 64 * 9 + 1, 64 * 8 * 5 + 3):
            rows = _mixed_rows(L, max(n, 40), 9000 + L + n)[:n]
            dev_rows = torch.from_numpy(np.ascontiguousarray(rows)).cuda()
            for pat in PATS[:3]:
                of, oa, ob = oracle_lib.batch(2, pat.encode(), rows, NT)
                prog = fx.Program(pat, fx.OP_SEARCH)
                buf = prog.match_device_packed(dev_rows, spans=True)
                f, a, b = fx.unpack_results(buf, n, L, spans=True)
                torch.cuda.synchronize()
                assert np.array_equal(f.cpu().numpy(), of) and np.array_equal(a.cpu().numpy(), oa) and np.array_equal(b.cpu().numpy(), ob), (pat, L, n, prog.last_path())


def test_bordered_prefix_and_overlapping_occurrences_spelled_with_overlong_bytes(fx):
    """Round 6 fix (found by tests/support/fuzz_prefilter.py FX_FUZZ_UTF8=1, a bug since round 3's overlap sink): a bordered prefix literal (`aa[bc]`, `αα[^a]*`) whose
    occurrences overlap at the SYMBOL level while the literal occurs nowhere byte-wise -- possible only through non-canonical encodings (`a`, `C1 A1`, `a`, `b`).  The
    reference's INDEX finds no occurrence and falls back to brute force (src/api_internal_m.F90:76-81); the product's row procedure used its reverse automaton for
    that fallback -- the one composed with the overlap detector, which stops recording hits in its absorbing state -- and reported a later start.  It now takes the
    restart loop with the forward automaton there.  Rows of several lengths (span kernel's follow-up, one-launch kernel, long rows), spans and flags only."""
    rng = np.random.default_rng(61)
    seeds = [b"a\xc1\xa1ab", b"\xc1\xa1aab", b"aa\xc1\xa1b", b"a\xc1\xa1\xc1\xa1c", b"\xce\xb1\xe8\x83\xa1\xce\xb1\xe0\x8e\xb1\xce\xb1", b"\xce\xb1\xe0\x8e\xb1\xce\xb1x",
             b"\xe0\x8e\xb1\xce\xb1\xce\xb1", b"-\xc0\xad-a", b"\xc0\xad--ab", b"aab", b"aaab", b"\xce\xb1\xce\xb1\xce\xb1", b"x"]
    alpha = np.frombuffer(b"abcx- ", dtype=np.uint8)
    for L in (16, 64, 100, 256, 400):
        n = 64 * 6 + 3
        rows = alpha[rng.integers(0, len(alpha), size=(n, L))].copy()
        for i in range(n):
            sd = np.frombuffer(seeds[i % len(seeds)], dtype=np.uint8)
            if len(sd) <= L:
                off = int(rng.integers(0, L - len(sd) + 1))
                rows[i, off:off + len(sd)] = sd
        import torch
        dev_rows = torch.from_numpy(rows).cuda()
        for pat in (r"aa[bc]", r"αα[^a]*", r"--[a-z]+", r"aa.*b"):
            of, oa, ob = oracle_lib.batch(2, pat.encode(), rows, NT)
            prog = fx.Program(pat, fx.OP_SEARCH)
            f, a, b = prog.match_device(dev_rows)
            torch.cuda.synchronize()
            f, a, b = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
            bad = np.flatnonzero((f != of) | (a != oa) | (b != ob))
            assert bad.size == 0, (pat, L, int(bad[0]), bytes(rows[bad[0]]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]))
            f2, _, _ = prog.match_device(dev_rows, spans=False)
            torch.cuda.synchronize()
            assert np.array_equal(f2.cpu().numpy(), of), (pat, L, "flags only")
