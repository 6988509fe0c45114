"""GPU parity tests of the span kernel (round 5, forgex_amd/csrc/fx_span.hpp): `.in.` / regex with spans over rows of 128 / 64 / 32 / 16 bytes,
where a lane owns a 128-byte span of one / two / four / eight whole rows and rows that need the exact start + forward pass are compacted in LDS.  Through the C ABI, bit-exact against the oracle and against the one-launch
kernel (FXAMD_NO_SPAN=1) on the same rows.  Reference semantics: src/forgex.F90:74 (elemental: rows are independent),
src/api_internal_m.F90:108-155 (leftmost start, longest end), :140-148 (span arithmetic)."""
import os
import random

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu
NT = os.cpu_count() or 1


@pytest.fixture(scope="module")
def fx(built):
    import torch
    import forgex_amd
    assert torch.cuda.is_available()
    return forgex_amd


# programs whose class-level tables decode UTF-8 (first pass + gated follow-up: last_path 18) ...
PATS_DECODE = [r"[a-z]+\d+", r"[a-z ]+\d*", r"^[a-z]+", r"\d+$", r"x*$", r"[^a-z]", r".", r"(|^)a", r"\d{2,3}[a-f]?", r"[a-z]*\s+\d", r"[α-ω]+\d", r"q.{100,}z"]
# ... and candidate-list driver programs (prefix literal: rows with bytes >= 0x80 / in the overlap state reach the general row procedure through the follow-up)
PATS_GEN = [r"foo(bar|baz)", r"aa[bc]", r"abc.*xyz", r"--[a-z]+", r"ab(c|d)e", r"zz\d+"]


def _rows(L, n, seed, hi_frac=0.0):
    """Rows of every kind the kernel has a branch for: the config-5 / config-2 distributions, matches at the row's first and last byte,
    over the whole row, none; NUL / LF / CR bytes; rows with valid and broken UTF-8 (hi_frac of them)."""
    rng = random.Random(seed)
    alpha = b"abcdefghijklmnopqrstuvwxyz"
    out = np.empty((n, L), dtype=np.uint8)
    hi = ["α".encode(), "ω".encode(), "あ".encode(), "é".encode(), b"\xf0\x9f\x98\x80", b"\x80", b"\xe3\x81", b"\xff", b"\xc0\xaf"]
    for i in range(n):
        kind = rng.random()
        if L < 8:           # very short rows: a small alphabet, so that matches of every pattern occur
            b = bytearray(rng.choice(b"ab01 -zfo\n") for _ in range(L))
            if rng.random() < hi_frac:
                s_ = rng.choice([x for x in hi if len(x) <= L])
                k = rng.randint(0, L - len(s_))
                b[k:k + len(s_)] = s_
            out[i] = np.frombuffer(bytes(b), dtype=np.uint8)
            continue
        if kind < 0.30:     # letters and blanks, digits planted late (config 3 / 5)
            b = bytearray(rng.choice(alpha + b"    ") for _ in range(L))
            if rng.random() < 0.5:
                k = rng.randint(min(L * 3 // 4, L - 4), L - 4)
                b[k - 1] = rng.choice(alpha)
                for j in range(rng.randint(1, 3)):
                    b[k + j] = rng.choice(b"0123456789")
        elif kind < 0.45:   # planted literals (config 2)
            b = bytearray(rng.choice(alpha) for _ in range(L))
            if rng.random() < 0.4:
                lit = rng.choice([b"foobar", b"foobaz", b"fooba", b"aab", b"aaab", b"aaac", b"abcqqxyz", b"--ab--", b"---a", b"abce", b"abde", b"zz9", b"zzz12"])[:L]
                k = rng.randint(0, L - len(lit))
                b[k:k + len(lit)] = lit
        elif kind < 0.55:   # a match over the whole row / at its ends
            b = bytearray(rng.choice(alpha) for _ in range(L))
            which = rng.randint(0, 4)
            if which == 0:
                b[L - 1] = rng.choice(b"0123456789")
            elif which == 1:
                b[0] = rng.choice(b"0123456789")
            elif which == 2:
                b[0:2] = b"a1"
            elif which == 3:
                b[0] = ord("q")
                b[L - 1] = ord("z")
            else:
                b[L - 3:L] = b"x12"
        elif kind < 0.65:   # control bytes
            b = bytearray(rng.choice(alpha + b"0123456789 ") for _ in range(L))
            for _ in range(rng.randint(1, 4)):
                b[rng.randint(0, L - 1)] = rng.choice(b"\0\n\r\t")
            if rng.random() < 0.3:
                b[L - 2:L] = b"\r\n"
        elif kind < 0.75:   # digits only / blanks only / one repeated byte
            c = rng.choice([b"0123456789", b" ", b"a", b"-", b"z"])
            b = bytearray(rng.choice(c) for _ in range(L))
        else:               # uniform printable ASCII
            b = bytearray(rng.randint(32, 126) for _ in range(L))
        if rng.random() < hi_frac:
            for _ in range(rng.randint(1, 3)):
                s = rng.choice(hi)
                k = rng.randint(0, L - len(s))
                b[k:k + len(s)] = s
            if rng.random() < 0.2:
                b[L - 1] = 0xE3   # truncated lead byte at the very end
        out[i] = np.frombuffer(bytes(b[:L]), dtype=np.uint8)
    return out


def _cell(L):
    """bytes of LDS a row of L bytes gets in the span kernel (its length rounded up to 16 / 32 / 64 / 128)"""
    return 16 if L <= 16 else (32 if L <= 32 else (64 if L <= 64 else 128))


def _span_path(prog, L):
    """The path a search with spans over 128- / 64- / 32- / 16-byte rows takes by fxamd.hip's rule: 18 (span kernel + gated follow-up) for
    programs on the 8-state tables; None = another kernel."""
    fl = prog.info()["flags"]
    if not (fl & 8) or (fl & ((1 << 20) | (1 << 10))) or prog.info()["mode"] != 1:   # FXP_F_FAST_OK; FXP_F_NEEDS_NONASCII, FXP_F_RAW_BYTES
        return None
    if (fl & 2) and L <= 64 and not (int(os.environ.get("FXAMD_SPAN_LENS", "47")) & 16):   # FXP_F_PREFILTER: sparse matches -- the one-launch kernel's match compaction
        return None
    return 18 if 2 <= L <= 128 else None


def _check(fx, pat, rows_np, dev_rows, want_path, label, ref=None):
    import torch
    of, oa, ob = ref if ref is not None else oracle_lib.batch(2, pat.encode(), rows_np, NT)
    prog = fx.Program(pat, fx.OP_SEARCH)
    f, a, b = prog.match_device(dev_rows)
    torch.cuda.synchronize()
    if want_path is None:
        L = rows_np.shape[1]
        want_path = (_span_path(prog, L),) if _span_path(prog, L) else tuple(range(0, 32))
    assert prog.last_path() in want_path, (label, pat, prog.last_path())
    f, a, b = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
    bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
    assert bad.size == 0, (label, pat, int(bad[0]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]),
                           rows_np[bad[0]].tobytes())
    return prog, f, a, b


@pytest.mark.parametrize("L", [128, 64, 32, 16, 100, 80, 65, 48, 33, 20, 17, 12, 5, 2])
@pytest.mark.parametrize("hi_frac", [0.0, 0.01, 0.5])
def test_span_kernel_vs_oracle(fx, L, hi_frac, monkeypatch):
    """Every row kind x patterns of both kinds; a batch that ends inside a lane's span; the default grid and a two-block grid (many
    tiles per wave: the general procedure's queue overflows mid-loop); then the same rows through the one-launch kernel."""
    import torch
    K = 128 // _cell(L)
    n = 64 * K * 37 + (K + 1 if K > 2 else 1)   # not a multiple of K * 64, nor of K
    rows = _rows(L, n, 100 * L + int(hi_frac * 100), hi_frac)
    dev_rows = torch.from_numpy(rows).cuda()
    n_span = 0
    for pat in PATS_DECODE + PATS_GEN:
        gen = pat in PATS_GEN
        # (candidate-list driver programs take the span kernel at 128-byte rows only by default: FXAMD_SPAN_LENS=63 sends them there at every length)
        monkeypatch.setenv("FXAMD_SPAN_LENS", "63") if gen else monkeypatch.delenv("FXAMD_SPAN_LENS", raising=False)
        ref = oracle_lib.batch(2, pat.encode(), rows, NT)   # (one oracle pass per pattern: both grids answer the same rows)
        for blocks in ("", "2"):
            monkeypatch.setenv("FXAMD_ONE_BLOCKS", blocks) if blocks else monkeypatch.delenv("FXAMD_ONE_BLOCKS", raising=False)
            monkeypatch.delenv("FXAMD_NO_SPAN", raising=False)
            prog, f, a, b = _check(fx, pat, rows, dev_rows, None, ("span", L, hi_frac, blocks), ref)
            n_span += 1 if prog.last_path() == 18 else 0
            # flags-only calls are not the span kernel's: they must agree all the same
            ff, _, _ = prog.match_device(dev_rows, spans=False)
            torch.cuda.synchronize()
            assert np.array_equal(ff.cpu().numpy(), f), (pat, L, "flags only")
        monkeypatch.delenv("FXAMD_ONE_BLOCKS", raising=False)
        monkeypatch.setenv("FXAMD_NO_SPAN", "1")
        prog2 = fx.Program(pat, fx.OP_SEARCH)
        f2, a2, b2 = prog2.match_device(dev_rows)
        torch.cuda.synchronize()
        assert prog2.last_path() != 18, (pat, prog2.last_path())
        assert np.array_equal(f2.cpu().numpy(), f) and np.array_equal(a2.cpu().numpy(), a) and np.array_equal(b2.cpu().numpy(), b), (pat, L, "one-launch kernel")
        monkeypatch.delenv("FXAMD_NO_SPAN", raising=False)
    monkeypatch.delenv("FXAMD_SPAN_LENS", raising=False)
    assert n_span >= 2 * 12, n_span   # (the span kernel did take the patterns it is meant for)


@pytest.mark.parametrize("L", [128, 64, 32, 16, 127, 99, 50, 21, 7, 3])
def test_span_kernel_batch_ends_and_base_addresses(fx, L):
    """Every batch length around the tile's (K * 64 rows) and the span's (K rows) boundaries, 1 row included; base addresses that are
    not 16-byte aligned (unaligned tile loads, same kernel); results behind the batch's end stay untouched."""
    import torch
    K = 128 // _cell(L)
    big = _rows(L, 64 * K * 3 + 7, 4242 + L, 0.02)
    pats = [r"[a-z]+\d+", r"foo(bar|baz)", r"\d+$"]
    refs = {p: oracle_lib.batch(2, p.encode(), big, NT) for p in pats}
    lens = sorted(set([1, 2, K - 1, K, K + 1, 63, 64, 65, 64 * K - 1, 64 * K, 64 * K + 1, 64 * K + K, 2 * 64 * K - K - 1, 64 * K * 3 + 7]) - {0, -1})
    for pat in pats:
        prog = fx.Program(pat, fx.OP_SEARCH)
        of, oa, ob = refs[pat]
        for n in lens:
            for off in (0, 1, 4, 7):
                buf = torch.zeros(n * L + 32, dtype=torch.uint8, device="cuda")
                view = buf[off:off + n * L].view(n, L)
                view.copy_(torch.from_numpy(big[:n]))
                # result arrays with a guard behind the batch's end
                f = torch.full((n + 8,), 77, dtype=torch.uint8, device="cuda")
                a = torch.full((n + 8,), -5, dtype=torch.int32, device="cuda")
                b = torch.full((n + 8,), -5, dtype=torch.int32, device="cuda")
                prog.match_device(view, out=(f[:n], a[:n], b[:n]))
                torch.cuda.synchronize()
                assert _span_path(prog, L) is None or prog.last_path() == _span_path(prog, L), (pat, n, prog.last_path())
                fn, an, bn = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
                assert np.array_equal(fn[:n], of[:n]) and np.array_equal(an[:n], oa[:n]) and np.array_equal(bn[:n], ob[:n]), (pat, L, n, off)
                assert (fn[n:] == 77).all() and (an[n:] == -5).all() and (bn[n:] == -5).all(), (pat, L, n, off, "wrote behind the batch")


def test_span_kernel_generated_configs_and_handle_reuse(fx, monkeypatch):
    """BASELINE configs 5 and 2 (their generators) against the oracle; one handle alternating between the span pipeline and others
    (the counter groups of the multi-pass pipelines alternate per call); mostly-UTF-8 batches called repeatedly (the adaptive first pass)."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    for cfg, n in (("cfg5", 40000), ("cfg2", 50000)):
        rows = synth.batch(cfg, 12345, n, dev)
        pat = synth.PATTERNS[cfg]
        _check(fx, pat, rows.cpu().numpy(), rows, None, cfg)
    # one handle, row lengths 128 / 256 / 64 / 100 / 128 again, with a UTF-8 section so that the follow-ups have work
    pat = r"[a-z]+\d+"
    prog = fx.Program(pat, fx.OP_SEARCH)
    for rep in range(3):
        for L in (128, 256, 64, 100, 32, 128, 16, 20, 200, 64):
            rows = _rows(L, 64 * 30 + 5, 9000 + L + rep, 0.05 if rep != 1 else 0.0)
            of, oa, ob = oracle_lib.batch(2, pat.encode(), rows, NT)
            f, a, b = prog.match_device(torch.from_numpy(rows).cuda())
            torch.cuda.synchronize()
            assert np.array_equal(f.cpu().numpy(), of) and np.array_equal(a.cpu().numpy(), oa) and np.array_equal(b.cpu().numpy(), ob), (L, rep, prog.last_path())
    # mostly UTF-8, then ASCII, then mixed: every call equal to the oracle, with and without the adaptive skip
    n = 64 * 2 * 300
    u = torch.full((n, 128), 32, dtype=torch.uint8, device=dev)
    u[:, :120] = synth.batch("cfg4", 500, n, dev)[:, :120]
    a_rows = synth.batch("cfg5", 900, n, dev)
    mixed = a_rows.clone()
    mixed[64 * 100:64 * 300] = u[64 * 100:64 * 300]
    for hook in (None, "1"):
        monkeypatch.setenv("FXAMD_NO_ADAPT", hook) if hook else monkeypatch.delenv("FXAMD_NO_ADAPT", raising=False)
        for pat in (r"[a-z ]+\d*", r"\d+[a-z]"):
            prog = fx.Program(pat, fx.OP_SEARCH)
            for rows, calls in ((u, 12), (a_rows, 11), (mixed, 4), (u, 3), (a_rows, 2)):
                k = 3000
                of, oa, ob = oracle_lib.batch(2, pat.encode(), rows[:k].cpu().numpy(), NT)
                first = None
                for c in range(calls):
                    f, fa, fb = prog.match_device(rows)
                    torch.cuda.synchronize()
                    assert prog.last_path() == 18, (pat, prog.last_path())
                    got = (f.clone(), fa.clone(), fb.clone())
                    if first is None:
                        first = got
                        assert np.array_equal(got[0][:k].cpu().numpy(), of) and np.array_equal(got[1][:k].cpu().numpy(), oa) and np.array_equal(got[2][:k].cpu().numpy(), ob), (pat, hook, c)
                    else:
                        assert torch.equal(got[0], first[0]) and torch.equal(got[1], first[1]) and torch.equal(got[2], first[2]), (pat, hook, c)
    monkeypatch.delenv("FXAMD_NO_ADAPT", raising=False)


def test_span_kernel_fuzz_patterns(fx):
    """Generated patterns (the fuzz generator of the other tile-kernel tests) over rows of 128 and 64 bytes: whatever path a pattern
    takes, results equal the oracle's; the span kernel must be among the paths."""
    import torch
    import fuzz_diff
    seed = int(os.environ.get("FX_FUZZ_SEED", "5"))
    npat = int(os.environ.get("FX_FUZZ_PATTERNS", "60"))
    rng = random.Random(seed * 7919 + 5)
    paths = set()
    for L in (128, 64, 32, 16, 100, 24, 9):
        rows = _rows(L, 64 * (128 // _cell(L)) * 5 + 3, seed * 31 + L, 0.03)
        dev_rows = torch.from_numpy(rows).cuda()
        for _ in range(npat):
            pat = fuzz_diff.gen_pattern(rng).encode()
            prog = fx.Program(pat, fx.OP_SEARCH)
            if not prog.supported or prog.status != 0:
                continue
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            f, a, b = prog.match_device(dev_rows)
            torch.cuda.synchronize()
            paths.add(prog.last_path())
            f, a, b = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
            bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
            assert bad.size == 0, (pat, L, prog.last_path(), int(bad[0]), rows[bad[0]].tobytes(), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]),
                                   int(ob[bad[0]]))
    assert 18 in paths, paths


@pytest.mark.parametrize("L", [256, 128, 64, 32, 16, 100, 20, 6])
def test_packed_results_from_the_first_pass_kernels(fx, L, monkeypatch):
    """Round 5: the half-row first pass of 256-byte rows and the span kernel write PACKED results themselves (the wave's ballot / the lanes'
    K bits folded into bytes, narrow spans) and leave a byte per deferred tile for the follow-up instead of the rows' flag bytes.  Against
    the plain outputs of the same call, against FXAMD_NO_PACK_FIRST=1 (unpacked + fx_pack / the one-launch kernel's packing) and the torch
    implementation of the layout -- pure-ASCII batches, batches with UTF-8 and broken rows (deferred tiles: the follow-up writes their
    words), mostly-UTF-8 batches called repeatedly (the adaptive first pass), batch sizes around the tile / word boundaries."""
    import torch
    from forgex_amd import dist as fxdist
    dev = torch.device("cuda")
    K = max(1, 128 // _cell(L)) if L <= 128 else 1
    sizes = [1, 63, 64, 65, 64 * K - 1, 64 * K, 64 * K + 1, 64 * K * 5 + 3, 64 * K * 41 + 64 + 7]
    pats = [r"[a-z]+\d+", r"\d+$", r"^[a-z]+"] + ([r"aa[bc]", r"foo(bar|baz)"] if L == 128 else [])
    for hi_frac in (0.0, 0.03, 0.9):
        big = _rows(L, max(sizes), 77 * L + int(hi_frac * 100), hi_frac)
        for pat in pats:
            for n in sizes if hi_frac != 0.9 else sizes[-2:]:
                rows = torch.from_numpy(big[:n]).to(dev)
                monkeypatch.delenv("FXAMD_NO_PACK_FIRST", raising=False)
                prog = fx.Program(pat, fx.OP_SEARCH)
                f, a, b = prog.match_device(rows)
                torch.cuda.synchronize()
                path_plain = prog.last_path()
                for rep in range(3 if hi_frac == 0.9 else 1):
                    packed = prog.match_device_packed(rows, spans=True)
                    torch.cuda.synchronize()
                    assert prog.last_path() == path_plain == (16 if L == 256 else 18), (pat, L, n, prog.last_path(), path_plain)
                    off_f, off_t, total, w = fx.packed_layout(n, L, True)
                    uf, ua, ub = fx.unpack_results(packed, n, L, True)
                    torch.cuda.synchronize()
                    bad = torch.nonzero((uf != f) | (ua != a) | (ub != b))
                    assert bad.numel() == 0, (pat, L, n, hi_frac, rep, int(bad[0]), int(uf[bad[0]]), int(f[bad[0]]), int(ua[bad[0]]), int(a[bad[0]]))
                    bits, a8, b8 = fxdist.pack_results(f, a, b, L)
                    assert torch.equal(packed[:bits.numel()], bits), (pat, L, n, hi_frac)
                    assert torch.equal(packed[off_f:off_f + n * w], a8.view(torch.uint8)) and torch.equal(packed[off_t:off_t + n * w], b8.view(torch.uint8))
                monkeypatch.setenv("FXAMD_NO_PACK_FIRST", "1")
                packed2 = fx.Program(pat, fx.OP_SEARCH).match_device_packed(rows, spans=True)
                torch.cuda.synchronize()
                uf2, ua2, ub2 = fx.unpack_results(packed2, n, L, True)
                torch.cuda.synchronize()
                assert torch.equal(uf2, f) and torch.equal(ua2, a) and torch.equal(ub2, b), (pat, L, n, "FXAMD_NO_PACK_FIRST")
    monkeypatch.delenv("FXAMD_NO_PACK_FIRST", raising=False)
