"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, through the C ABI, against the oracle
(oracle/liboracle.so), the golden vectors recorded from the real reference, and -- at BASELINE.json's full sizes --
size-independent properties of the generators.  Bit-exact: flags and (from,to) are integers."""
import os

import numpy as np
import pytest

import golden
import oracle_lib

pytestmark = pytest.mark.gpu
NT = os.cpu_count() or 1


@pytest.fixture(scope="module")
def fx(built):
    import torch
    import forgex_amd
    assert torch.cuda.is_available()
    assert forgex_amd.lib().fxamd_device_count() >= 1
    return forgex_amd


def _device_run(fx, pattern, op, rows_np, spans=True):
    import torch
    prog = fx.Program(pattern, op)
    rows = torch.from_numpy(np.ascontiguousarray(rows_np)).cuda()
    f, a, b = prog.match_device(rows, spans=spans)
    torch.cuda.synchronize()
    return prog, f.cpu().numpy(), (a.cpu().numpy() if spans else None), (b.cpu().numpy() if spans else None)


def test_golden_vectors_through_c_abi(fx):
    """Every .in./.match./regex assertion of the reference's own tests, as recorded from the real reference."""
    recs = golden.load_ref_tests()
    n_checked = n_unsupported = 0
    for prog_name, kind, f in recs:
        if kind not in ("in", "match", "regex"):
            continue
        pat, txt = golden.unhx(f[0]), golden.unhx(f[1])
        op = fx.OP_MATCH if kind == "match" else fx.OP_SEARCH
        p = fx.Program(pat, op)
        if not p.supported:
            n_unsupported += 1
            continue
        rows = np.frombuffer(txt, dtype=np.uint8).reshape(1, len(txt))
        if p.status != 0:
            flag, frm, to = 0, fx.INVALID_CHAR_INDEX, fx.INVALID_CHAR_INDEX
        else:
            fl, a, b = p.match_host(rows, spans=True)
            flag, frm, to = int(fl[0]), int(a[0]), int(b[0])
        if kind == "match":
            assert (flag != 0) == (f[3] == "T"), (prog_name, pat, txt)
        elif kind == "in":
            assert (flag != 0) == (f[3] == "T"), (prog_name, pat, txt)
            assert (frm, to) == (int(f[4]), int(f[5])), (prog_name, pat, txt, frm, to)
        else:
            assert (frm, to) == (int(f[4]), int(f[5])), (prog_name, pat, txt, frm, to)
            sub = txt[frm - 1:to] if frm > 0 and to > 0 else b""
            assert sub == golden.unhx(f[3]), (prog_name, pat, txt)
        n_checked += 1
    assert n_checked > 990 and n_unsupported == 0


@pytest.mark.parametrize("cfg,n", [("cfg1", 1000), ("cfg2", 30000), ("cfg3", 20000), ("cfg4", 6000), ("cfg5", 30000)])
def test_config_rows_vs_oracle(fx, cfg, n):
    import torch
    from forgex_amd import synth
    pat = synth.PATTERNS[cfg].encode()
    rows = synth.batch(cfg, 0, n, torch.device("cuda")).cpu().numpy()
    if cfg == "cfg1":
        prog, f, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows)
        of, _, _ = oracle_lib.batch(1, pat, rows, NT)
        assert np.array_equal(f, of)
        assert 400 < int(f.sum()) < 600
        return
    prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
    of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
    assert np.array_equal(f, of), cfg
    assert np.array_equal(a, oa) and np.array_equal(b, ob), cfg
    assert int(f.sum()) > 0
    # flags-only entry (what `.in.` returns) agrees with the span entry
    _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
    assert np.array_equal(f2, of)
    if cfg in ("cfg2", "cfg3", "cfg5"):
        assert prog.last_path() in (1, 3, 8, 9, 10, 11, 12, 13, 14, 16, 18)   # tile kernels (9-11: the one-launch kernel; 18: the span kernel)


def test_fast_and_general_kernels_agree(fx, monkeypatch):
    """The same rows through the tile kernels (aligned and MISALIGNED base address: unaligned tile loads, same kernels) and through
    the general kernel (one lane per row; FXAMD_FORCE_GENERAL is the only way left to reach it with a supported pattern)."""
    import torch
    from forgex_amd import synth
    for cfg, n in (("cfg3", 8192), ("cfg2", 4096), ("cfg4", 2048), ("cfg5", 4096)):
        rows = synth.batch(cfg, 5000, n, torch.device("cuda"))
        pat = synth.PATTERNS[cfg]
        p = fx.Program(pat, fx.OP_SEARCH)
        f1, a1, b1 = p.match_device(rows)
        tile_path = p.last_path()
        assert tile_path in (1, 3, 8, 9, 10, 11, 12, 13, 14, 16, 18)
        for off in (1, 4, 7):   # base addresses that are not 16-byte aligned: still the tile kernels
            buf = torch.empty(rows.numel() + 16, dtype=torch.uint8, device=rows.device)
            wide = buf[off:off + rows.numel()].view(rows.shape)
            wide.copy_(rows)
            f2, a2, b2 = p.match_device(wide)
            assert p.last_path() == tile_path, (cfg, off, p.last_path())
            torch.cuda.synchronize()
            assert torch.equal(f1, f2) and torch.equal(a1, a2) and torch.equal(b1, b2), (cfg, off)
            m = fx.Program(pat, fx.OP_MATCH)
            g1 = m.match_device(rows, spans=False)[0]
            g2 = m.match_device(wide, spans=False)[0]
            torch.cuda.synchronize()
            assert torch.equal(g1, g2), (cfg, off)
        monkeypatch.setenv("FXAMD_FORCE_GENERAL", "1")
        f3, a3, b3 = p.match_device(rows)
        assert p.last_path() == 2
        monkeypatch.delenv("FXAMD_FORCE_GENERAL")
        torch.cuda.synchronize()
        assert torch.equal(f1, f3) and torch.equal(a1, a3) and torch.equal(b1, b3), cfg


def test_non_ascii_rows_take_the_fixup_pass(fx):
    """Rows with bytes >= 0x80 inside an otherwise fast batch are redone by the general kernel (on-device UTF-8 decode)."""
    import torch
    from forgex_amd import synth
    rows = synth.batch("cfg3", 0, 4096, torch.device("cuda")).cpu().numpy().copy()
    rng = np.random.default_rng(7)
    for i in rng.choice(rows.shape[0], 600, replace=False):
        k = int(rng.integers(0, 250))
        rows[i, k:k + 3] = np.frombuffer("あ".encode(), dtype=np.uint8)
        if i % 3 == 0:
            rows[i, 255] = 0xE3   # truncated lead byte at the very end
    pat = "[a-zぁ-ん]+\\d+".encode()
    prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
    of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
    assert np.array_equal(f, of) and np.array_equal(a, oa) and np.array_equal(b, ob)


PATTERNS_MISC = [rb"aa[bc]", rb"^abc$", rb"abc$", rb"a*", rb"b*", rb"(|^)a", rb"\s+\d", rb"[^a-z]", rb"x*$", rb"\w+@\w+", rb"ab(c|d)e", rb"ab[cd]e*f",
                 rb"foo", rb"a{2}[xy]", rb".", rb"\D\d", rb"(a|b)*c", rb"[a-c]{2,3}z"]


def test_misc_patterns_random_rows_vs_oracle(fx):
    """Quirk probes of SURVEY Appendix A (prefilter with overlapping prefix, anchors as NUL symbols, empty matches,
    literal paths) on random short rows, both operators."""
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"abcxyz019 \n\r\t@.f o", dtype=np.uint8)
    for L in (1, 3, 7, 16, 33):
        rows = alphabet[rng.integers(0, len(alphabet), size=(3000, L))]
        for pat in PATTERNS_MISC:
            prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            assert np.array_equal(f, of), (pat, L)
            assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L)
            prog, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows)
            om, _, _ = oracle_lib.batch(1, pat, rows, NT)
            assert np.array_equal(fm, om), (pat, L, "match")


def test_invalid_pattern_and_empty_batch(fx):
    rows = np.frombuffer(b"abcdabcd", dtype=np.uint8).reshape(2, 4)
    p = fx.Program(b"a(", fx.OP_SEARCH)
    assert p.status == 2
    f, a, b = p.match_host(rows)
    assert not f.any() and not a.any() and not b.any()
    assert fx.regex(b"a(", b"zz")[2:5] == (fx.INVALID_CHAR_INDEX, fx.INVALID_CHAR_INDEX, 2)
    assert fx.in_(b"b*", b"") is True and fx.in_(b"b*", b"aaa") is False
    assert fx.regex(rb"[a-z]+\d+", b"ab12  cd345")[:4] == (b"ab12", 4, 1, 4)
    assert list(fx.in_(rb"\d", [b"a1", b"bcd", b"", b"7"])) == [True, False, False, True]


def test_full_size_cfg3_properties(fx):
    """BASELINE config 3 at full size (10M x 256 B): the generator knows the answer for every row."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    n, L = synth.SHAPES["cfg3"]
    prog = fx.Program(synth.PATTERNS["cfg3"], fx.OP_SEARCH)
    step = 2_000_000
    for start in range(0, n, step):
        rows = synth.batch("cfg3", start, step, dev)
        f, a, b = prog.match_device(rows)
        idx = torch.arange(start, start + step, dtype=torch.int64, device=dev)
        r = synth._rowhash(idx, synth.SEEDS["cfg3"], 0)
        is_match = (r & 1) == 1
        nd = 1 + (synth._lsr(r, 1) % 3)
        off = 192 + (synth._lsr(r, 8) % 61)
        assert torch.equal(f != 0, is_match)
        assert torch.equal(b.to(torch.int64)[is_match], (off + nd)[is_match])          # longest end = last planted digit
        am = a.to(torch.int64)[is_match]
        rm = rows[is_match]
        ar = torch.arange(rm.shape[0], device=dev)
        assert bool(((rm[ar, am - 1] >= 97) & (rm[ar, am - 1] <= 122)).all())            # span starts on a letter ...
        prev = torch.where(am >= 2, rm[ar, (am - 2).clamp(min=0)], torch.full_like(rm[:, 0], 32))
        assert bool((prev == 32).all())                                                  # ... right after a blank or at the row start
        assert int(a[~is_match].abs().sum()) == 0 and int(b[~is_match].abs().sum()) == 0
        # no blank inside [from, off): the leftmost start of the letter run that reaches the digits
        j = torch.arange(L, device=dev)[None, :]
        inside = (j >= (am - 1)[:, None]) & (j < off[is_match][:, None])
        assert not bool(((rm == 32) & inside).any())


def test_fortran_dropin_module(fx):
    """`use forgex` from Fortran (flang): operators on scalars and rank-1 arrays, regex, regex_f -- through iso_c_binding."""
    import subprocess
    fdir = os.path.join(golden.ROOT, "forgex_amd", "fortran")
    exe = os.path.join(fdir, "build", "fortran_dropin_test")
    if not os.path.exists(exe):
        if not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
            pytest.skip("flang not available")
        subprocess.check_call(["make", "-C", fdir])
    r = subprocess.run([exe], capture_output=True, timeout=300)
    assert r.returncode == 0 and b"FORTRAN DROP-IN OK" in r.stdout, (r.stdout[-500:], r.stderr[-500:])


def test_fortran_golden_replay(fx):
    """Every record of the reference's own tests that goes through the public module (in / match / regex / validate / error: 1329 of
    the 1496) replayed by a Fortran program through `use forgex` -- the drop-in boundary itself, not ctypes -- plus a `pure` function
    and a `do concurrent` loop that call the public names (their purity is part of the reference's interface)."""
    import subprocess
    fdir = os.path.join(golden.ROOT, "forgex_amd", "fortran")
    exe = os.path.join(fdir, "build", "fortran_golden_replay")
    if not os.path.exists(exe):
        if not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
            pytest.skip("flang not available")
        subprocess.check_call(["make", "-C", fdir])
    r = subprocess.run([exe, os.path.join(golden.GOLDEN, "ref_tests.tsv")], capture_output=True, timeout=900)
    assert r.returncode == 0 and b"FORTRAN GOLDEN REPLAY OK" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])
    assert b"in 88, match 857, regex 52, validate 207, error 125" in r.stdout, r.stdout[-500:]


def test_fuzzed_patterns_through_gpu_vs_oracle(fx):
    """Random patterns x random texts (the generator that pinned the oracle to the real reference), grouped by pattern so
    that every pattern is compiled once and its texts go through the kernels as small batches -- vs the oracle CLI."""
    import fuzz_diff
    cases = [c for c in fuzz_diff.gen_cases(4242, 6000) if c[0] in "IMR"]
    expect = golden.run_protocol(golden.ORACLE_CLI, cases)
    n_checked = n_unsupported = 0
    for (op, pat, txt), exp in zip(cases, expect):
        p = fx.Program(pat, fx.OP_MATCH if op == "M" else fx.OP_SEARCH)
        if not p.supported:
            n_unsupported += 1
            continue
        rows = np.frombuffer(txt, dtype=np.uint8).reshape(1, len(txt))
        if p.status != 0:
            got = "%s F" % op if op in "IM" else "R -9999 -9999 0 %d -" % p.status
        else:
            f, a, b = p.match_host(rows, spans=True)
            if op in "IM":
                got = "%s %s" % (op, "T" if f[0] else "F")
            else:
                m = a[0] > 0 and b[0] > 0
                sub = txt[a[0] - 1:b[0]] if m else b""
                got = "R %d %d %d 0 %s" % (a[0] if m else 0, b[0] if m else 0, (b[0] - a[0] + 1) if m else 0, golden.hx(sub))
        assert got == exp, (op, pat, txt, got, exp)
        n_checked += 1
    assert n_checked > 4000 and n_unsupported == 0


def test_python_api_mirror(fx):
    """forgex_amd.in_/match/regex/regex_f/is_valid_regex behave like the reference's public names."""
    assert fx.match(rb"\d{3}-\d{4}", b"100-1002") is True and fx.match(rb"\d{3}-\d{4}", b"1234567") is False
    res = fx.match(rb"\d{3}-\d{4}", [b"100-1002", b"1234567 ", b"999-0000"])
    assert list(res) == [True, False, True]
    assert fx.regex_f(r"foo(bar|baz)", "xxfoobarbaz") == b"foobar"
    sub, length, frm, to, status, msg = fx.regex(r"foo(bar|baz)", "xxfoobarbaz")
    assert (sub, length, frm, to, status, msg) == (b"foobar", 6, 3, 8, 0, "Given pattern is valid.")
    subs, lens, frms, tos, status, _ = fx.regex(rb"[a-z]+\d+", [b"ab12  cd345", b"nothing", b"   z9"])
    assert subs == [b"ab12", b"", b"z9"] and list(frms) == [1, 0, 4] and list(tos) == [4, 0, 5] and list(lens) == [4, 0, 2]
    assert fx.is_valid_regex("[a-z") is False
    assert fx.in_("あ+", "かあああ") is True and fx.regex_f("あ+", "かあああ") == "あああ".encode()


def test_nfa_simulation_fallback_vs_oracle(fx):
    """Patterns whose DFA explodes (2^21 states) run through on-device NFA state-set simulation, both operators."""
    rng = np.random.default_rng(5)
    rows = np.frombuffer(b"ab", dtype=np.uint8)[rng.integers(0, 2, size=(600, 40))]
    pat = rb"[ab]*a[ab]{20}"
    p = fx.Program(pat, fx.OP_SEARCH)
    assert p.status == 0 and (p.info()["flags"] & 128)
    prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
    assert prog.last_path() == 4
    of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
    assert np.array_equal(f, of) and np.array_equal(a, oa) and np.array_equal(b, ob)
    prog, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows)
    om, _, _ = oracle_lib.batch(1, pat, rows, NT)
    assert np.array_equal(fm, om) and 0 < int(om.sum()) < 600


@pytest.mark.parametrize("byte_tables", [True, False])
def test_fast_kernel_fuzz_patterns_and_row_lengths(fx, byte_tables, monkeypatch):
    """(byte_tables False: FXAMD_NO_BYTE_DFA=1 keeps the deferred tiles on the decode pass instead of the byte-level tables.)
    The hot kernel under random PATTERNS: every generated pattern that qualifies for the fast path is run over batches
    of random rows at each row length the fast kernel is instantiated for (ASCII-only, and mixed with valid and broken
    UTF-8 so that the deferred-tile pass with on-device decode runs), flags and spans vs the oracle."""
    import random
    import fuzz_diff
    if not byte_tables:
        monkeypatch.setenv("FXAMD_NO_BYTE_DFA", "1")
    seed = int(os.environ.get("FX_FUZZ_SEED", "0"))   # (FX_FUZZ_SEED / FX_FUZZ_PATTERNS: longer soak runs than the default suite)
    want = int(os.environ.get("FX_FUZZ_PATTERNS", "60"))
    rng = random.Random((99 if byte_tables else 100) + 1000 * seed)
    nrng = np.random.default_rng(99 + seed)
    ascii_alpha = np.frombuffer(b"abcxyz019 .-\n\tAZ_@", dtype=np.uint8)
    pieces = [b"a", b"b", b"c", b"x", b"0", b"9", b" ", b".", "あ".encode(), "ん".encode(), "α".encode(), "ω".encode(), "é".encode(),
              b"\x80", b"\xe3\x81", b"\xc0\xaf", b"\xf0\x9f\x98\x80", b"\xff", b"-", b"\n"]
    n_fast = 0
    tried = 0
    while n_fast < want and tried < 20 * want:
        tried += 1
        pat = fuzz_diff.gen_pattern(rng).encode()
        p = fx.Program(pat, fx.OP_SEARCH)
        if p.status != 0 or not (p.info()["flags"] & (8 | 256)):   # v_perm tables or class-indexed chain tables
            continue
        n_fast += 1
        L = rng.choice([16, 32, 48, 64, 96, 128, 192, 256, 20, 36, 52, 80, 100, 132, 200, 252, 272, 512, 784,
                        17, 30, 50, 75, 99, 250, 255, 2, 3, 5, 8, 11, 15, 257, 300, 1000])   # incl. ragged (any length that is not a multiple of 16), tiny and long rows
        n = 192
        rows_a = ascii_alpha[nrng.integers(0, len(ascii_alpha), size=(n, L))]
        mixed = []
        for _ in range(n):
            buf = b""
            while len(buf) < L:
                buf += rng.choice(pieces)
            mixed.append(np.frombuffer(buf[:L], dtype=np.uint8))
        rows_m = np.stack(mixed)
        rows_few = rows_a.copy()   # a few broken rows among ASCII ones: the gathered tile of a FEW exception rows (fx_few.hpp at 192 / 256 bytes)
        rows_few[[5, 6, 70, 130, 191]] = rows_m[[0, 1, 2, 3, 4]]
        for rows in (rows_a, rows_m, np.concatenate([rows_a[:64], rows_m[:64], rows_a[64:128]]), rows_few):
            prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
            assert prog.last_path() in (1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 18, 20), (pat, L)   # (2: chain tables too large for LDS next to the tiles)
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            assert np.array_equal(f, of), (pat, L)
            assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L)
            _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
            assert np.array_equal(f2, of), (pat, L, "flags-only")
            # `.match.` of the same pattern over the same rows (tile kernel when its tables fit, else the general kernel)
            pm, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
            om, _, _ = oracle_lib.batch(1, pat, rows, NT)
            assert np.array_equal(fm, om), (pat, L, "match", pm.last_path())
    assert n_fast >= 60


def test_utf8_rows_byte_tables_and_decode_pass(fx, monkeypatch):
    """Non-ASCII rows two ways: the byte-level tables (UTF-8 composed into the automata, structurally invalid rows through the
    row-level fix-up) and, with FXAMD_NO_BYTE_DFA=1, the decode pass; both bit-exact against the oracle."""
    import torch
    from forgex_amd import synth
    nrng = np.random.default_rng(5)
    rows4 = synth.batch("cfg4", 0, 20000, torch.device("cuda")).cpu().numpy()
    pieces = [s.encode() for s in "あいうえおかんアイウαβγωé　"] + [b"a", b"z", b"0", b"7", b" ", b"\x80", b"\xe3\x81", b"\xc0\xaf", b"\xf0\x9f\x98\x80",
                                                                  b"\xff", b"\xf4\x90\x80\x80", b"\xed\xa0\x80", b"\xc3"]
    mixed = np.stack([np.frombuffer((b"".join(pieces[i] for i in nrng.integers(0, len(pieces), size=80)))[:64], dtype=np.uint8) for _ in range(6000)])
    cases = [(synth.PATTERNS["cfg4"].encode(), rows4, 8), ("[ぁ-ん]+[0-9]*".encode(), mixed, 8), (b".+", mixed, 8), ("[^a]+".encode(), mixed, 8),
             ("[ぁ-ん]{3}[ァ-ヶ]{3}[0-9]{3}".encode(), mixed, 7), (rb"\S+\d", mixed, 8), ("(あ|ア)+い".encode(), mixed, 7)]
    for pat, rows, path in cases:
        of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
        om, _, _ = oracle_lib.batch(1, pat, rows, NT)
        for bytes_on, wide in ((True, True), (True, False), (False, True), (False, False)):
            monkeypatch.delenv("FXAMD_NO_BYTE_DFA", raising=False)
            monkeypatch.delenv("FXAMD_NO_W16", raising=False)
            if not bytes_on:
                monkeypatch.setenv("FXAMD_NO_BYTE_DFA", "1")
            if not wide:
                monkeypatch.setenv("FXAMD_NO_W16", "1")   # chain tables instead of the 16-state v_perm ones
            prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
            if bytes_on:
                assert prog.info()["flags"] & 4096, pat
                assert prog.last_path() in (7, 8, 10, 11, 13, 14, 16, 18), (pat, prog.last_path())
            else:
                assert prog.last_path() in (1, 3, 5, 6, 9, 12, 16, 18), (pat, prog.last_path())
            assert np.array_equal(f, of), (pat, bytes_on)
            assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, bytes_on)
            _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
            assert np.array_equal(f2, of), (pat, bytes_on, "flags-only")
            pm, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
            assert np.array_equal(fm, om), (pat, bytes_on, "match", pm.last_path())
    monkeypatch.delenv("FXAMD_NO_BYTE_DFA", raising=False)
    monkeypatch.delenv("FXAMD_NO_W16", raising=False)


CHAIN_PATTERNS = [rb"\d{3}-\d{4}", rb"\w+@\w+\.(com|org|net)", rb"(19|20)\d\d-(0[1-9]|1[012])-(0[1-9]|[12][0-9]|3[01])",
                  rb"[a-z]{3,5}\d{2,4}x", rb"(ab|cd|ef|gh|ij)+k", "[ぁ-ん]{3}[ァ-ヶ]{3}[0-9]{3}".encode()]


@pytest.mark.parametrize("wide", [True, False])
def test_chain_scheme_patterns_vs_oracle(fx, wide, monkeypatch):
    """Automata with more than 8 states run on the same tile kernel: up to 16 states through the wide v_perm tables (two
    v_perm_b32 per byte), beyond that -- or with FXAMD_NO_W16=1 -- through class-indexed LDS chain tables."""
    if not wide:
        monkeypatch.setenv("FXAMD_NO_W16", "1")
    nrng = np.random.default_rng(17)
    alpha = np.frombuffer(b"abcdefghijkx0123456789-@._ comrgnt", dtype=np.uint8)
    for pat in CHAIN_PATTERNS:
        p = fx.Program(pat, fx.OP_SEARCH)
        assert p.status == 0 and (p.info()["flags"] & 256), pat
        for L in (32, 64, 128, 256, 80, 100):
            rows = alpha[nrng.integers(0, len(alpha), size=(4096, L))].copy()
            # plant some matches
            seeds = [b"555-1234", b"bob@mail.org", b"2024-02-29", b"abcd123x", b"abcdefk", "あいうアイウ123".encode()]
            for i in range(0, 4096, 7):
                sd = np.frombuffer(seeds[(i // 7) % len(seeds)], dtype=np.uint8)
                off = int(nrng.integers(0, L - len(sd)))
                rows[i, off:off + len(sd)] = sd
            prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
            assert prog.last_path() in (5, 6, 7, 8, 9, 11, 12, 14, 20), (pat, L, prog.last_path())   # (8: 256-byte rows, half-row first pass; 20: the span kernel as first pass)
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            assert np.array_equal(f, of), (pat, L)
            assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L)
            assert int(of.sum()) > 0


def test_match_operator_on_tile_kernel(fx):
    """`.match.` through fx_match_fast (v_perm and chain schemes), incl. the literal / prefix / suffix gate quirks."""
    nrng = np.random.default_rng(23)
    cases = [(rb"\d{3}-\d{4}", b"0123456789-", 16), (rb"[a-z]+\d+", b"abcxyz0189 ", 32), (rb"ab[cd]e*f", b"abcdef", 16),
             (rb"foo(bar|baz)x*", b"fobarzx", 16), (rb"^abc.*xyz$", b"abcxyz.\n", 32), ("[ぁ-ん]+[0-9]*".encode(), None, 48),
             # ragged row lengths (any length that is not a multiple of 16): padded with the inert symbol in LDS
             (rb"[a-z]+\d+", b"abcxyz0189 ", 20), (rb"ab[cd]e*f", b"abcdef", 36), (rb"^abc.*xyz$", b"abcxyz.\n", 100),
             (rb"[a-z]+\d+", b"abcxyz0189 ", 29), (rb"^abc.*xyz$", b"abcxyz.\n", 101), (rb"(ab|cd)+\d", b"abcd01", 255),
             (rb"(ab|cd|ef|gh|ij|kl)+\d{2}", b"abcdefghijkl01", 52), ("[ぁ-ん]+[0-9]*".encode(), None, 60)]
    for pat, alpha, L in cases:
        if alpha is None:
            pieces = [s.encode() for s in "あいうえおかん"] + [b"0", b"1", b"9", b" ", b"\x80"]
            rows = np.stack([np.frombuffer((b"".join(pieces[i] for i in nrng.integers(0, len(pieces), size=L)))[:L], dtype=np.uint8) for _ in range(3000)])
        else:
            a = np.frombuffer(alpha, dtype=np.uint8)
            rows = a[nrng.integers(0, len(a), size=(3000, L))].copy()
        # rows that actually match: fill some with canonical matches padded to L
        full = {rb"\d{3}-\d{4}": b"555-1234", rb"ab[cd]e*f": b"abc" + b"e" * 12 + b"f", rb"foo(bar|baz)x*": b"foobar" + b"x" * 10}
        if pat in full and len(full[pat]) == L:
            rows[::5] = np.frombuffer(full[pat], dtype=np.uint8)
        if pat == rb"[a-z]+\d+":
            rows[::4] = np.frombuffer((b"a" * 20 + b"1" * 12)[32 - L:], dtype=np.uint8)
        if pat == rb"ab[cd]e*f" and L == 36:
            rows[::5] = np.frombuffer(b"abc" + b"e" * 32 + b"f", dtype=np.uint8)
        if L == 52:
            rows[::3] = np.frombuffer(b"abcdefghijkl" * 4 + b"ab42"[:4], dtype=np.uint8)
        prog, f, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
        assert prog.last_path() in (9, 10, 11, 12, 13, 14, 17), (pat, prog.last_path())   # `.match.` in ONE launch (fx_search_one, MATCH; 17: fx_match_tiny, rows of 4 / 8 / 16 / 32 bytes)
        of, _, _ = oracle_lib.batch(1, pat, rows, NT)
        assert np.array_equal(f, of), pat
    # the 8-byte rows of BASELINE config 1: fx_match_tiny since round 4 (a lane takes eight whole rows)
    import torch
    from forgex_amd import synth
    rows = synth.batch("cfg1", 0, 1000, torch.device("cpu")).numpy()
    prog, f, _, _ = _device_run(fx, synth.PATTERNS["cfg1"].encode(), fx.OP_MATCH, rows, spans=False)
    of, _, _ = oracle_lib.batch(1, synth.PATTERNS["cfg1"].encode(), rows, NT)
    assert np.array_equal(f, of) and prog.last_path() == 17, prog.last_path()


def test_match_one_launch_vs_multipass_pipeline(fx, monkeypatch):
    """`.match.` in ONE launch (fx_search_one with MATCH: class-level tables on pure-ASCII tiles, byte-level tables or the in-LDS decode
    on the others, exception rows through the per-wave queues) against the multi-pass pipeline of fx_match_fast (FXAMD_MULTIPASS=1) and
    the oracle: ASCII, valid UTF-8 and 2 / 30 / 100 % structurally broken rows, every table scheme, whole-chunk and ragged rows, packed
    verdicts.  Reference: src/api_internal_m.F90:171-303."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(31)
    cases = [("cfg4", "[α-ωぁ-ん ]+", 1 << 14), ("cfg4", "[α-ωぁ-ん]+ *", 1 << 14), ("cfg4", ".+", 4096), ("cfg3", r"[a-z ]+\d*[a-z ]*", 1 << 14),
             ("cfg3", r"[a-z ]*(\d{1,3}[a-z ]*)?", 8192), ("cfg2", r"[a-z]+", 1 << 14), ("cfg2", r"[a-z]*foo(bar|baz)[a-z]*", 1 << 14),
             ("cfg5", r"([a-z]+ ?)+\d*.*", 8192), ("cfg1", r"\d{3}-\d{4}", 4000), ("cfg4", "(α|β|[ぁ-ん]|[γ-ω])+ +", 8192)]
    n_one = 0
    for cfg, pat, n in cases:
        base = synth.batch(cfg, 1000, n, dev)
        for bad_frac in (0.0, 0.02, 0.3, 1.0):
            rows = base.clone()
            if bad_frac > 0:
                L = rows.shape[1]
                sel = (torch.rand(n, generator=g) < bad_frac).to(dev)
                pos = torch.randint(0, L, (n,), generator=g).to(dev)
                val = torch.randint(0x80, 0x100, (n,), generator=g).to(torch.uint8).to(dev)
                idx = torch.arange(n, device=dev)[sel]
                rows[idx, pos[sel]] = val[sel]
            for views in (None, 52):   # the rows as they are, and the same bytes viewed as ragged rows of 52 bytes
                r = rows if views is None else rows.reshape(-1)[: (rows.numel() // views) * views].reshape(-1, views)
                for wide in (True, False):
                    monkeypatch.delenv("FXAMD_MULTIPASS", raising=False)
                    monkeypatch.delenv("FXAMD_NO_W16", raising=False)
                    if not wide:
                        monkeypatch.setenv("FXAMD_NO_W16", "1")
                    prog = fx.Program(pat, fx.OP_MATCH)
                    assert prog.status == 0, pat
                    f1, _, _ = prog.match_device(r, spans=False)
                    torch.cuda.synchronize()
                    path = prog.last_path()
                    if path in (9, 10, 11, 12, 13, 14):
                        n_one += 1
                    packed = prog.match_device_packed(r, spans=False)
                    fp, _, _ = fx.unpack_results(packed, r.shape[0], r.shape[1], False)
                    torch.cuda.synchronize()
                    monkeypatch.setenv("FXAMD_MULTIPASS", "1")
                    ref = fx.Program(pat, fx.OP_MATCH)
                    f2, _, _ = ref.match_device(r, spans=False)
                    torch.cuda.synchronize()
                    assert ref.last_path() in (1, 2, 3, 5, 6, 7, 8), (pat, ref.last_path())
                    bad = torch.nonzero(f1 != f2)
                    assert bad.numel() == 0, (cfg, pat, bad_frac, views, wide, path, int(bad[0]), int(f1[bad[0]]), int(f2[bad[0]]))
                    assert torch.equal(fp, f1), (cfg, pat, bad_frac, views, wide, "packed")
            monkeypatch.delenv("FXAMD_MULTIPASS", raising=False)
            monkeypatch.delenv("FXAMD_NO_W16", raising=False)
            k = min(n, 1500)
            of, _, _ = oracle_lib.batch(1, pat.encode() if isinstance(pat, str) else pat, rows[:k].cpu().numpy(), NT)
            assert np.array_equal(f1.cpu().numpy()[:0], of[:0])   # (shape check only: f1 is the last ragged view)
            prog = fx.Program(pat, fx.OP_MATCH)
            fo, _, _ = prog.match_device(rows[:k].contiguous(), spans=False)
            torch.cuda.synchronize()
            assert np.array_equal(fo.cpu().numpy(), of), (cfg, pat, bad_frac, "oracle")
    assert n_one >= 100, n_one


def test_literal_index_search_on_tile_kernel(fx):
    """Whole-pattern literals (`.in.` = raw-byte INDEX, forgex.F90:111-130): reverse-KMP tables on the tile kernel, raw bytes
    (no UTF-8 decode), short (v_perm) and long (chain) literals, overlapping occurrences, occurrences at both row ends."""
    nrng = np.random.default_rng(31)
    for lit, L in ((b"ab", 20), (b"abcab", 100), (b"needle in a hay", 252), (b"ab", 17), (b"abcab", 99), (b"needle in a hay", 255), ("あいう".encode(), 77), ("あいう".encode(), 84), (b"ab", 16), (b"aa", 32), (b"abcab", 64), (b"fooba", 64), (b"abcabcabx", 128), ("あいう".encode(), 96), (b"needle in a hay", 256),
                   (b"a", 16), (b"zzzzzzz", 48)):
        p = fx.Program(lit, fx.OP_SEARCH)
        assert p.status == 0 and p.info()["mode"] == 2 and (p.info()["flags"] & (8 | 256)), lit
        alpha = np.frombuffer(bytes(set(lit)) + b"xy", dtype=np.uint8)
        rows = alpha[nrng.integers(0, len(alpha), size=(3000, L))].copy()
        la = np.frombuffer(lit, dtype=np.uint8)
        rows[::9, :len(la)] = la
        rows[4::9, L - len(la):] = la
        rows[2::11] = np.frombuffer(b"q" * L, dtype=np.uint8)
        prog, f, a, b = _device_run(fx, lit, fx.OP_SEARCH, rows)
        assert prog.last_path() in (1, 5), (lit, prog.last_path())
        of, oa, ob = oracle_lib.batch(2, lit, rows, NT)
        assert np.array_equal(f, of), lit
        assert np.array_equal(a, oa) and np.array_equal(b, ob), lit
        _, f2, _, _ = _device_run(fx, lit, fx.OP_SEARCH, rows, spans=False)
        assert np.array_equal(f2, of)


def test_long_rows_on_tile_kernels(fx):
    """Rows longer than 256 bytes (a multiple of 16): the tile kernels walk them in 256-byte segments (backward pass through the
    LDS tile, forward pass from global memory); matches planted across segment borders, UTF-8 through the byte-level tables,
    structurally invalid rows through the row-level fix-up, `.match.` and literal search."""
    import random
    rng = random.Random(41)
    nrng = np.random.default_rng(41)
    alpha = np.frombuffer(b"abcxyz .-_@\n", dtype=np.uint8)
    pieces = [s.encode() for s in "あいうえおかんアイウαβγω"] + [b"a", b"z", b"0", b"7", b" ", b".", b"\x80", b"\xe3\x81", b"\xff", b"\xc3"]
    for L in (512, 1024, 2048, 272, 400, 1008, 784, 257, 300, 1000, 515):   # multiples of 256, of 16, and any other length (a shorter last segment)
        n = 1500
        rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
        seeds = [b"abc123", b"555-1234", b"needle in a hay", b"bob@mail.org", "あいう".encode(), b"zz9"]
        for i in range(0, n, 3):
            sd = np.frombuffer(seeds[(i // 3) % len(seeds)], dtype=np.uint8)
            # around the segment borders as well as anywhere
            border = 256 * int(nrng.integers(1, L // 256 + 1)) if L >= 512 else 256   # a segment border
            off = int(nrng.integers(0, L - len(sd))) if i % 2 else max(0, min(L - len(sd), border - int(nrng.integers(0, len(sd) + 1))))
            rows[i, off:off + len(sd)] = sd
        mixed = np.stack([np.frombuffer((b"".join(rng.choice(pieces) for _ in range(L)))[:L], dtype=np.uint8) for _ in range(300)])
        both = np.concatenate([rows[:700], mixed, rows[700:]])
        for pat in (rb"[a-z]+\d+", rb"\d{3}-\d{4}", "[α-ωぁ-ん]+".encode(), b"needle in a hay", rb"\w+@\w+\.[a-z]+", rb"^[a-c]", rb"[0-9]$", "[ぁ-ん]+[0-9]*".encode()):
            for data in (rows, both):
                prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, data)
                assert prog.last_path() != 2, (pat, L, prog.last_path())
                of, oa, ob = oracle_lib.batch(2, pat, data, NT)
                assert np.array_equal(f, of), (pat, L, prog.last_path())
                assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L, prog.last_path())
                _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, data, spans=False)
                assert np.array_equal(f2, of), (pat, L, "flags-only")
        # `.match.`: whole-row patterns
        full = rows.copy()
        full[::4] = np.frombuffer((b"ab" * L)[:2 * (L // 2)].ljust(L, b"a"), dtype=np.uint8)
        for pat in (rb"(ab)+", rb"[a-z .@_\n-]+", rb"(ab)*a?", "[^あ]+".encode()):
            for data in (full, np.concatenate([full[:200], mixed, full[200:400]])):
                pm, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, data, spans=False)
                assert pm.last_path() != 2, (pat, L, pm.last_path())
                om, _, _ = oracle_lib.batch(1, pat, data, NT)
                assert np.array_equal(fm, om), (pat, L, "match", pm.last_path())
                assert int(om.sum()) > 0 or pat != rb"(ab)+" or L % 2 == 1


def test_prefix_and_suffix_literal_patterns_on_tile_kernel(fx):
    """`literal.*literal` shapes: with the suffix proven a necessary ending (and no match shorter than prefix + suffix) the
    candidate-list driver of the reference equals brute force, so these patterns run on the tile kernels too."""
    nrng = np.random.default_rng(61)
    alpha = np.frombuffer(b"abcxyz01 =;-", dtype=np.uint8)
    pats = [rb"abc.*xyz", rb"id=\d+;", rb"ab[a-c]*ba", rb"x=.+;", rb"foo(bar|baz)+z", rb"a b\w+c-", rb"abc.*abc"]
    for L in (64, 256, 100, 512):
        rows = alpha[nrng.integers(0, len(alpha), size=(6000, L))].copy()
        seeds = [b"abcqqxyz", b"id=42;", b"abcacba", b"x=1;", b"foobarbazz", b"a bzzc-", b"abcabc", b"abcxyzxyz", b"id=;", b"abab a"]
        for i in range(0, 6000, 2):
            sd = np.frombuffer(seeds[(i // 2) % len(seeds)], dtype=np.uint8)
            off = int(nrng.integers(0, L - len(sd)))
            rows[i, off:off + len(sd)] = sd
        # the same rows with multi-byte characters (valid UTF-8: byte-level tables) and broken sequences (row-level fix-up) spliced in
        mixed = rows[:3000].copy()
        pieces = [np.frombuffer(x, dtype=np.uint8) for x in ("あ".encode(), "é".encode(), "ω".encode(), b"\x80", b"\xe3\x81", "\U0001F600".encode())]
        for i in range(3000):
            for _ in range(int(nrng.integers(0, 4))):
                pc = pieces[int(nrng.integers(0, len(pieces)))] if i % 3 else pieces[int(nrng.integers(0, 3))]
                off = int(nrng.integers(0, L - len(pc)))
                mixed[i, off:off + len(pc)] = pc
        for pat in pats:
            for data in (rows, mixed):
                prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, data)
                assert prog.last_path() != 2, (pat, L, prog.last_path())
                of, oa, ob = oracle_lib.batch(2, pat, data, NT)
                assert np.array_equal(f, of), (pat, L)
                assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L)
                assert int(of.sum()) > 0, pat


@pytest.mark.parametrize("shape", ["cfg3", "cfg4", "cfg2-ragged", "cfg5-many"])
def test_many_patterns_over_one_batch(fx, shape, monkeypatch):
    """fxamd_match_multi_device: an array of patterns against the same rows, results pattern-major.  Patterns on the 8-state tile
    tables share ONE pass over the rows (fx_search_multi, last_path 15) -- pure-ASCII rows, UTF-8 rows with broken sequences (tiles
    deferred to each pattern's own passes), ragged row lengths, bordered prefix literals (rows listed for the fix-up), more patterns
    than one launch takes -- the others run their own pipeline; every result against the oracle."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    # (the shared pass takes rows of up to 128 bytes -- longer rows run one pipeline per pattern, round 4 -- so the 256- and 192-byte
    #  configs are viewed as 128-byte rows: config 3's bytes as they come, config 4's first 25 five-byte slots blank-padded)
    pats = [rb"[a-z]+\d+", rb"\d{3}-\d{4}", rb"zz+", rb"[0-9]$", b"needle", rb"(ab|cd)+\d"]
    if shape == "cfg3":
        rows = synth.batch("cfg3", 0, 20000, dev).reshape(40000, 128).contiguous()
    elif shape == "cfg4":
        rows = synth.batch("cfg4", 0, 12000, dev)[:, :128].contiguous()
        rows[:, 125:] = 32
        rows[::9, 7] = 0xFF
        pats = [synth.PATTERNS["cfg4"].encode(), "[ぁ-ん]+".encode(), rb"[a-z]+", rb"aa[bc]", "ω[α-ω]".encode(), rb"\w+x"]
    elif shape == "cfg2-ragged":
        rows = synth.batch("cfg2", 0, 30000, dev).reshape(-1)[:19200 * 100].reshape(19200, 100).contiguous()
        pats = [rb"foo(bar|baz)", rb"[a-z]+\d+", rb"aa[bc]", rb"q[u-z]+", b"foobar", rb"(ab|cd)+e"]
    else:
        rows = synth.batch("cfg5", 0, 15000, dev)
        pats = [rb"[a-z]+\d+", rb"[0-9]$", b"needle", rb"(ab|cd)+\d", rb"x[yz]+\d", rb"[a-f]+ [g-z]", rb"q\d", rb"[0-9]+", rb"k+ ", rb"a.c\d", rb"zz+", rb"\d{3}-\d{4}"]
    progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
    f, a, b = fx.match_many(progs, rows)
    torch.cuda.synchronize()
    assert sum(1 for p in progs if p.last_path() == 15) >= 4, [p.last_path() for p in progs]
    f2, _, _ = fx.match_many(progs, rows, spans=False)
    torch.cuda.synchronize()
    host = rows.cpu().numpy()
    for i, p in enumerate(pats):
        of, oa, ob = oracle_lib.batch(2, p, host, NT)
        assert np.array_equal(f[i].cpu().numpy(), of), (shape, p, progs[i].last_path())
        assert np.array_equal(a[i].cpu().numpy(), oa) and np.array_equal(b[i].cpu().numpy(), ob), (shape, p)
        assert np.array_equal(f2[i].cpu().numpy(), of), (shape, p, "flags only")


@pytest.mark.parametrize("L", [128, 64, 100, 16])
def test_many_patterns_nibble_tables_share_the_pass(fx, L, monkeypatch):
    """Round 6: automata of 9..16 states (the nibble tables: `\\d{3}-\\d{4}`, an e-mail pattern, ...) CAN take part in the shared first pass of
    fxamd_match_multi_device next to the 8-state ones (last_path 15; same 4 KB of LDS per pattern) -- FXAMD_MULTI_W16=1; off by default: measured
    no faster than their own span-kernel pipelines (fxamd.hip).  Aligned and ragged rows, ASCII and mixed with valid / broken UTF-8 (tiles
    deferred to each pattern's own follow-up); every result against the oracle, against one pattern at a time, and against the default
    dispatch (the hook is live: more patterns report the shared pass with it)."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    pats = [rb"\d{3}-\d{4}", rb"[a-z]+\d+", rb"[a-z0-9._]+@[a-z0-9]+\.[a-z]+", rb"(ab|cd)+\d", rb"[a-c]{2}[0-9]{3}x", rb"zz+", rb"\d\d:\d\d:\d\d"]
    n = 24000
    flat = synth.batch("cfg3", 0, (n * L + 255) // 256, dev).reshape(-1)[: n * L]
    rows = flat.reshape(n, L).contiguous().clone()
    rows[5::17, :15] = torch.tensor(list(b"555-1234 a@b.cc"), dtype=torch.uint8, device=dev)
    rows[3::29, L // 2] = 0xE3            # a lone lead byte: broken UTF-8 in some tiles
    rows[7::31, 0:2] = torch.tensor([0xCE, 0xB1], dtype=torch.uint8, device=dev)   # valid UTF-8 in others
    progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
    singles = []
    wide = []
    for p in progs:
        singles.append(tuple(t.clone() for t in p.match_device(rows)))
        torch.cuda.synchronize()
        wide.append(p.last_path() in (5, 6, 7, 8, 11, 14, 20))   # the nibble / chain pipelines of a single call
    assert sum(wide) >= 2, [p.last_path() for p in progs]
    f0, a0, b0 = fx.match_many(progs, rows)   # the default dispatch: the shared pass for the 8-state programs only
    torch.cuda.synchronize()
    paths0 = [p.last_path() for p in progs]
    monkeypatch.setenv("FXAMD_MULTI_W16", "1")
    f, a, b = fx.match_many(progs, rows)
    torch.cuda.synchronize()
    paths = [p.last_path() for p in progs]
    assert sum(1 for w, lp in zip(wide, paths) if w and lp == 15) >= 2, paths
    assert sum(1 for lp in paths0 if lp == 15) < sum(1 for lp in paths if lp == 15), (paths, paths0)
    assert torch.equal(f, f0) and torch.equal(a, a0) and torch.equal(b, b0)
    f2, _, _ = fx.match_many(progs, rows, spans=False)
    torch.cuda.synchronize()
    monkeypatch.delenv("FXAMD_MULTI_W16", raising=False)
    k = 6000
    host = rows[:k].cpu().numpy()
    for i, p in enumerate(pats):
        sf, sa, sb = singles[i]
        assert torch.equal(f[i], sf) and torch.equal(a[i], sa) and torch.equal(b[i], sb), (L, p, paths[i])
        assert torch.equal(f2[i], sf), (L, p, "flags only")
        of, oa, ob = oracle_lib.batch(2, p, host, NT)
        assert np.array_equal(f[i][:k].cpu().numpy(), of) and np.array_equal(a[i][:k].cpu().numpy(), oa) and np.array_equal(b[i][:k].cpu().numpy(), ob), (L, p)
        if i < 3:   # (the planted phone number, word + digits, e-mail address)
            assert int(of.sum()) > 0, p


def test_many_patterns_utf8_tiles_in_the_shared_pass(fx, monkeypatch):
    """fx_search_multi brings the patterns' byte-level tables along (nibble format; forward automaton in the v_perm format where the
    program has it): tiles with bytes >= 0x80 are scanned in the shared pass instead of being deferred to a pass per pattern.  Same
    results as with the deferring shared pass (FXAMD_MULTI_NO_BYTES=1) and as one pattern at a time, on config-4 rows with 0 / 3 / 100 %
    structurally broken rows (exception rows go to each pattern's worklist pass)."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    pats = [synth.PATTERNS["cfg4"], "[ぁ-ん]+", "[α-ω][ぁ-ん]", "ん[α-ω]+", "[a-z]+", "(α|β|γ)[ぁ-ん]."]
    g = torch.Generator().manual_seed(77)
    for bad_frac in (0.0, 0.03, 1.0):
        n = 1 << 15
        # (rows of 128 bytes -- the shared pass takes nothing longer: config 4's first 25 five-byte slots, blank-padded)
        rows = synth.batch("cfg4", 3000, n, dev)[:, :128].contiguous()
        rows[:, 125:] = 32
        if bad_frac > 0:
            sel = (torch.rand(n, generator=g) < bad_frac).to(dev)
            pos = torch.randint(0, 128, (n,), generator=g).to(dev)
            val = torch.randint(0x80, 0x100, (n,), generator=g).to(torch.uint8).to(dev)
            idx = torch.arange(n, device=dev)[sel]
            rows[idx, pos[sel]] = val[sel]
        monkeypatch.delenv("FXAMD_MULTI_NO_BYTES", raising=False)
        progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
        infos = [p.info()["flags"] for p in progs]
        assert sum(1 for fl in infos if fl & (1 << 15)) >= 4   # FXP_F_BYTE_W16: byte-level nibble tables
        f, a, b = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        assert sum(1 for p in progs if p.last_path() == 15) >= 4, [p.last_path() for p in progs]
        monkeypatch.setenv("FXAMD_MULTI_NO_BYTES", "1")
        f2, a2, b2 = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        monkeypatch.delenv("FXAMD_MULTI_NO_BYTES", raising=False)
        assert torch.equal(f, f2) and torch.equal(a, a2) and torch.equal(b, b2), bad_frac
        monkeypatch.setenv("FXAMD_MULTI_INQ", "1")   # exception rows finished inside the shared pass (mixed-pattern queue per wave)
        f3, a3, b3 = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        monkeypatch.delenv("FXAMD_MULTI_INQ", raising=False)
        assert torch.equal(f, f3) and torch.equal(a, a3) and torch.equal(b, b3), bad_frac
        for i, p in enumerate(progs):
            f1, a1, b1 = p.match_device(rows)
            torch.cuda.synchronize()
            assert torch.equal(f[i], f1) and torch.equal(a[i], a1) and torch.equal(b[i], b1), (bad_frac, pats[i])
        k = 1200
        host = rows[:k].cpu().numpy()
        for i, p in enumerate(pats):
            of, oa, ob = oracle_lib.batch(2, p.encode(), host, NT)
            assert np.array_equal(f[i][:k].cpu().numpy(), of) and np.array_equal(a[i][:k].cpu().numpy(), oa) and np.array_equal(b[i][:k].cpu().numpy(), ob), (bad_frac, p)


def test_many_patterns_fuzz_groups(fx, monkeypatch):
    """fx_search_multi under random pattern GROUPS: 2..10 generated patterns (whatever path each one qualifies for: shared first pass,
    own one-launch kernel, general kernel) against the same rows -- ASCII, mixed with valid and broken UTF-8, whole-chunk and ragged
    row lengths, long rows (no shared pass) -- flags and spans of every pattern vs the oracle, and flags-only calls."""
    import random
    import torch
    import fuzz_diff
    seed = int(os.environ.get("FX_FUZZ_SEED", "0"))
    groups = int(os.environ.get("FX_FUZZ_GROUPS", "24"))
    rng = random.Random(4242 + 1000 * seed)
    nrng = np.random.default_rng(4242 + seed)
    ascii_alpha = np.frombuffer(b"abcxyz019 .-\n\tAZ_@", dtype=np.uint8)
    pieces = [b"a", b"b", b"c", b"x", b"0", b"9", b" ", b".", "あ".encode(), "ん".encode(), "α".encode(), "ω".encode(), "é".encode(),
              b"\x80", b"\xe3\x81", b"\xc0\xaf", b"\xf0\x9f\x98\x80", b"\xff", b"-", b"\n"]
    dev = torch.device("cuda")
    shared = 0
    shared_possible = 0
    for _ in range(groups):
        pats = []
        while len(pats) < rng.randint(2, 10):
            pat = fuzz_diff.gen_pattern(rng).encode()
            if fx.Program(pat, fx.OP_SEARCH).status == 0:
                pats.append(pat)
        if rng.random() < 0.3:
            pats.append(pats[0])   # the same pattern twice: the compile cache hands out ONE handle for both
        L = rng.choice([16, 32, 64, 96, 128, 192, 256, 20, 52, 100, 200, 252, 17, 75, 255, 5, 11, 300])
        n = rng.choice([64, 200, 1000])
        rows_a = ascii_alpha[nrng.integers(0, len(ascii_alpha), size=(n, L))]
        mixed = []
        for _ in range(n // 2):
            buf = b""
            while len(buf) < L:
                buf += rng.choice(pieces)
            mixed.append(np.frombuffer(buf[:L], dtype=np.uint8))
        host = np.ascontiguousarray(np.concatenate([rows_a[:n // 2], np.stack(mixed)]))
        rows = torch.from_numpy(host).to(dev)
        progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
        f, a, b = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        shared += sum(1 for p in progs if p.last_path() == 15)
        shared_possible += 1 if 2 <= L <= 128 else 0
        f2, _, _ = fx.match_many(progs, rows, spans=False)
        torch.cuda.synchronize()
        for i, p in enumerate(pats):
            of, oa, ob = oracle_lib.batch(2, p, host, NT)
            assert np.array_equal(f[i].cpu().numpy(), of), (pats, i, L, n, progs[i].last_path())
            assert np.array_equal(a[i].cpu().numpy(), oa) and np.array_equal(b[i].cpu().numpy(), ob), (pats, i, L, n, progs[i].last_path())
            assert np.array_equal(f2[i].cpu().numpy(), of), (pats, i, L, n, "flags only")
    assert shared >= shared_possible   # the shared pass did run where it may (rows of up to 128 bytes; most generated patterns fit the 8-state tables)


def test_batch_shapes_and_handle_reuse(fx):
    """ONE program handle across calls of different batch sizes, row lengths and output sets (the per-handle counter words
    alternate between calls, the worklist grows on demand): tile boundaries (n = 1, 63, 64, 65, ...), ASCII and UTF-8 rows mixed."""
    import random
    rng = random.Random(77)
    pieces = [b"a", b"b", b"z", b"0", b"7", b" ", b"-", "あ".encode(), "ω".encode(), "é".encode(), b"\x80", b"\xe3\x81", b"ab12", b"555-1234"]
    for pat in (rb"[a-z]+\d+", rb"\d{3}-\d{4}", "[ぁ-ん]+[0-9]*".encode(), rb"id=\d+"):
        ps = fx.Program(pat, fx.OP_SEARCH)
        pm = fx.Program(pat, fx.OP_MATCH)
        for n, L in ((1, 64), (63, 256), (64, 64), (65, 100), (4097, 32), (130, 512), (7, 8), (1000, 255), (64, 64), (5, 272), (2049, 256)):
            rows = np.stack([np.frombuffer((b"".join(rng.choice(pieces) for _ in range(L)))[:L].ljust(L, b"x"), dtype=np.uint8) for _ in range(n)])
            if n > 3:
                rows[::3] = np.frombuffer((b"abc123 " * L)[:L], dtype=np.uint8)   # pure-ASCII rows with matches in between
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            om, _, _ = oracle_lib.batch(1, pat, rows, NT)
            for spans in (True, False, True):
                f, a, b = ps.match_host(rows, spans=spans)
                assert np.array_equal(f, of), (pat, n, L, spans)
                if spans:
                    assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, n, L)
            fm, _, _ = pm.match_host(rows, spans=False)
            assert np.array_equal(fm, om), (pat, n, L, "match")


def test_bordered_prefix_literals_on_tile_kernel(fx):
    """Prefix literals with a border (`--x`, `aa[bc]`, `abab\\d`): the reference collects NON-overlapping occurrences, so rows where
    two occurrences overlap (`---x`, `aaab`) differ from brute force; R carries an absorbing state for exactly those rows and
    the tile kernels hand them to the general engine (`aa[bc]` .in. `aaab` is F in the reference)."""
    nrng = np.random.default_rng(73)
    alpha = np.frombuffer(b"ab-cxz01 ", dtype=np.uint8)
    for L in (64, 256, 100, 512):
        rows = alpha[nrng.integers(0, len(alpha), size=(8000, L))].copy()
        seeds = [b"--ab", b"---ab", b"aab", b"aaab", b"aaaac", b"abab1", b"ababab2", b"zz9", b"zzz9", b"abaxy", b"ababa", b"--x--", b"---x---",
                 b"b0aaxb0", b"aaab0b0", b"abay", b"ababacy"]
        for i in range(0, 8000, 2):
            sd = np.frombuffer(seeds[(i // 2) % len(seeds)], dtype=np.uint8)
            off = int(nrng.integers(0, L - len(sd)))
            rows[i, off:off + len(sd)] = sd
        for pat in (rb"--[a-z]+", rb"aa[bc]", rb"abab\d", rb"zz\d+", rb"aba[a-z]+", rb"aba[a-z]*y", rb"aa.*b0", rb"--[a-z ]+--"):
            prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
            assert prog.info()["flags"] & 0x20000, pat
            assert prog.last_path() in (3, 6, 12, 18), (pat, L, prog.last_path())   # (18: the span kernel, 64-byte rows)
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            assert np.array_equal(f, of), (pat, L)
            assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L)
            _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
            assert np.array_equal(f2, of), (pat, L, "flags-only")
            assert int(of.sum()) > 0, pat
    # the quirk itself
    prog, f, a, b = _device_run(fx, rb"aa[bc]", fx.OP_SEARCH, np.frombuffer(b"aaab".ljust(16) + b"xaab".ljust(16), dtype=np.uint8).reshape(2, 16))
    assert list(f) == [0, 1]
    # a worklist longer than one grid of the list fix-up (16384 blocks x 64 rows): every row holds an overlap witness or a byte >= 0x80
    n = 16384 * 64 + 4321
    rows = alpha[nrng.integers(0, len(alpha), size=(n, 32))].copy()
    rows[0::2, 5:8] = np.frombuffer(b"aaa", dtype=np.uint8)
    rows[1::2, 9] = 0xE9
    rows[1::4, 20:23] = np.frombuffer(b"aab", dtype=np.uint8)
    prog, f, a, b = _device_run(fx, rb"aa[bc]", fx.OP_SEARCH, rows)
    assert prog.last_path() in (3, 12)
    of, oa, ob = oracle_lib.batch(2, rb"aa[bc]", rows, NT)
    assert np.array_equal(f, of) and np.array_equal(a, oa) and np.array_equal(b, ob)
    assert 0 < int(of.sum()) < n


def test_prefix_literal_checked_per_row_on_tile_kernel(fx, monkeypatch):
    """Round 6: search programs whose prefix literal the compile-time proof does NOT cover -- not a necessary beginning of every match
    (`(}[abc]){2}\\d*c{2,}`), or bordered without an overlap state (`(\\t{3}[a-z]){2}`) -- used to run on the general kernel, a 20-30 x cliff.  They carry
    FXP_F_PREFIX_CHECK now: the one-launch kernel searches with the tables and checks per row that the start it found is one the reference's candidate list
    (src/essential/utility_m.f90:94-116, src/api_internal_m.F90:84-164) tries first; any other row with a hit is finished by the general row procedure inside
    the launch.  Random patterns whose program carries the flag + hand-made shapes, texts with several / overlapping occurrences and matches that do not start
    at an occurrence, in rows of whole chunks and ragged rows, spans and flags only, plain and packed -- against the oracle (which follows the driver)."""
    import random
    import fuzz_diff
    import fuzz_prefilter
    rng = random.Random(int(os.environ.get("FX_FUZZ_SEED", "0")) + 88)
    pool = [q.encode() for q in fuzz_prefilter.CHECK_SHAPES if fx.Program(q.encode(), fx.OP_SEARCH).info()["flags"] & (1 << 21)]
    n_hand = len(pool)
    tried = 0
    while len(pool) < n_hand + 24 and tried < 20000:
        tried += 1
        q = fuzz_diff.gen_pattern(rng).encode()
        pr = fx.Program(q, fx.OP_SEARCH)
        if pr.status == 0 and pr.info()["flags"] & (1 << 21):
            pool.append(q)
    assert n_hand >= 3 and len(pool) >= n_hand + 12, (n_hand, len(pool))
    n_exc = 0
    for pat in pool:
        for L in (16, 64, 100, 192, 256, 33):
            n = 64 * 5 + 7
            rows = np.full((n, L), 0x20, dtype=np.uint8)
            for i in range(n):
                t = b""
                while len(t) < L and rng.random() < 0.97:
                    t += fuzz_prefilter.gen_case_check(rng, [pat])[2] + rng.choice([b"", b" ", b"x", b"\n"])
                t = t[:L]
                if rng.random() < 0.8:
                    t = t.replace(b"\xce\xb1", b"ab").replace(b"\xff", b"c")   # (most rows pure ASCII: the tables' rows; the others are the general procedure's anyway)
                rows[i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
            assert prog.last_path() in (12, 13, 14), (pat, L, prog.last_path())
            bad = np.flatnonzero((f != of) | (a != oa) | (b != ob))
            assert bad.size == 0, (pat, L, int(bad[0]), bytes(rows[bad[0]]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]))
            _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
            assert np.array_equal(f2, of), (pat, L, "flags only")
            import torch
            img = prog.match_device_packed(torch.from_numpy(rows).cuda(), spans=True)
            pf, pa, pb = fx.unpack_results(img, n, L, True)
            torch.cuda.synchronize()
            assert np.array_equal(pf.cpu().numpy(), of) and np.array_equal(pa.cpu().numpy(), oa) and np.array_equal(pb.cpu().numpy(), ob), (pat, L, "packed")
            # brute force on the same rows (what the tables alone would answer): where it differs from the reference the check was needed
            n_exc += int((of == 0).sum())
    # longer rows and the multi-pass hook: the general kernel, as before
    rows = np.full((100, 400), 0x61, dtype=np.uint8)
    prog, f, a, b = _device_run(fx, pool[0], fx.OP_SEARCH, rows)
    assert prog.last_path() == 2, prog.last_path()
    monkeypatch.setenv("FXAMD_MULTIPASS", "1")
    prog, f, a, b = _device_run(fx, pool[0], fx.OP_SEARCH, rows[:, :64].copy())
    assert prog.last_path() == 2, prog.last_path()


def _hostwalk_mt():
    import ctypes
    lib = ctypes.CDLL(os.path.join(golden.ROOT, "tests", "support", "libhostwalk.so"))
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.hw_batch_mt.argtypes = [ctypes.c_char_p, i64, ctypes.c_int, vp, i64, i64, vp, vp, vp, ctypes.c_int]
    return lib


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_full_size_configs_vs_host_table_walker(fx, cfg):
    """GPU flags / from / to against the product's tables walked on the host over the FULL batch of the config
    (config 5: one rank's shard of 12.5M rows) -- every row, not a prefix of the batch.  The walker itself is pinned to the
    oracle and the real reference by the CPU tests."""
    import ctypes
    import torch
    from forgex_amd import synth
    lib = _hostwalk_mt()
    vp = ctypes.c_void_p
    dev = torch.device("cuda")
    n, L = synth.SHAPES[cfg]
    if cfg == "cfg5":
        n //= 8
    pat = synth.PATTERNS[cfg].encode()
    prog = fx.Program(pat, fx.OP_SEARCH)
    step = 2_500_000
    total = matches = 0
    for start in range(0, n, step):
        m = min(step, n - start)
        base = start if cfg != "cfg5" else 3 * n + start   # (shard 3 of the 8)
        rows = synth.batch(cfg, base, m, dev)
        f, a, b = prog.match_device(rows)
        torch.cuda.synchronize()
        host = np.ascontiguousarray(rows.cpu().numpy())
        hf, ha, hb = np.zeros(m, np.uint8), np.zeros(m, np.int32), np.zeros(m, np.int32)
        assert lib.hw_batch_mt(pat, len(pat), 0, host.ctypes.data_as(vp), m, L, hf.ctypes.data_as(vp), ha.ctypes.data_as(vp), hb.ctypes.data_as(vp), NT) == 0
        gf, ga, gb = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
        bad = np.flatnonzero((gf != hf) | (ga != ha) | (gb != hb))
        assert bad.size == 0, (cfg, start + int(bad[0]), int(gf[bad[0]]), int(ga[bad[0]]), int(gb[bad[0]]), int(hf[bad[0]]), int(ha[bad[0]]), int(hb[bad[0]]))
        total += m
        matches += int(hf.sum())
    assert total == n and 0 < matches < n


def test_program_from_blob_runs_on_the_device(fx):
    """Wire format: a program shipped as a blob -- what one rank would send another -- gives the same results on the
    device as the program it was taken from, for every table family (v_perm, wide, chain, byte-level, literal, NFA simulation)."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    cases = [("cfg3", rb"[a-z]+\d+", fx.OP_SEARCH), ("cfg4", synth.PATTERNS["cfg4"].encode(), fx.OP_SEARCH), ("cfg2", rb"foo(bar|baz)", fx.OP_SEARCH),
             ("cfg3", rb"\d{3}-\d{4}", fx.OP_SEARCH), ("cfg3", rb"[\w.]+@[\w.]+\.[a-z]{2,4}", fx.OP_SEARCH), ("cfg2", b"foobar", fx.OP_SEARCH),
             ("cfg1", rb"\d{3}-\d{4}", fx.OP_MATCH), ("cfg3", rb"[ab]*a[ab]{20}", fx.OP_SEARCH), ("cfg3", rb"aa[bc]", fx.OP_SEARCH)]
    for cfg, pat, op in cases:
        n = 5000 if b"{20}" not in pat else 300
        rows = synth.batch(cfg, 0, n, dev)
        p = fx.Program(pat, op)
        q = fx.Program.from_blob(p.blob(), op)
        spans = op == fx.OP_SEARCH
        f1, a1, b1 = p.match_device(rows, spans=spans)
        f2, a2, b2 = q.match_device(rows, spans=spans)
        torch.cuda.synchronize()
        assert torch.equal(f1, f2) and p.last_path() == q.last_path(), pat
        if spans:
            assert torch.equal(a1, a2) and torch.equal(b1, b2), pat
        of, oa, ob = oracle_lib.batch(2 if spans else 1, pat, rows.cpu().numpy(), NT)
        assert np.array_equal(f2.cpu().numpy(), of), pat
        if spans:
            assert np.array_equal(a2.cpu().numpy(), oa) and np.array_equal(b2.cpu().numpy(), ob), pat


def test_one_handle_on_several_streams_and_threads(fx):
    """Re-entrancy per handle: ONE program used from four host threads, each on its own stream, repeatedly and at
    the same time -- the per-call device scratch is kept per (device, stream), so the calls may overlap on the device."""
    import threading
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    prog = fx.Program(synth.PATTERNS["cfg4"], fx.OP_SEARCH)   # deferred tiles + worklist + decode pass: every piece of scratch is in use
    batches = [synth.batch("cfg4", 40000 * i, 40000 - 64 * i, dev) for i in range(4)]
    want = []
    for rows in batches:
        f, a, b = prog.match_device(rows)
        torch.cuda.synchronize()
        want.append((f.clone(), a.clone(), b.clone()))
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(25):
                    f, a, b = prog.match_device(batches[i])
                    st.synchronize()
                    if not (torch.equal(f, want[i][0]) and torch.equal(a, want[i][1]) and torch.equal(b, want[i][2])):
                        errs.append(i)
        except Exception as e:   # noqa
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_one_handle_on_two_devices(fx):
    """A handle used on a second GPU keeps separate tables and scratch there (round-1 advisor finding)."""
    import torch
    from forgex_amd import synth
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    prog = fx.Program(synth.PATTERNS["cfg4"], fx.OP_SEARCH)
    res = []
    for d in (0, 1, 0, 1):
        rows = synth.batch("cfg4", 0, 20000, torch.device("cuda", d))
        f, a, b = prog.match_device(rows)
        torch.cuda.synchronize(d)
        res.append((f.cpu(), a.cpu(), b.cpu()))
    for r in res[1:]:
        assert all(torch.equal(x, y) for x, y in zip(r, res[0]))


def test_host_buffer_entry_chunks_and_match_spans(fx):
    """fxamd_match_batch_host: batches larger than one chunk slot (two slots alternate, results drained per chunk), a ragged last
    chunk, and `.match.` programs leaving the caller's from / to untouched."""
    import torch
    from forgex_amd import synth
    n = 700_001   # 700001 x 256 B = 179 MB: three chunks of ~64 MB, the last one short and not a multiple of 64 rows
    rows = synth.batch("cfg3", 0, n, torch.device("cuda"))
    host = np.ascontiguousarray(rows.cpu().numpy())
    prog = fx.Program(synth.PATTERNS["cfg3"], fx.OP_SEARCH)
    f, a, b = prog.match_device(rows)
    torch.cuda.synchronize()
    for spans in (True, False, True):
        hf, ha, hb = prog.match_host(host, spans=spans)
        assert np.array_equal(hf, f.cpu().numpy())
        if spans:
            assert np.array_equal(ha, a.cpu().numpy()) and np.array_equal(hb, b.cpu().numpy())
    pm = fx.Program(rb"\d{3}-\d{4}", fx.OP_MATCH)
    small = synth.batch("cfg1", 0, 1000, torch.device("cpu")).numpy()
    import ctypes
    vp = ctypes.c_void_p
    fl = np.zeros(1000, np.uint8)
    fa = np.full(1000, 1234567, np.int32)
    fb = np.full(1000, -7654321, np.int32)
    rc = fx.lib().fxamd_match_batch_host(pm._h, small.ctypes.data_as(vp), 1000, 8, fl.ctypes.data_as(vp), fa.ctypes.data_as(vp), fb.ctypes.data_as(vp))
    assert rc == 0 and (fa == 1234567).all() and (fb == -7654321).all()
    of, _, _ = oracle_lib.batch(1, rb"\d{3}-\d{4}", small, NT)
    assert np.array_equal(fl, of)


@pytest.mark.parametrize("bad_frac", [0.0, 0.02, 0.3, 1.0])
def test_one_launch_kernel_exception_queues_vs_multipass_and_oracle(fx, bad_frac, monkeypatch):
    """fx_search_one: tiles with bytes >= 0x80 through the byte-level tables, structurally invalid rows through the per-wave queues
    (a third of the rows broken: the queues overflow every few tiles and are drained mid-loop; all rows broken: a gathered tile per
    tile) -- against the multi-pass pipeline of fx_search_fast on the same rows and against the oracle on a slice."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    n = 150_000 + 37
    for cfg, pat in (("cfg4", synth.PATTERNS["cfg4"].encode()), ("cfg4", "[ぁ-ん]+[α-ω]".encode()), ("cfg3", rb"[a-z]+\d+"), ("cfg2", rb"\w+(bar|baz)")):
        rows = synth.batch(cfg, 0, n, dev)
        g = torch.Generator(device="cpu").manual_seed(int(bad_frac * 1000) + 3)
        if bad_frac > 0:
            L = rows.shape[1]
            sel = (torch.rand(n, generator=g) < bad_frac).to(dev)
            pos = torch.randint(0, L, (n,), generator=g).to(dev)
            val = torch.randint(0x80, 0x100, (n,), generator=g).to(torch.uint8).to(dev)
            idx = torch.arange(n, device=dev)[sel]
            rows[idx, pos[sel]] = val[sel]
        for wide in (True, False):
            monkeypatch.delenv("FXAMD_MULTIPASS", raising=False)
            monkeypatch.delenv("FXAMD_NO_W16", raising=False)
            monkeypatch.setenv("FXAMD_NO_HALF", "1")   # (256-byte rows on the 8-state tables would keep the multi-pass pipeline)
            monkeypatch.setenv("FXAMD_NO_SPAN", "1")   # (... and 64-byte rows with spans the span kernel, tests/test_gpu_span.py)
            if not wide:
                monkeypatch.setenv("FXAMD_NO_W16", "1")
            prog = fx.Program(pat, fx.OP_SEARCH)
            f1, a1, b1 = prog.match_device(rows)
            torch.cuda.synchronize()
            assert prog.last_path() in (9, 10, 11, 12, 13, 14), (pat, prog.last_path())
            ff, _, _ = prog.match_device(rows, spans=False)
            torch.cuda.synchronize()
            monkeypatch.setenv("FXAMD_MULTIPASS", "1")
            monkeypatch.delenv("FXAMD_NO_HALF", raising=False)   # (so the half-row first pass meets the broken rows too)
            ref = fx.Program(pat, fx.OP_SEARCH)
            f2, a2, b2 = ref.match_device(rows)
            torch.cuda.synchronize()
            assert ref.last_path() in (1, 3, 5, 6, 7, 8), (pat, ref.last_path())
            bad = torch.nonzero((f1 != f2) | (a1 != a2) | (b1 != b2))
            assert bad.numel() == 0, (cfg, pat, bad_frac, wide, int(bad[0]), int(f1[bad[0]]), int(a1[bad[0]]), int(b1[bad[0]]), int(f2[bad[0]]), int(a2[bad[0]]), int(b2[bad[0]]))
            assert torch.equal(ff, f2), (cfg, pat, bad_frac, "flags only")
        monkeypatch.delenv("FXAMD_MULTIPASS", raising=False)
        monkeypatch.delenv("FXAMD_NO_W16", raising=False)
        monkeypatch.delenv("FXAMD_NO_HALF", raising=False)
        k = 4000
        of, oa, ob = oracle_lib.batch(2, pat, rows[:k].cpu().numpy(), NT)
        assert np.array_equal(f1[:k].cpu().numpy(), of) and np.array_equal(a1[:k].cpu().numpy(), oa) and np.array_equal(b1[:k].cpu().numpy(), ob), (cfg, pat, bad_frac)


def test_packed_results_equal_unpacked_on_every_config(fx):
    """packed results (1 bit per flag from the tile's ballot, spans narrowed to the row length) against the plain
    outputs of the same program on the same rows -- in-kernel packing (one-launch kernel), the packing kernel behind the other
    paths (256-byte rows on the half-row pipeline, long rows, `.match.`, general kernel), flags only and flags + spans, batch sizes
    around the 64-row word -- and against the torch implementation of the layout in forgex_amd.dist."""
    import torch
    from forgex_amd import synth
    from forgex_amd import dist as fxdist
    dev = torch.device("cuda")
    cases = [("cfg5", synth.PATTERNS["cfg5"].encode(), fx.OP_SEARCH, 200_003), ("cfg4", synth.PATTERNS["cfg4"].encode(), fx.OP_SEARCH, 100_001),
             ("cfg2", rb"foo(bar|baz)", fx.OP_SEARCH, 100_000), ("cfg3", rb"[a-z]+\d+", fx.OP_SEARCH, 60_001), ("cfg3", rb"\d{3}-\d{4}", fx.OP_SEARCH, 30_000),
             ("cfg1", rb"\d{3}-\d{4}", fx.OP_MATCH, 1000), ("cfg2", rb"aa[bc]", fx.OP_SEARCH, 50_000), ("cfg5", rb"[a-z]+\d+", fx.OP_SEARCH, 1),
             ("cfg5", rb"[a-z]+\d+", fx.OP_SEARCH, 63), ("cfg5", rb"[a-z]+\d+", fx.OP_SEARCH, 65), ("cfg5", rb"a(", fx.OP_SEARCH, 100)]
    in_kernel = 0
    for cfg, pat, op, n in cases:
        rows = synth.batch(cfg, 0, n, dev)
        if cfg == "cfg4":   # broken rows too: their bits arrive through the exception queues (atomic OR into the tile's word)
            rows[::7, 5] = 0xFF
        L = rows.shape[1]
        prog = fx.Program(pat, op)
        for spans in (True, False):
            if prog.status != 0:
                f = torch.zeros(n, dtype=torch.uint8, device=dev)
                a = b = torch.zeros(n, dtype=torch.int32, device=dev)
            else:
                f, a, b = prog.match_device(rows, spans=spans and op == fx.OP_SEARCH)
            packed = prog.match_device_packed(rows, spans=spans)
            torch.cuda.synchronize()
            if prog.status == 0 and prog.last_path() in (9, 10, 11, 12, 13, 14):
                in_kernel += 1
            sp = spans and op == fx.OP_SEARCH
            off_f, off_t, total, w = fx.packed_layout(n, L, sp)
            assert packed.numel() >= total and w == (0 if not sp else (1 if L <= 255 else 2))
            uf, ua, ub = fx.unpack_results(packed, n, L, sp)
            torch.cuda.synchronize()
            assert torch.equal(uf, f), (cfg, pat, n, spans)
            if sp:
                assert torch.equal(ua, a) and torch.equal(ub, b), (cfg, pat, n, spans)
                # the same image from the torch implementation of the layout
                bits, a8, b8 = fxdist.pack_results(f, a, b, L)
                assert torch.equal(packed[:bits.numel()], bits), (cfg, pat, n)
                assert torch.equal(packed[off_f:off_f + n * w], a8.view(torch.uint8)) and torch.equal(packed[off_t:off_t + n * w], b8.view(torch.uint8))
                assert fxdist.packed_layout(n, L) == (off_f, off_t, total)
    assert in_kernel >= 8


def test_one_handle_alternating_between_pipelines(fx):
    """One handle (the compile cache shares them) used in turn on shapes that take the multi-pass pipeline (256-byte rows: counter
    words, worklist) and the one-launch kernel (no counters): the counter group may only flip when a first-pass kernel runs, or the
    third call meets the first call's worklist count."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    pat = rb"[a-z]+\d+"
    prog = fx.Program(pat, fx.OP_SEARCH)

    def broken(cfg, n, every):
        rows = synth.batch(cfg, 0, n, dev)
        rows[::every, 9] = 0xE3   # a lead byte without its continuation bytes: exception rows
        rows[1::every, 17] = 0x80
        return rows
    seq = [("cfg3", 60000, 3), ("cfg5", 2000, 5), ("cfg3", 300, 2), ("cfg5", 70000, 3), ("cfg3", 64, 1), ("cfg2", 5000, 4), ("cfg3", 9000, 7)]
    for rep in range(2):
        for cfg, n, every in seq:
            rows = broken(cfg, n, every)
            for spans in (True, False):
                f, a, b = prog.match_device(rows, spans=spans)
                torch.cuda.synchronize()
                k = min(n, 3000)
                of, oa, ob = oracle_lib.batch(2, pat, rows[:k].cpu().numpy(), NT)
                assert np.array_equal(f[:k].cpu().numpy(), of), (cfg, n, spans, prog.last_path())
                if spans:
                    assert np.array_equal(a[:k].cpu().numpy(), oa) and np.array_equal(b[:k].cpu().numpy(), ob), (cfg, n)
    # the same rows through a fresh, uncached handle give the same results everywhere (not only on the oracle's slice)
    rows = broken("cfg3", 60000, 3)
    f1, a1, b1 = prog.match_device(rows)
    q = fx.Program.from_blob(prog.blob(), fx.OP_SEARCH)
    f2, a2, b2 = q.match_device(rows)
    torch.cuda.synchronize()
    assert torch.equal(f1, f2) and torch.equal(a1, a2) and torch.equal(b1, b2)


_SLICE_WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import forgex_amd as fx
from forgex_amd import synth
import oracle_lib
dev = torch.device("cuda")
checked = 0
for cfg, n, pat, op in (("cfg3", 10000, rb"[a-z]+\d+", fx.OP_SEARCH), ("cfg4", 9000, synth.PATTERNS["cfg4"].encode(), fx.OP_SEARCH),
                        ("cfg2", 12345, rb"foo(bar|baz)", fx.OP_SEARCH), ("cfg5", 7001, rb"[a-z]+", fx.OP_MATCH)):
    rows = synth.batch(cfg, 0, n, dev)
    rows[::11, 5] = 0xFF
    host = rows.cpu().numpy()
    prog = fx.Program(pat, op)
    of, oa, ob = oracle_lib.batch(2 if op == fx.OP_SEARCH else 1, pat, host, 8)
    f, a, b = prog.match_device(rows, spans=(op == fx.OP_SEARCH))
    torch.cuda.synchronize()
    assert np.array_equal(f.cpu().numpy(), of), (cfg, "flags")
    if op == fx.OP_SEARCH:
        assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(b.cpu().numpy(), ob), (cfg, "spans")
        img = prog.match_device_packed(rows, spans=True)
        uf, ua, ub = fx.unpack_results(img, n, rows.shape[1], True)
        torch.cuda.synchronize()
        assert np.array_equal(uf.cpu().numpy(), of) and np.array_equal(ua.cpu().numpy(), oa) and np.array_equal(ub.cpu().numpy(), ob), (cfg, "packed")
        progs = [prog, fx.Program(rb"[0-9]$", fx.OP_SEARCH), fx.Program(rb"zz+", fx.OP_SEARCH)]
        mf, ma, mb = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        assert np.array_equal(mf[0].cpu().numpy(), of) and np.array_equal(ma[0].cpu().numpy(), oa) and np.array_equal(mb[0].cpu().numpy(), ob), (cfg, "many")
        o2, _, _ = oracle_lib.batch(2, rb"[0-9]$", host, 8)
        assert np.array_equal(mf[1].cpu().numpy(), o2), (cfg, "many, second pattern")
    checked += 1
print("SLICES OK", checked)
"""


def test_batches_larger_than_one_enqueue_are_sliced(fx):
    """Worklists and exception queues hold 32-bit row numbers, so the entries enqueue a batch of more rows than FXAMD_SLICE_ROWS (2^30
    by default) slice by slice.  With the slice set to 4096 rows (a fresh process: the value is read once), plain, packed and
    many-pattern results of every config shape -- broken UTF-8 included, so worklists and queues are in use in every slice -- equal the oracle's."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, FXAMD_SLICE_ROWS="4096")
    r = subprocess.run([sys.executable, "-c", _SLICE_WORKER, os.path.dirname(here), os.path.join(here, "support")], capture_output=True, timeout=600, env=env)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and "SLICES OK 4" in out, out[-3000:]


def test_long_rows_last_bytes_of_a_tile(fx):
    """Rows longer than 256 bytes start at any byte; the final piece of the LAST row of a 64-row tile may reach past the tile, and the
    buffer range check drops an unaligned dword that straddles the extent -- the row's last bytes have to be re-read (a 257-byte
    row's byte 256 came back as NUL, which `$` matches).  Patterns that look at the end of the row, lengths of every residue mod 4
    and mod 16, batches that end on and off a tile boundary, `.in.` with spans, flags only and `.match.`."""
    rng = np.random.default_rng(17)
    alpha = np.frombuffer(b"abz019 \n", dtype=np.uint8)
    for L in (257, 258, 259, 260, 261, 271, 273, 300, 511, 513, 1001):
        for n in (64, 128, 100, 65, 191):
            rows = alpha[rng.integers(0, len(alpha), size=(n, L))]
            rows[:, -1] = alpha[rng.integers(0, len(alpha), size=n)]
            for pat in (rb"${2,}", rb"z$", rb"[0-9]$", rb"a\d$", rb"[a-z]+$"):
                of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
                prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
                assert np.array_equal(f, of), (pat, L, n, np.flatnonzero(f != of)[:5])
                assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L, n)
                _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
                assert np.array_equal(f2, of), (pat, L, n, "flags only", np.flatnonzero(f2 != of)[:5])
            for pat in (rb"[a-z0-9 ]+\n*[a-z0-9 \n]*z", rb".*[0-9]"):
                om, _, _ = oracle_lib.batch(1, pat, rows, NT)
                pm, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
                assert np.array_equal(fm, om), (pat, L, n, "match", np.flatnonzero(fm != om)[:5])


@pytest.mark.parametrize("hook", ["", "FXAMD_MULTIPASS", "FXAMD_NO_A8", "FXAMD_NO_BYTE_DFA", "FXAMD_NO_W16", "FXAMD_NO_HALF", "FXAMD_FORCE_GENERAL",
                                  "FXAMD_NO_TINY", "FXAMD_NO_SPEC", "FXAMD_HALF_SCH=1", "FXAMD_HALF_SCH=7", "FXAMD_NO_ADAPT", "FXAMD_NO_SPAN", "FXAMD_SPAN_LENS=31",
                                  "PACKED", "PACKED+FXAMD_NO_PACK_FIRST", "PACKED+FXAMD_NO_SPAN", "FXAMD_NO_LATCH", "PACKED+FXAMD_NO_LATCH"])
def test_config_scale_rows_vs_real_reference_fixture(fx, hook, monkeypatch):
    """tests/golden/config_rows.tsv: the REAL reference's flag / from / to (recorded in the container by
    tests/golden/make_config_goldens.py through oracle/_ref/ref_driver) on 2000-6144 rows of each BASELINE config (first and last
    rows of the batch, every shard of config 5's 8-GPU partition, 1024 corrupted rows of config 4), 4096 invalid-UTF-8 mutations at
    row lengths 64..256 under four patterns and the Appendix A quirk probes embedded in rows of 64..256 bytes.  The rows are
    regenerated here by (section, index), checked against the recorded CRC, and go through the C ABI on the device.  This is the
    comparison that does NOT share the front end with the product: reference src/api_internal_m.F90:108-164, :171-303,
    src/essential/utf8_m.f90:168-246."""
    import zlib
    import config_rows as cr
    # (every pipeline against the reference's answers: the default dispatch, the multi-pass pipelines, byte-level tables without the
    #  v_perm forward automaton, the in-LDS decode instead of byte-level tables, chain instead of nibble tables, whole-row instead of
    #  half-row staging, the general kernel)
    # (round 5: + the tiny-row kernels off, the speculative pass off, half rows for the v_perm tables only / for every table scheme's
    #  default set, the adaptive first pass off, the span kernel off / also for programs with byte-level tables -- and the fixture's
    #  round-5 sections: rows of 2..32 bytes, chain / nibble tables over 256- and 128-byte rows, a chain program over 400- and 1024-byte
    #  rows, speculative-pass rows)
    # (round 6: the half-row first pass and the span kernel walk the LATCHED format of R by default where the program has one -- configs 3 and 5 do --
    #  FXAMD_NO_LATCH: the plain format)
    # (round 6, VERDICT r05: the PACKED entry -- fxamd_match_batch_device_packed, what a multi-GPU host gathers or has written peer-direct -- against the
    #  real reference's answers too: images straight from the first passes, through fx_pack, and from the one-launch kernel; unpacked on the device)
    packed = hook.startswith("PACKED")
    if packed:
        hook = hook[7:] if "+" in hook else ""
    if hook:
        monkeypatch.setenv(hook.split("=")[0], hook.split("=")[1] if "=" in hook else "1")
    fix, crcs = cr.load_fixture(os.path.join(golden.GOLDEN, "config_rows.tsv"))
    n_rows = 0
    paths = {}

    def run(pat, op_, rows, spans):
        if not packed:
            return _device_run(fx, pat, op_, rows, spans=spans)
        import torch
        prog = fx.Program(pat, op_)
        d = torch.from_numpy(np.ascontiguousarray(rows)).cuda()
        sp = spans and op_ == fx.OP_SEARCH
        img = prog.match_device_packed(d, spans=sp)
        f, a, b = fx.unpack_results(img, rows.shape[0], rows.shape[1], sp)
        torch.cuda.synchronize()
        z = np.zeros(rows.shape[0], dtype=np.int32)
        return prog, f.cpu().numpy(), (a.cpu().numpy() if sp else z), (b.cpu().numpy() if sp else z)

    for name, op, pat, L, n, getter in cr.all_sections():
        rows = getter()
        assert cr.crc_of(rows) == crcs[name], name
        prog, f, a, b = run(pat, fx.OP_MATCH if op == "M" else fx.OP_SEARCH, rows, True)
        paths[name] = prog.last_path()   # (now: the flags-only call below runs on the same cached handle)
        want = fix[name]
        bad = np.flatnonzero(f.astype(np.int64) != want[:, 0])
        assert bad.size == 0, (name, int(bad[0]), bytes(rows[bad[0]]))
        if op == "R":
            bad = np.flatnonzero((a.astype(np.int64) != want[:, 1]) | (b.astype(np.int64) != want[:, 2]))
            assert bad.size == 0, (name, int(bad[0]), int(a[bad[0]]), int(b[bad[0]]), want[bad[0]].tolist(), bytes(rows[bad[0]]))
            # flags-only entry: the same verdicts
            _, f2, _, _ = run(pat, fx.OP_SEARCH, rows, False)
            assert np.array_equal(f2.astype(np.int64), want[:, 0]), name
        n_rows += n
    # the BASELINE configs run on the tile kernels (one-launch kernel / half-row pipeline), not on the general kernel
    if packed:
        if not hook:
            assert paths["cfg3"] == 16 and paths["cfg5"] == 18, paths   # (packed results straight from the first passes)
    elif not hook:
        assert paths["cfg3"] == 16 and paths["cfg2"] in (9, 10, 11, 12, 13, 14) and paths["cfg4"] in (10, 11) and paths["cfg5"] == 18, paths
        assert paths["tiny_M0_L8"] == 17 and paths["tab_p0_L256"] in (5, 6, 8) and paths["long_chain_L1024"] == 7, paths
    elif hook == "FXAMD_FORCE_GENERAL":
        assert set(paths.values()) == {2}, paths
    elif hook == "FXAMD_MULTIPASS":
        assert not set(paths.values()) & {9, 10, 11, 12, 13, 14}, paths
    cases = cr.probe_cases()
    blob = b"".join(p.encode() + b"\0" + op.encode() + t for p, op, t in cases)
    assert (zlib.crc32(blob) & 0xFFFFFFFF) == crcs["probes"]
    groups = {}
    for i, (pat, op, row) in enumerate(cases):
        groups.setdefault((pat, op, len(row)), []).append(i)
    for (pat, op, L), idx in groups.items():
        rows = np.frombuffer(b"".join(cases[i][2] for i in idx), dtype=np.uint8).reshape(len(idx), L)
        _, f, a, b = run(pat, fx.OP_MATCH if op == "M" else fx.OP_SEARCH, rows, True)
        want = fix["probes"][idx]
        assert np.array_equal(f.astype(np.int64), want[:, 0]), (pat, op, L, f.tolist(), want[:, 0].tolist())
        if op == "R":
            assert np.array_equal(a.astype(np.int64), want[:, 1]) and np.array_equal(b.astype(np.int64), want[:, 2]), (pat, op, L)
        n_rows += len(idx)
    assert n_rows > 44000


def test_one_launch_calls_replay_from_a_hip_graph(fx):
    """The one-launch paths (`last_path` 9-14: no host-side counter alternation, nothing allocated once the handle has its scratch) can be
    captured into a hipGraph and replayed on new row contents -- what a latency-bound caller of small batches does (a 1000-row batch is
    one ~5 us kernel; eager calls pay the launch path every time).  Search with spans, flags only, and `.match.`."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    for cfg, op, spans in (("cfg2", fx.OP_SEARCH, True), ("cfg5", fx.OP_SEARCH, True), ("cfg4", fx.OP_SEARCH, False), ("cfg1", fx.OP_MATCH, False)):
        n = 4096
        rows = synth.batch(cfg, 0, n, dev)
        prog = fx.Program(synth.PATTERNS[cfg], op)
        out = prog.match_device(rows, spans=spans)   # (first call: tables uploaded, scratch allocated, code objects loaded)
        torch.cuda.synchronize()
        # (config 1's 8-byte rows run fx_match_tiny -- a first pass with counter words -- outside a capture and the one-launch kernel inside one)
        # (... as do config 5's 128-byte rows -- and config 2's 64-byte ones where the span kernel takes them -- with its first pass + follow-up, 18)
        assert prog.last_path() in (9, 10, 11, 12, 13, 14) or (cfg == "cfg1" and prog.last_path() == 17) or (cfg in ("cfg5", "cfg2") and prog.last_path() == 18), (cfg, prog.last_path())
        assert fx.lib().fxamd_program_reserve(prog._h, n, torch.cuda.current_stream().cuda_stream) == 0
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):   # the capture stream needs its own scratch set: reserve it before capturing
            assert fx.lib().fxamd_program_reserve(prog._h, n, side.cuda_stream) == 0
            prog.match_device(rows, spans=spans, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            prog.match_device(rows, spans=spans, out=out)
        for start in (10000, 77777, 123456):
            rows.copy_(synth.batch(cfg, start, n, dev))
            g.replay()
            torch.cuda.synchronize()
            got = [t.clone() if t is not None else None for t in out]
            want = prog.match_device(rows, spans=spans)
            torch.cuda.synchronize()
            assert torch.equal(got[0], want[0]), (cfg, start)
            if spans:
                assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]), (cfg, start)


def test_multi_pass_pipelines_replay_from_a_hip_graph(fx):
    """Round 6: rows longer than 256 bytes have no one-launch kernel -- first pass + gated passes, whose counter words alternate between calls on the host.  Under
    capture the call zeroes its words with memset nodes (before its first pass, behind its last one), so the captured pipeline replays on new row contents:
    400- and 1024-byte rows (`last_path` 8 / 7), pure ASCII and with UTF-8 / broken rows mixed in (the gated passes have work), replays interleaved with
    eager calls on the same handle; search with spans, flags only, `.match.`."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(606)

    def batch(L, n, start, hi):
        flat = synth.batch("cfg3", start, (n * L + 255) // 256, dev).reshape(-1)
        rows = flat[: n * L].reshape(n, L).clone()
        if hi:   # every 37th row: a three-byte character, every 101st: a lone continuation byte
            rows[::37, 5:8] = torch.tensor([0xE3, 0x81, 0x82], dtype=torch.uint8, device=dev)
            rows[::101, L - 3] = 0x85
        return rows

    for L, pat, op, spans in ((400, r"[a-z]+\d+", fx.OP_SEARCH, True), (1024, r"[a-z]+\d+", fx.OP_SEARCH, True), (400, r"\d{3}-\d{4}|[a-z]+\d", fx.OP_SEARCH, False),
                              (1024, r"[a-z]{6}\d{1,3}[a-z ]{6}", fx.OP_SEARCH, True), (400, r"[a-z0-9 ]+", fx.OP_MATCH, False)):
        for hi in (False, True):
            n = 4096 + 17
            rows = batch(L, n, 0, hi)
            prog = fx.Program(pat, op)
            out = prog.match_device(rows, spans=spans)
            torch.cuda.synchronize()
            assert prog.last_path() in (1, 3, 5, 6, 7, 8), (L, pat, prog.last_path())
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):   # the capture stream's own scratch set, made before capturing
                assert fx.lib().fxamd_program_reserve(prog._h, n, side.cuda_stream) == 0
                prog.match_device(rows, spans=spans, out=out)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                prog.match_device(rows, spans=spans, out=out)
            for k, start in enumerate((31000, 90001, 250000, 31000)):
                rows.copy_(batch(L, n, start, hi and k != 1))   # (the second replay: a pure-ASCII batch after one with deferred tiles)
                graph.replay()
                torch.cuda.synchronize()
                got = [t.clone() if t is not None else None for t in out]
                with torch.cuda.stream(side):   # an eager call on the captured stream in between
                    want = prog.match_device(rows, spans=spans)
                torch.cuda.synchronize()
                assert torch.equal(got[0], want[0]), (L, pat, hi, start)
                if spans and op == fx.OP_SEARCH:
                    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]), (L, pat, hi, start)
            of, oa, ob = oracle_lib.batch(1 if op == fx.OP_MATCH else 2, pat.encode(), rows[:512].cpu().numpy(), NT)
            assert np.array_equal(got[0][:512].cpu().numpy(), of), (L, pat, hi)


def test_many_patterns_default_dispatch_by_row_length(fx):
    """By default the shared pass (`last_path` 15) takes rows of up to 128 bytes; longer rows run one pipeline per pattern (measured
    faster: tools/exp_multi.py).  Same results either way."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    pats = [rb"[a-z]+\d+", rb"[0-9]$", rb"(ab|cd)+\d", rb"x[yz]+\d"]
    for cfg, want_shared in (("cfg5", True), ("cfg2", True), ("cfg3", False), ("cfg4", False)):
        rows = synth.batch(cfg, 0, 8192, dev)
        progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
        f, a, b = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        paths = [p.last_path() for p in progs]
        assert (paths.count(15) == len(pats)) == want_shared and (want_shared or 15 not in paths), (cfg, paths)
        for i, p in enumerate(progs):
            f1, a1, b1 = p.match_device(rows)
            torch.cuda.synchronize()
            assert torch.equal(f[i], f1) and torch.equal(a[i], a1) and torch.equal(b[i], b1), (cfg, pats[i])


@pytest.mark.parametrize("L", [192, 256])
def test_few_exception_rows_chunk_parallel_scan(fx, L, monkeypatch):
    """The end of a wave of fx_search_one on rows of 192 / 256 bytes: a gathered tile of a FEW structurally invalid rows is scanned with
    the lanes spread over the rows' cells (fx_few.hpp: maps of the 8-state tables composed by v_perm, segmented scans over the row's
    lanes) instead of one lane per row.  Generated patterns on the 8-state tables with byte-level tables; 1..6 broken rows per tile
    (4 / 5 fit one call at 256 / 192 bytes; more take the lane-per-row scan), starts at the leading NUL, matches that run to the row's end
    and over broken bytes -- flags and spans vs the oracle, flags-only calls too."""
    import random
    import fuzz_diff
    monkeypatch.setenv("FXAMD_NO_HALF", "1")   # (256-byte rows: the one-launch kernel, not the half-row first pass)
    seed = int(os.environ.get("FX_FUZZ_SEED", "0"))   # (FX_FUZZ_SEED / FX_FUZZ_PATTERNS: soak runs)
    want = int(os.environ.get("FX_FUZZ_PATTERNS", "30"))
    rng = random.Random(7100 + L + 1000 * seed)
    nrng = np.random.default_rng(7100 + L + 1000 * seed)
    alpha = np.frombuffer(b"abcxyz019 .-\n\tAZ_@", dtype=np.uint8)
    pieces = [b"a", b"b", b"c", b"x", b"0", b"9", b" ", b".", "あ".encode(), "ん".encode(), "α".encode(), "ω".encode(), "é".encode(),
              b"\x80", b"\xe3\x81", b"\xc0\xaf", b"\xf0\x9f\x98\x80", b"\xff", b"-", b"\n", b"\xce", b"\xe3\x81\x81\x81"]
    fixed = [rb"[a-z]+\d+", "[α-ωぁ-ん]+".encode(), "[ぁ-ん]+[α-ω]".encode(), rb"^[a-c]*", rb"[^a]+", rb".+", rb"x*", "(あ|α)+.".encode(), rb"\d*$"]
    n_run = 0
    tried = 0
    while n_run < want and tried < 20 * want:
        tried += 1
        pat = fixed[tried - 1] if tried <= len(fixed) else fuzz_diff.gen_pattern(rng).encode()
        p = fx.Program(pat, fx.OP_SEARCH)
        if p.status != 0:
            continue
        fl = p.info()["flags"]
        if not (fl & 8) or not (fl & (1 << 12)):   # FXP_F_FAST_OK (8-state class-level tables) and FXP_F_BYTE_DFA
            continue
        n_run += 1
        n = 256
        rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
        # tiles 0..3 get 1, 3, 5 and 6 broken rows; a broken row = pieces (valid and invalid UTF-8) cut to the row length
        for t, cnt in enumerate((1, 3, 5, 6)):
            for r in rng.sample(range(64), cnt):
                buf = b""
                while len(buf) < L:
                    buf += rng.choice(pieces)
                rows[64 * t + r] = np.frombuffer(buf[:L], dtype=np.uint8)
        prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
        assert prog.last_path() in (9, 10, 11, 12, 13, 14), (pat, prog.last_path())   # (12-14: candidate-list driver programs, their exception rows go to the general procedure)
        of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
        bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
        assert bad.size == 0, (pat, L, int(bad[0]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]), rows[bad[0]].tobytes())
        _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
        assert np.array_equal(f2, of), (pat, L, "flags only")
    assert n_run >= 20


@pytest.mark.parametrize("L", [64, 128, 192, 256])
def test_speculative_forward_pass_vs_oracle_and_full_scan(fx, L, monkeypatch):
    """Round 4: the speculative forward pass of fx_search_one (fx_spec_forward, programs with FXP_F_SPEC_FWD): a walk of the anchored
    byte-level automaton from the row's first character decides rows whose match starts there (api_internal_m.F90:84-88, 108-155:
    the candidates are tried left to right); rows it cannot answer are queued per wave and take the backward + forward scan; a tile
    dense in them is scanned in place and pauses the pass.  Rows with the match at the first character, later, nowhere, behind
    blanks / NUL / LF, with structure errors before, inside and after the match, in sections whose share of failing rows runs from
    0 to 100 % -- on a tiny grid (many tiles per wave: queue overflow, pause and retry) and on the default one; spans and flags only;
    against the oracle and against the same kernel without the pass (FXAMD_NO_SPEC).  Patterns with and without the flag."""
    import random
    import torch
    rng = random.Random(9000 + L)
    greek = [chr(c).encode() for c in range(0x3B1, 0x3CA)]
    hira = [chr(c).encode() for c in range(0x3041, 0x3094)]
    other = ["é".encode(), "漢".encode(), b"\xf0\x9f\x98\x80", b"\xc0\xaf", b"\xe0\x80\xaf"]   # valid elsewhere, 4-byte, overlong forms
    junk = [b"\x80", b"\xbf", b"\xce", b"\xe3\x81", b"\xe3", b"\xf0\x9f", b"\xff", b"\xf8", b"\xc1"]   # structure errors

    def text(k):
        return b"".join(rng.choice(greek if rng.random() < 0.5 else hira) for _ in range(k))

    def ascii_run(k):
        return bytes(rng.choice(b"abcxyz 09-_") for _ in range(k))

    def make_row(fail):
        if not fail:   # the match starts at the first character
            kind = rng.random()
            body = text(rng.randint(1, L // 2))
            if kind < 0.15:
                body += rng.choice(junk) + text(rng.randint(0, 5))          # a structure error ends the match
            elif kind < 0.25:
                body += rng.choice(other) + text(rng.randint(0, 5))
            elif kind < 0.35:
                body = text(L)                                              # runs to the row's end (possibly cut inside a character)
        else:
            kind = rng.random()
            if kind < 0.35:
                body = ascii_run(L)                                          # no match at all
            elif kind < 0.6:
                body = ascii_run(rng.randint(1, 20)) + text(rng.randint(1, 30))   # a later match
            elif kind < 0.7:
                body = rng.choice([b" ", b"\n", b"\0", b"\r\n"]) + text(rng.randint(1, 30))
            elif kind < 0.85:
                body = rng.choice(junk) + text(rng.randint(1, 30))          # a structure error first: exception row of the full scan
            else:
                body = rng.choice(other) + text(rng.randint(0, 30)) + rng.choice(junk)
        body += ascii_run(L)
        return np.frombuffer(body[:L], dtype=np.uint8)

    shares = [0.0, 0.03, 0.1, 0.2, 0.3, 0.45, 1.0, 0.05]
    tiles = 96
    rows = np.stack([make_row(rng.random() < shares[(t // 3) % len(shares)]) for t in range(tiles) for _ in range(64)])[:64 * tiles - 17]
    rows[64 * 40:64 * 43] = np.stack([np.frombuffer(ascii_run(L), dtype=np.uint8) for _ in range(64 * 3)])   # three pure-ASCII tiles (FXP_F_NEEDS_NONASCII shortcut)
    dev_rows = torch.from_numpy(np.ascontiguousarray(rows)).cuda()
    pats = ["[α-ωぁ-ん]+", "ぁ*[α-ω]", "[α-ω]+\\d*", "[α-ωぁ-ん]+[a-c]", "[ぁ-ん]*[α-ω]+", "[α-ω][ぁ-ん]?[a-z ]*", "[α-ω]+[ぁ-ん]*", "(α|β|ぁ)[α-ωぁ-ん]{2,}", "[α-ωぁ-ん]+$", "[^a]+",
            "^[α-ω]+", "(あ|α)+."]
    n_spec = 0
    for pat in pats:
        pat = pat.encode()
        of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
        for blocks in ("2", ""):
            monkeypatch.delenv("FXAMD_NO_SPEC", raising=False)
            monkeypatch.setenv("FXAMD_NO_HALF", "1") if blocks else monkeypatch.delenv("FXAMD_NO_HALF", raising=False)   # (256-byte rows: both pipelines)
            monkeypatch.setenv("FXAMD_ONE_BLOCKS", blocks) if blocks else monkeypatch.delenv("FXAMD_ONE_BLOCKS", raising=False)
            prog = fx.Program(pat, fx.OP_SEARCH)
            n_spec += 1 if (prog.info()["flags"] & (1 << 19)) and blocks else 0   # FXP_F_SPEC_FWD
            f, a, b = prog.match_device(dev_rows)
            ff, _, _ = prog.match_device(dev_rows, spans=False)
            torch.cuda.synchronize()
            f, a, b, ff = f.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy(), ff.cpu().numpy()
            bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
            assert bad.size == 0, (pat, L, blocks, int(bad[0]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]), rows[bad[0]].tobytes())
            assert np.array_equal(ff, of), (pat, L, blocks, "flags only", int(np.nonzero(ff != of)[0][0]))
            monkeypatch.setenv("FXAMD_NO_SPEC", "1")
            f2, a2, b2 = prog.match_device(dev_rows)
            torch.cuda.synchronize()
            assert np.array_equal(f2.cpu().numpy(), f) and np.array_equal(a2.cpu().numpy(), a) and np.array_equal(b2.cpu().numpy(), b), (pat, L, blocks, "no spec")
    monkeypatch.delenv("FXAMD_NO_SPEC", raising=False)
    assert n_spec >= 4   # the pass did run for the patterns it is meant for (`[^a]+`, `^...`, `.` lack the flag: U+FFFF / the NUL survive)


def test_fortran_resident_batches(fx):
    """Round 4: type(fx_batch) of the Fortran module -- rows uploaded once, `pattern .in. batch`, `.match.`, `patterns(:) .in. batch`,
    `call regex(pattern, batch, from, to)`, fx_batch_search + fx_batch_count / fx_batch_fetch -- against the host-buffer forms of the same
    module on the same rows (reference surface: src/forgex.F90:24-54), and the rate of fx_batch_search over resident config-3-like rows
    (results left on the device), which has to be the kernels' rate, not the PCIe link's."""
    import re
    import subprocess
    fdir = os.path.join(golden.ROOT, "forgex_amd", "fortran")
    exe = os.path.join(fdir, "build", "fortran_batch_test")
    if not os.path.exists(exe):
        if not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
            pytest.skip("flang not available")
        subprocess.check_call(["make", "-C", fdir])
    r = subprocess.run([exe, str(4 * 1024 * 1024)], capture_output=True, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "FORTRAN BATCH OK" in out, (out[-1500:], r.stderr[-500:])
    m = re.search(r"RESIDENT RATE rows \d+ x 256 B\s+([0-9.]+) GB/s", out)
    assert m, out
    assert float(m.group(1)) > 2500.0, out   # (4 M x 256 B: 1 GB per call; the host-buffer forms manage about 45 GB/s)


def test_python_resident_batch_mirror(fx):
    """forgex_amd.Batch (the Python mirror of fxamd_batch_*): uploaded numpy rows and a wrapped CUDA tensor, one and several programs,
    fetch / count against match_device on the same rows."""
    import torch
    from forgex_amd import synth
    rows_t = synth.batch("cfg5", 0, 50_000, torch.device("cuda"))
    rows_np = rows_t.cpu().numpy()
    p1 = fx.Program(rb"[a-z]+\d+", fx.OP_SEARCH)
    p2 = fx.Program(rb"\d\d", fx.OP_SEARCH)
    pm = fx.Program(rb"[a-z ]+", fx.OP_MATCH)
    f1, a1, b1 = p1.match_device(rows_t)
    f2, a2, b2 = p2.match_device(rows_t)
    fm = pm.match_device(rows_t, spans=False)[0]
    torch.cuda.synchronize()
    for batch in (fx.Batch(rows_np), fx.Batch(rows_t)):
        batch.run(p1)
        f, a, b = batch.fetch()
        assert np.array_equal(f, f1.cpu().numpy()) and np.array_equal(a, a1.cpu().numpy()) and np.array_equal(b, b1.cpu().numpy())
        assert batch.count() == int(f1.sum())
        batch.run([p1, p2])
        for w, (ef, ea, eb) in enumerate(((f1, a1, b1), (f2, a2, b2))):
            f, a, b = batch.fetch(w)
            assert np.array_equal(f, ef.cpu().numpy()) and np.array_equal(a, ea.cpu().numpy()) and np.array_equal(b, eb.cpu().numpy()), w
            assert batch.count(w) == int(ef.sum())
        batch.run(pm, spans=False)
        assert np.array_equal(batch.fetch(spans=False)[0], fm.cpu().numpy())
        with pytest.raises(RuntimeError):
            batch.fetch(1)   # only one result set in the last run


def test_cache_trim_frees_idle_scratch_and_results_stay_right(fx, monkeypatch):
    """cached programs keep their device scratch between calls (a hipFree synchronises the device); the library accounts for
    it across the whole compile cache (least recently used idle programs are trimmed beyond 1 GB) and fxamd_cache_trim hands it back
    on request.  Eight patterns over a packed 256-byte batch with FXAMD_NO_PACK_FIRST=1 (that path stages 9 bytes per row in the handle's
    scratch; since round 5 the half-row first pass packs by itself and stages nothing): the trim frees at least that much, and the same
    calls give the same results afterwards."""
    import torch
    from forgex_amd import synth
    monkeypatch.setenv("FXAMD_NO_PACK_FIRST", "1")
    n = 400_000
    rows = synth.batch("cfg3", 0, n, torch.device("cuda"))
    pats = [rb"[a-z]+\d+", rb"\d+[a-z]", rb"[a-z]+ \d", rb"q[a-z]*\d", rb"\d\d+", rb"[a-z]\d[a-z]", rb"x+\d", rb"[a-f]+\d"]
    L = fx.lib()
    L.fxamd_cache_trim()
    before = []
    for p in pats:
        prog = fx.Program(p, fx.OP_SEARCH)
        img = prog.match_device_packed(rows, spans=True)
        torch.cuda.synchronize()
        before.append(img.clone())
        del prog   # (the compile cache keeps the program and its scratch)
    freed = L.fxamd_cache_trim()
    assert freed >= len(pats) * n * 9, freed
    assert L.fxamd_cache_trim() == 0
    for p, want in zip(pats, before):
        prog = fx.Program(p, fx.OP_SEARCH)
        img = prog.match_device_packed(rows, spans=True)
        torch.cuda.synchronize()
        assert torch.equal(img, want), p


def test_prefix_suffix_literals_round4_on_tile_kernel(fx):
    """Round 4 widened the proof "candidate-list driver == brute force" (compile.cpp, suffix_is_necessary_ending): prefix / suffix
    literals with NON-ASCII characters (`夢.{1,7}胡蝶` of the reference's own tests, `α.*β`, `ab.*é`) and literals that OVERLAP in the
    shortest match (`ab{2,}`: prefix `abb`, suffix `bb` -- every match must only be LONGER than the suffix) now run on the tile kernels;
    `A{1,2}bb` (a match may BE its suffix: the reference's cut-off drops it when no later occurrence follows) stays on the
    statement-for-statement driver.  Rows: the literals' own characters, other multi-byte characters, overlong encodings of the same
    code points (other bytes: the driver's INDEX does not see them), structure errors -- flags and spans vs the oracle."""
    import random
    rng = random.Random(4242)
    pats = ["夢.{1,7}胡蝶", "α.*β", "ab.*é", "é.*ab", "α[a-z]+β", "ab{2,}", "ab{2,}c", "a\\t{2,}", "(ab)+-α", "ぁ[α-ω]*ぁあ"]
    pieces = [x.encode() for x in ("夢", "胡蝶", "胡", "蝶", "α", "β", "é", "ぁ", "あ", "a", "b", "ab", "abb", "bb", "c", "\t", "\t\t", "-", " ", "x")] + \
             [b"\xc1\xa1", b"\xe0\x8e\xb1", b"\xce", b"\xb1", b"\xe5\xa4", b"\xff"]
    for L in (64, 100, 256):
        rows = []
        for _ in range(4096):
            buf = b""
            while len(buf) < L:
                buf += rng.choice(pieces) if rng.random() < 0.9 else bytes(rng.choice(b"abcxyz ") for _ in range(rng.randint(1, 9)))
            rows.append(np.frombuffer(buf[:L], dtype=np.uint8))
        data = np.stack(rows)
        for pat in pats:
            prog, f, a, b = _device_run(fx, pat.encode(), fx.OP_SEARCH, data)
            assert prog.last_path() != 2, (pat, L, prog.last_path())   # tile kernels (exception rows inside them go to the row procedure)
            of, oa, ob = oracle_lib.batch(2, pat.encode(), data, NT)
            bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
            assert bad.size == 0, (pat, L, int(bad[0]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]), data[bad[0]].tobytes())
            _, f2, _, _ = _device_run(fx, pat.encode(), fx.OP_SEARCH, data, spans=False)
            assert np.array_equal(f2, of), (pat, L, "flags only")
    # the quirk that must NOT be smoothed over: a match that is its own suffix literal
    quirk = np.frombuffer(b"Abb" + b" " * 61 + b"AbbAbb" + b" " * 58 + b"AAbb" + b" " * 60, dtype=np.uint8).reshape(3, 64)
    prog, f, a, b = _device_run(fx, rb"A{1,2}bb", fx.OP_SEARCH, quirk)
    of, oa, ob = oracle_lib.batch(2, rb"A{1,2}bb", quirk, NT)
    assert f.tolist() == of.tolist() == [0, 1, 1] and a.tolist() == oa.tolist() and b.tolist() == ob.tolist()


TINY_LENGTHS = [2, 3, 4, 5, 7, 8, 10, 12, 13, 16, 20, 21, 24, 27, 31, 32]


@pytest.mark.parametrize("L", TINY_LENGTHS)
def test_match_over_tiny_rows(fx, L, monkeypatch):
    """Round 4: `.match.` over rows of 2 to 32 bytes on fx_match_tiny (`last_path` 17: a lane takes a span of 64 / L whole rows -- 64 bytes
    for the divisors of 64, ragged spans for the other lengths; BASELINE config 1's shape is `\\d{3}-\\d{4}` over 8-byte rows) -- v_perm and nibble tables, programs with a literal / prefix /
    suffix gate (the `prefix == text => true` quirk of api_internal_m.F90:200-205 included), rows with bytes >= 0x80 (listed for the
    row-level fix-up), batches whose last lane span is partial -- against the oracle and against the one-launch kernel (FXAMD_NO_TINY)."""
    import random
    import torch
    rng = random.Random(1700 + L)
    nrng = np.random.default_rng(1700 + L)
    alpha = np.frombuffer(b"0123456789-ab cd", dtype=np.uint8)
    n = 64 * 64 * 3 + 37                      # three full trips and a partial lane span
    rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
    seeds = [b"100-1002", b"ab", b"abc", b"abd", b"aaaa", b"12345678", "é".encode() * 2, "あ".encode(), b"\xff", b"ab-0", b"abcd" * 8, b"-" * 32]
    for i in range(0, n, 3):
        sd = seeds[(i // 3) % len(seeds)][:L]
        buf = (sd + bytes(rng.choice(b"0123456789") for _ in range(L)))[:L] if i % 2 else (sd + b" " * L)[:L]
        rows[i] = np.frombuffer(buf, dtype=np.uint8)
    if L == 8:   # config 1's generator
        from forgex_amd import synth
        rows[:1000] = synth.batch("cfg1", 0, 1000, torch.device("cpu")).numpy()
    pats = [rb"\d{3}-\d{4}", rb"\d+", rb"[0-9a-d -]+", rb"ab[cd]", rb"ab(c|d)e?", rb"a{2}[ab]*", rb"\d*-?\d*", rb"abcd", "[é0-9]+".encode(), rb".+", rb"(ab|cd)+\d*"]
    n17 = 0
    for pat in pats:
        monkeypatch.delenv("FXAMD_NO_TINY", raising=False)
        prog, f, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
        n17 += 1 if prog.last_path() == 17 else 0
        of, _, _ = oracle_lib.batch(1, pat, rows, NT)
        bad = np.nonzero(f != of)[0]
        assert bad.size == 0, (pat, L, prog.last_path(), int(bad[0]), int(f[bad[0]]), int(of[bad[0]]), rows[bad[0]].tobytes())
        monkeypatch.setenv("FXAMD_NO_TINY", "1")
        prog2, f2, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
        assert prog2.last_path() != 17 and np.array_equal(f2, f), (pat, L, prog2.last_path())
    monkeypatch.delenv("FXAMD_NO_TINY", raising=False)
    assert n17 >= 8, n17   # (programs whose automaton fits neither the v_perm nor the nibble tables keep the other kernels)


@pytest.mark.parametrize("L", TINY_LENGTHS)
def test_in_verdict_over_tiny_rows(fx, L, monkeypatch):
    """Round 4: the `.in.` VERDICT (flags only -- what the reference's operator returns, forgex.F90:74-160) over rows of 2 to 32 bytes
    on fx_search_tiny (`last_path` 17): per row the reverse automaton from the row's last byte, a hit inside the text = TRUE; a start at the
    leading NUL is the leftmost one and takes the forward walk's answer (`^`-anchored patterns: max_match > 2, api_internal_m.F90:140-148);
    rows with bytes >= 0x80 and overlap rows of bordered prefix literals go to the row-level fix-up.  Against the oracle and the one-launch
    kernel (FXAMD_NO_TINY); the span entry keeps the other kernels."""
    import random
    rng = random.Random(1800 + L)
    nrng = np.random.default_rng(1800 + L)
    alpha = np.frombuffer(b"0123456789-ab cd\nxa", dtype=np.uint8)
    n = 64 * 64 * 2 + 53
    rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
    seeds = [b"100-1002", b"ab", b"aab", b"aaab", b"--a", b"---a", b"12", "é".encode() * 2, "あ1".encode(), b"\xffa", b"ab\n", b"\nab", b"x" * 32]
    for i in range(0, n, 3):
        sd = seeds[(i // 3) % len(seeds)][:L]
        off = rng.randint(0, L - len(sd))
        rows[i, off:off + len(sd)] = np.frombuffer(sd, dtype=np.uint8)
    pats = [rb"\d{3}-\d{4}", rb"[a-d]+\d+", rb"^\d+", rb"\d+$", rb"^ab", rb"^", rb"$", rb"^$", rb"aa[bc]", rb"--[a-z]+", rb"a b|cd", rb"x*", rb"[^0-9]+x", "é+".encode(), rb"ab$"]
    n17 = 0
    for pat in pats:
        monkeypatch.delenv("FXAMD_NO_TINY", raising=False)
        prog, f, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
        n17 += 1 if prog.last_path() == 17 else 0
        of, _, _ = oracle_lib.batch(0, pat, rows, NT)
        bad = np.nonzero(f != of)[0]
        assert bad.size == 0, (pat, L, prog.last_path(), int(bad[0]), int(f[bad[0]]), int(of[bad[0]]), rows[bad[0]].tobytes())
        # the span entry (regex) on the same rows: other kernels, same verdicts
        prog_s, fs, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=True)
        assert prog_s.last_path() != 17 and np.array_equal(fs, of), (pat, L, prog_s.last_path())
        monkeypatch.setenv("FXAMD_NO_TINY", "1")
        prog2, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
        assert prog2.last_path() != 17 and np.array_equal(f2, f), (pat, L, prog2.last_path())
    monkeypatch.delenv("FXAMD_NO_TINY", raising=False)
    assert n17 >= 10, n17


def test_tiny_rows_every_length_and_batch_end(fx):
    """Every row length 2..32 on the tiny-row kernels with batch sizes that end inside a lane's span, inside a tile, on a tile boundary and
    one row behind it (the ragged span loader's byte extent), `.match.` and the `.in.` verdict against the oracle."""
    nrng = np.random.default_rng(1900)
    alpha = np.frombuffer(b"0123456789-ab", dtype=np.uint8)
    for L in range(2, 33):
        rpl = 64 // L
        for n in (1, rpl - 1 if rpl > 1 else 2, rpl + 1, 64 * rpl - 1, 64 * rpl, 64 * rpl + 1, 64 * rpl * 9 + rpl // 2 + 1):
            rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
            rows[::5, :2] = np.frombuffer(b"ab", dtype=np.uint8)
            for op, kind, pat in ((fx.OP_MATCH, 1, rb"ab[0-9-]*"), (fx.OP_SEARCH, 0, rb"^ab\d|\d-\d"), (fx.OP_SEARCH, 0, rb"b\d+$")):
                prog, f, _, _ = _device_run(fx, pat, op, rows, spans=False)
                assert prog.last_path() == 17, (L, n, pat, prog.last_path())
                of, _, _ = oracle_lib.batch(kind, pat, rows, NT)
                bad = np.nonzero(f != of)[0]
                assert bad.size == 0, (pat, L, n, int(bad[0]), int(f[bad[0]]), int(of[bad[0]]), rows[bad[0]].tobytes())


def test_tiny_rows_fuzz_patterns(fx):
    """Random patterns (the generator that pinned the oracle to the real reference) over rows of every length 2..32, ASCII rows and rows mixed
    with valid and broken UTF-8 (listed for the row-level fix-up), `.match.` and the `.in.` verdict against the oracle; batch sizes around
    the lane-span and tile boundaries.  (FX_FUZZ_SEED / FX_FUZZ_PATTERNS: longer soak runs.)"""
    import random
    import fuzz_diff
    seed = int(os.environ.get("FX_FUZZ_SEED", "0"))
    want = int(os.environ.get("FX_FUZZ_PATTERNS", "60"))
    rng = random.Random(2100 + 1000 * seed)
    nrng = np.random.default_rng(2100 + seed)
    ascii_alpha = np.frombuffer(b"abcxyz019 .-\n\tAZ_@", dtype=np.uint8)
    pieces = [b"a", b"b", b"c", b"x", b"0", b"9", b" ", b".", "あ".encode(), "α".encode(), "é".encode(), b"\x80", b"\xe3\x81", b"\xff", b"-", b"\n"]
    n17 = done = tried = 0
    while done < want and tried < 20 * want:
        tried += 1
        pat = fuzz_diff.gen_pattern(rng).encode()
        if fx.Program(pat, fx.OP_SEARCH).status != 0:
            continue
        done += 1
        L = rng.randint(2, 32)
        rpl = 64 // L
        n = rng.choice([1, rpl, 64 * rpl - 1, 64 * rpl + 1, 64 * rpl * 4 + rng.randint(0, 64 * rpl), 64 * rpl * 5])
        rows = ascii_alpha[nrng.integers(0, len(ascii_alpha), size=(n, L))].copy()
        for i in range(0, n, 7):   # every seventh row: UTF-8 pieces, whole and broken
            buf = b""
            while len(buf) < L:
                buf += rng.choice(pieces)
            rows[i] = np.frombuffer(buf[:L], dtype=np.uint8)
        for op, kind in ((fx.OP_MATCH, 1), (fx.OP_SEARCH, 0)):
            prog, f, _, _ = _device_run(fx, pat, op, rows, spans=False)
            n17 += 1 if prog.last_path() == 17 else 0
            of, _, _ = oracle_lib.batch(kind, pat, rows, NT)
            bad = np.nonzero(f != of)[0]
            assert bad.size == 0, (pat, L, n, op, prog.last_path(), int(bad[0]), int(f[bad[0]]), int(of[bad[0]]), rows[bad[0]].tobytes())
    assert done >= want and n17 >= want, (done, n17)   # (programs without class-level v_perm / nibble tables keep the other kernels)


def test_many_patterns_from_several_streams_and_threads(fx):
    """the side streams of the per-pattern follow-ups are kept per (device, caller stream).  Four host threads, each on its own
    stream, run the same group of UTF-8 patterns (the same cached handles) over their own batches with broken rows -- every pattern has
    follow-up work -- at the same time and repeatedly; results equal the single-threaded ones."""
    import threading
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    pats = [synth.PATTERNS["cfg4"], "[ぁ-ん]+", "[α-ω][ぁ-ん]", "ん[α-ω]+", "[a-z]+"]
    progs = [fx.Program(p, fx.OP_SEARCH) for p in pats]
    g = torch.Generator().manual_seed(78)
    batches = []
    for i in range(4):
        n = 20000 - 64 * i
        rows = synth.batch("cfg4", 50000 * i, n, dev)[:, :128].contiguous()
        rows[:, 125:] = 32
        sel = (torch.rand(n, generator=g) < 0.05).to(dev)
        pos = torch.randint(0, 128, (n,), generator=g).to(dev)
        idx = torch.arange(n, device=dev)[sel]
        rows[idx, pos[sel]] = 0xFF
        batches.append(rows)
    want = []
    for rows in batches:
        f, a, b = fx.match_many(progs, rows)
        torch.cuda.synchronize()
        want.append((f.clone(), a.clone(), b.clone()))
    assert sum(1 for p in progs if p.last_path() == 15) >= 4
    errs = []

    def work(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(20):
                    f, a, b = fx.match_many(progs, batches[i])
                    st.synchronize()
                    if not (torch.equal(f, want[i][0]) and torch.equal(a, want[i][1]) and torch.equal(b, want[i][2])):
                        errs.append(i)
        except Exception as e:   # noqa
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_half_rows_for_chain_and_nibble_tables(fx, monkeypatch):
    """Round 4: 256-byte rows of programs on the chain tables and the nibble tables (spans and flags only), and 128-byte rows of programs on
    the chain tables, take the multi-pass pipeline whose first pass stages HALF rows (four waves per SIMD; `last_path` 8 / 5 / 6) --
    pure-ASCII batches, batches with UTF-8 tiles (the byte-level pass over the tiles the first pass deferred) and with structurally broken
    rows (worklist), matches in either half, across the half boundary, at the row's first and last byte -- against the oracle and against
    the one-launch kernel (FXAMD_HALF_SCH=1)."""
    import random
    rng = random.Random(2300)
    nrng = np.random.default_rng(2300)
    alpha = np.frombuffer(b"abcdefghijkx0123456789-@._ comrgnt", dtype=np.uint8)
    seeds = [b"555-1234", b"bob@mail.org", b"2024-02-29", b"abcdef123abc de", b"abcdefk", "あいうアイウ123".encode(), b"carol@example.com"]
    pats = [rb"\d{3}-\d{4}", rb"[a-z0-9]+@[a-z0-9]+\.[a-z]{2,4}", rb"[a-z]{6}\d{1,3}[a-z ]{6}", rb"\d{4}-\d{2}-\d{2}", "[ぁ-ん]{3}[ァ-ヶ]{3}\\d+".encode(), rb"^[a-k]{5}.*\d{2}$",
            rb"(19|20)\d\d-(0[1-9]|1[012])-(0[1-9]|[12][0-9]|3[01])"]
    n_half = {256: 0, 128: 0}
    n_span20 = [0]
    for L in (256, 128):
        n = 64 * 37 + 11
        H = L // 2
        for kind in ("ascii", "utf8", "broken"):
            rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
            for i in range(0, n, 3):
                sd = np.frombuffer(seeds[(i // 3) % len(seeds)], dtype=np.uint8)
                where = (i // 3) % 5
                off = [0, L - len(sd), H - len(sd) // 2, rng.randint(0, H - 8 - len(sd)), rng.randint(H + 2, L - len(sd))][where]
                rows[i, off:off + len(sd)] = sd
            if kind != "ascii":
                for i in range(5, n, 11):   # a UTF-8 character somewhere (the tile is deferred to the byte-level pass)
                    c = np.frombuffer("んω€".encode(), dtype=np.uint8)
                    off = rng.randint(0, L - len(c))
                    rows[i, off:off + len(c)] = c
            if kind == "broken":
                for i in range(7, n, 29):
                    rows[i, rng.randint(0, L - 1)] = rng.choice([0x80, 0xC0, 0xFF, 0xE3])
            for pat in pats:
                of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
                for spans in (True, False):
                    monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)
                    # (round 5: 128-byte rows with spans take the span kernel as their first pass -- last_path 20; FXAMD_SPAN_LENS without bit 6
                    #  keeps round 4's dispatch, the 64-byte halves of the chain tables)
                    for lens in ("", "47"):
                        monkeypatch.setenv("FXAMD_SPAN_LENS", lens) if lens else monkeypatch.delenv("FXAMD_SPAN_LENS", raising=False)
                        prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=spans)
                        fl = prog.info()["flags"]
                        if not (fl & 8) and (fl & ((1 << 13) | 256)):   # no 8-state tables; nibble tables (taken when present) or chain tables
                            # (FXP_F_NEEDS_NONASCII programs keep the one-launch kernel; 128-byte rows: the chain tables only)
                            want_half = not (fl & (1 << 20)) and (L == 256 or not (fl & (1 << 13)))
                            if L == 128 and spans and not lens and not (fl & (1 << 20)):
                                assert prog.last_path() in (20, 5, 6, 8), (pat, L, kind, prog.last_path())   # (5 / 6 / 8: chain tables too large for the span kernel's LDS)
                                n_span20[0] += 1 if prog.last_path() == 20 else 0
                            else:
                                assert (prog.last_path() in (5, 6, 8)) == want_half, (pat, L, kind, spans, lens, prog.last_path())
                            n_half[L] += 1 if (want_half and lens) or L == 256 else 0
                        bad0 = np.nonzero(f != of)[0]
                        assert bad0.size == 0, (pat, L, kind, spans, lens, int(bad0[0]))
                        if spans:
                            assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L, kind, lens)
                    monkeypatch.delenv("FXAMD_SPAN_LENS", raising=False)
                    bad = np.nonzero(f != of)[0]
                    assert bad.size == 0, (pat, L, kind, spans, int(bad[0]), int(f[bad[0]]), int(of[bad[0]]))
                    if spans:
                        assert np.array_equal(a, oa) and np.array_equal(b, ob), (pat, L, kind)
                    monkeypatch.setenv("FXAMD_HALF_SCH", "1")
                    prog2, f2, a2, b2 = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=spans)
                    assert prog2.last_path() not in (5, 6, 8) and np.array_equal(f2, of), (pat, L, kind, spans, prog2.last_path())
    monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)
    assert n_half[256] >= 24 and n_half[128] >= 12 and n_span20[0] >= 6, (n_half, n_span20)


def test_match_half_rows_for_chain_tables(fx, monkeypatch):
    """Round 4: `.match.` over 256-byte rows of programs on the chain tables takes the multi-pass pipeline with a half-row first
    pass (fx_match_fast<8,...,LONG>: four waves per SIMD; `last_path` 5 / 8) -- pure-ASCII batches, UTF-8 tiles (byte-level pass over the
    deferred tiles), broken rows (worklist), rows that fail in the first / second half and at the last byte -- against the oracle and the
    one-launch kernel (FXAMD_HALF_SCH=1)."""
    import random
    rng = random.Random(2400)
    nrng = np.random.default_rng(2400)
    pats = [rb"[a-z ]{6}[a-z ]*\d{0,3}[a-z ]{6}[a-z ]*", rb"[a-z ]{12}[a-z ]*\d*[a-z ]{6}", "[a-zぁ-ん ]{9}[a-zぁ-ん ]*\\d{0,3}[a-z ω€]{8}[a-z ω€]*".encode()]
    alpha = np.frombuffer(b"abcdefghij klmnopqrstuvwxyz", dtype=np.uint8)
    for L in (256,):   # (128-byte rows keep the one-launch kernel: 64-byte halves gained 1.6 % for `.match.`)
        n, H = 64 * 29 + 5, L // 2
        for kind in ("ascii", "utf8", "broken"):
            rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
            for i in range(0, n, 2):   # digits somewhere: a run of 1..3 (match), 4 (no match), two runs (no match)
                off = [rng.randint(6, H - 8), rng.randint(H + 2, L - 16), H - 2, L - 3][(i // 2) % 4]
                k = [1, 3, 4, 2][(i // 2) % 4]
                rows[i, off:off + k] = np.frombuffer(b"0123"[:k], dtype=np.uint8)
                if (i // 2) % 7 == 0:
                    rows[i, 3:4] = ord("7")
            if kind != "ascii":
                for i in range(5, n, 9):
                    c = np.frombuffer("んω€".encode(), dtype=np.uint8)
                    off = rng.randint(0, L - len(c))
                    rows[i, off:off + len(c)] = c
            if kind == "broken":
                for i in range(7, n, 23):
                    rows[i, rng.randint(0, L - 1)] = rng.choice([0x80, 0xC0, 0xFF, 0xE3])
            for pat in pats:
                of, _, _ = oracle_lib.batch(1, pat, rows, NT)
                monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)
                prog, f, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
                assert prog.info()["flags"] & 256 and not prog.info()["flags"] & ((1 << 13) | 8), pat
                assert prog.last_path() in (5, 8), (pat, L, kind, prog.last_path())
                bad = np.nonzero(f != of)[0]
                assert bad.size == 0, (pat, L, kind, int(bad[0]), int(f[bad[0]]), int(of[bad[0]]), rows[bad[0]].tobytes())
                assert 0 < int(of.sum()) < n, (pat, L, kind)
                monkeypatch.setenv("FXAMD_HALF_SCH", "1")
                prog2, f2, _, _ = _device_run(fx, pat, fx.OP_MATCH, rows, spans=False)
                assert prog2.last_path() not in (5, 8) and np.array_equal(f2, of), (pat, L, kind, prog2.last_path())
    monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)


def test_long_rows_chain_tables_in_128_byte_segments(fx, monkeypatch):
    """Round 4: rows longer than 256 bytes of programs on the chain tables are walked in 128-byte segments (fx_search_fast NOHALF: the
    half-row kernel's loop at four waves per SIMD, finished from global memory) -- class-level first pass and byte-level pass over all
    tiles, lengths of every residue (a shorter last segment that ends at any byte, the tile's last bytes), matches across segment
    borders, UTF-8 and broken rows, spans and flags only -- against the oracle and the 256-byte segments (FXAMD_HALF_SCH=7)."""
    import random
    rng = random.Random(2500)
    nrng = np.random.default_rng(2500)
    alpha = np.frombuffer(b"abcdefghij klmnop0123456789-@._", dtype=np.uint8)
    pats = [rb"[a-z]{6}\d{1,3}[a-z ]{6}", rb"(19|20)\d\d-(0[1-9]|1[012])-(0[1-9]|[12][0-9]|3[01])", rb"[a-z]{3,5}\d{2,4}[a-z ]{3}$", "[ぁ-ん]{3}[ァ-ヶ]{3}[0-9]{3}".encode()]
    seeds = [b"abcdef123abc de", b"2024-02-29", b"abcd1234efg", "あいうアイウ123".encode(), b"1999-12-31", b"ghijkl7mnop  "]
    for pat in pats[:3]:   # (the fourth runs on the nibble tables: the 256-byte segments, for comparison)
        fl = fx.Program(pat, fx.OP_SEARCH).info()["flags"]
        assert (fl & 256) and not (fl & ((1 << 13) | 8)), (pat, fl)   # chain tables only
    for L in (257, 259, 300, 383, 384, 385, 400, 512, 515, 1000, 1024, 2049):
        n = 64 * 5 + 17
        rows = alpha[nrng.integers(0, len(alpha), size=(n, L))].copy()
        for i in range(0, n, 2):
            sd = np.frombuffer(seeds[(i // 2) % len(seeds)], dtype=np.uint8)
            border = 128 * rng.randint(1, L // 128)
            off = [0, L - len(sd), max(0, min(L - len(sd), border - rng.randint(0, len(sd)))), rng.randint(0, L - len(sd))][(i // 2) % 4]
            rows[i, off:off + len(sd)] = sd
        for i in range(3, n, 13):
            c = np.frombuffer("んω€".encode(), dtype=np.uint8)
            off = rng.randint(0, L - len(c))
            rows[i, off:off + len(c)] = c
        for i in range(9, n, 31):
            rows[i, rng.randint(0, L - 1)] = rng.choice([0x80, 0xC0, 0xFF, 0xE3])
        rows[1::64, L - 11:] = np.frombuffer(b"abcd1234efg", dtype=np.uint8)   # (a match of the `$` pattern at the row's last byte)
        for pat in pats:
            of, oa, ob = oracle_lib.batch(2, pat, rows, NT)
            for hook in (None, "7"):
                if hook:
                    monkeypatch.setenv("FXAMD_HALF_SCH", hook)
                else:
                    monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)
                prog, f, a, b = _device_run(fx, pat, fx.OP_SEARCH, rows)
                bad = np.nonzero((f != of) | (a != oa) | (b != ob))[0]
                assert bad.size == 0, (pat, L, hook, prog.last_path(), int(bad[0]), int(f[bad[0]]), int(a[bad[0]]), int(b[bad[0]]), int(of[bad[0]]), int(oa[bad[0]]), int(ob[bad[0]]))
                _, f2, _, _ = _device_run(fx, pat, fx.OP_SEARCH, rows, spans=False)
                assert np.array_equal(f2, of), (pat, L, hook, "flags only")
            assert int(of.sum()) > 0, (pat, L)
        # `.match.` (fx_match_fast NOHALF): letters and blanks with one run of 1..4 digits (up to 3: a match), some rows with UTF-8 / broken bytes
        mrows = np.frombuffer(b"abcdefghij klmnop", dtype=np.uint8)[nrng.integers(0, 17, size=(n, L))].copy()
        for i in range(n):
            k = 1 + i % 4
            off = min([7, L - 9, 128 * rng.randint(1, L // 128) - 2, rng.randint(6, L - 12)][i % 4], L - 8)
            mrows[i, off:off + k] = np.frombuffer(b"0123"[:k], dtype=np.uint8)
        for i in range(5, n, 17):
            c = np.frombuffer("んω".encode(), dtype=np.uint8)
            off = rng.randint(0, L - len(c))
            mrows[i, off:off + len(c)] = c
        for i in range(11, n, 37):
            mrows[i, rng.randint(0, L - 1)] = 0xFF
        for pat in (rb"[a-z ]{6}[a-z ]*\d{0,3}[a-z ]{6}[a-z ]*", "[a-zぁ-ん ]{9}[a-zぁ-ん ]*\\d{0,3}[a-z ω]{8}[a-z ω]*".encode()):
            om, _, _ = oracle_lib.batch(1, pat, mrows, NT)
            for hook in (None, "7"):
                if hook:
                    monkeypatch.setenv("FXAMD_HALF_SCH", hook)
                else:
                    monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)
                pm, fm, _, _ = _device_run(fx, pat, fx.OP_MATCH, mrows, spans=False)
                bad = np.nonzero(fm != om)[0]
                assert bad.size == 0, (pat, L, hook, "match", pm.last_path(), int(bad[0]), int(fm[bad[0]]), int(om[bad[0]]))
            assert 0 < int(om.sum()) < n, (pat, L)
    monkeypatch.delenv("FXAMD_HALF_SCH", raising=False)


def test_chain_table_rows_under_graph_capture_keep_the_one_launch_kernel(fx):
    """Round 4 moved 256-byte rows of chain-table programs to the half-row multi-pass pipeline, whose counter groups alternate on the host; a call on a stream that is being captured into a hipGraph keeps the one-launch kernel
    (`last_path` 9-14) and the graph replays on new row contents."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    pat = r"[a-z]{6}\d{1,3}[a-z ]{6}"
    for L, op, spans in ((256, fx.OP_SEARCH, True), (256, fx.OP_SEARCH, False), (256, fx.OP_MATCH, False)):
        n = 4096
        p = r"[a-z ]{6}[a-z ]*\d{0,3}[a-z ]{6}[a-z ]*" if op == fx.OP_MATCH else pat
        def batch(start):
            flat = synth.batch("cfg3", start, n, dev).reshape(-1)
            return flat[: (flat.numel() // L) * L].reshape(-1, L)[:n * 256 // L].contiguous()
        rows = batch(0)
        m = rows.shape[0]
        prog = fx.Program(p, op)
        out = prog.match_device(rows, spans=spans)
        torch.cuda.synchronize()
        assert prog.last_path() in (5, 7, 8), (L, op, prog.last_path())
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            assert fx.lib().fxamd_program_reserve(prog._h, m, side.cuda_stream) == 0
            prog.match_device(rows, spans=spans, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            prog.match_device(rows, spans=spans, out=out)
        assert prog.last_path() in (9, 10, 11, 12, 13, 14), (L, op, prog.last_path())
        for start in (10000, 77777):
            rows.copy_(batch(start))
            g.replay()
            torch.cuda.synchronize()
            got = [t.clone() if t is not None else None for t in out]
            want = prog.match_device(rows, spans=spans)
            torch.cuda.synchronize()
            assert torch.equal(got[0], want[0]), (L, op, start)
            if spans:
                assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]), (L, op, start)


def test_adaptive_first_pass_on_mostly_utf8_batches(fx, monkeypatch):
    """Round 4 (FX_ADAPT_CALLS): on 256-byte rows the half-row pipeline's first pass, having found most tiles holding bytes >= 0x80, marks
    every tile for the follow-up WITHOUT loading it for the next calls on the same scratch set (a persistent count-down on the device).
    Results must never depend on that word: a UTF-8 batch called many times in a row, then a pure-ASCII batch (the stale word sends its
    tiles through the follow-up), then a mixed one -- every call equal to the oracle; the same with FXAMD_NO_ADAPT=1."""
    import torch
    from forgex_amd import synth
    dev = torch.device("cuda")
    n = 64 * 400
    u = torch.full((n, 256), 32, dtype=torch.uint8, device=dev)
    u[:, :190] = synth.batch("cfg4", 500, n, dev)[:, :190]
    a = synth.batch("cfg3", 900, n, dev)
    mixed = a.clone()
    mixed[64 * 100:64 * 300] = u[64 * 100:64 * 300]
    for hook in (None, "1"):
        if hook:
            monkeypatch.setenv("FXAMD_NO_ADAPT", hook)
        else:
            monkeypatch.delenv("FXAMD_NO_ADAPT", raising=False)
        for pat in (r"[a-z ]+\d*", r"\d+[a-z]"):
            prog = fx.Program(pat, fx.OP_SEARCH)
            for spans in (True, False):
                for rows, calls in ((u, 12), (a, 11), (mixed, 4), (u, 3), (a, 2)):
                    k = 3000
                    of, oa, ob = oracle_lib.batch(2, pat.encode(), rows[:k].cpu().numpy(), NT)
                    first = None
                    for c in range(calls):
                        f, fa, fb = prog.match_device(rows, spans=spans)
                        torch.cuda.synchronize()
                        assert prog.last_path() == 16, (pat, prog.last_path())
                        got = (f.clone(), fa.clone() if spans else None, fb.clone() if spans else None)
                        if first is None:
                            first = got
                            assert np.array_equal(got[0][:k].cpu().numpy(), of), (pat, spans, hook, c)
                            if spans:
                                assert np.array_equal(got[1][:k].cpu().numpy(), oa) and np.array_equal(got[2][:k].cpu().numpy(), ob), (pat, hook, c)
                        else:
                            assert torch.equal(got[0], first[0]), (pat, spans, hook, c)
                            if spans:
                                assert torch.equal(got[1], first[1]) and torch.equal(got[2], first[2]), (pat, hook, c)
    monkeypatch.delenv("FXAMD_NO_ADAPT", raising=False)
