"""CPU tests (no GPU) of the product's host side: the table compiler + per-row procedure (compiled for the host by the
TEST-ONLY harness tests/support/host_walk.cpp) against the golden vectors and the oracle; the C ABI loads and exports
every symbol of include/forgex_amd.h; compile/status/blob behaviour through the ABI (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden
import fuzz_diff

HW = os.path.join(golden.ROOT, "tests", "support", "host_walk")


def test_tables_reproduce_golden_vectors(built):
    pairs = [(c, e) for c, e in golden.expected_lines_from_golden(golden.load_ref_tests()) if c[0] in "IMRV"]
    out = golden.run_protocol(HW, [c for c, _ in pairs])
    bad = [(c, e, a) for (c, e), a in zip(pairs, out) if not a.startswith("U") and not golden.line_matches(e, a)]
    unsupported = [c for (c, e), a in zip(pairs, out) if a.startswith("U")]
    assert not bad, bad[:5]
    assert len(unsupported) == 0     # DFA state explosion (e.g. `.*a(a|b){500}c{20}`) falls back to NFA state-set simulation


def test_tables_equal_oracle_on_fuzz(built):
    total = 0
    for seed in (7, 8):
        cases = [c for c in fuzz_diff.gen_cases(seed, 4000) if c[0] in "IMRV"]
        a = golden.run_protocol(HW, cases)
        b = golden.run_protocol(golden.ORACLE_CLI, cases)
        diffs = [(c, x, y) for c, x, y in zip(cases, a, b) if not x.startswith("U") and x != y]
        assert not diffs, diffs[:5]
        total += len(cases)
    assert total > 5000


def test_byte_level_tables_equal_oracle_on_utf8_fuzz(built, monkeypatch):
    """FXP_F_BYTE_DFA: the automata composed with the UTF-8 decoder, walked over raw bytes by the host harness exactly as the
    tile kernels' BYTES modes do (structurally invalid rows fall through to the decode path), against the oracle on texts full
    of multi-byte, overlong, out-of-range and broken sequences; also the golden vectors once more through these tables."""
    import random
    import fuzz_bytes
    monkeypatch.setenv("FX_HW_BYTES", "1")
    rng = random.Random(11)
    cases = []
    for _ in range(4000):
        pat = rng.choice(fuzz_bytes.EXTRA_PATTERNS) if rng.random() < 0.4 else fuzz_diff.gen_pattern(rng)
        cases.append((rng.choice(["I", "M", "R", "R"]), pat.encode(), fuzz_bytes.gen_text(rng)))
    a = golden.run_protocol(HW, cases)
    b = golden.run_protocol(golden.ORACLE_CLI, cases)
    diffs = [(c, x, y) for c, x, y in zip(cases, a, b) if not x.startswith("U") and x != y]
    assert not diffs, diffs[:5]
    pairs = [(c, e) for c, e in golden.expected_lines_from_golden(golden.load_ref_tests()) if c[0] in "IMR"]
    out = golden.run_protocol(HW, [c for c, _ in pairs])
    bad = [(c, e, x) for (c, e), x in zip(pairs, out) if not x.startswith("U") and not golden.line_matches(e, x)]
    assert not bad, bad[:5]
    # the same automata in the 16-state NIBBLE format (what the tile kernels' wide scheme walks), where a program carries it
    monkeypatch.setenv("FX_HW_BYTES", "w16")
    a16 = golden.run_protocol(HW, cases[:2500])
    diffs = [(c, x, y) for c, x, y in zip(cases[:2500], a16, b[:2500]) if not x.startswith("U") and x != y]
    assert not diffs, diffs[:5]
    out = golden.run_protocol(HW, [c for c, _ in pairs])
    bad = [(c, e, x) for (c, e), x in zip(pairs, out) if not x.startswith("U") and not golden.line_matches(e, x)]
    assert not bad, bad[:5]
    monkeypatch.setenv("FX_HW_BYTES", "1")
    # the tables exist for the BASELINE patterns (with a prefilter literal only where it is proven equal to brute force) and are small
    lib = ctypes.CDLL(os.path.join(golden.ROOT, "tests", "support", "libhostwalk.so"))
    lib.hw_byte_info.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    for pat, want in (("[a-z]+\\d+", 1), ("[α-ωぁ-ん]+", 1), ("foo(bar|baz)", 1), ("aa[bc]", 0)):   # (a prefix with a border: no proof, no tables)
        info = (ctypes.c_int32 * 8)()
        lib.hw_byte_info(pat.encode(), len(pat.encode()), 0, info)
        assert info[0] == want and info[4] < 8192, (pat, list(info))


def test_prefilter_equivalence_proof_holds_on_fuzz(built, monkeypatch):
    """Programs that carry tile-kernel tables are searched by brute force on pure-ASCII rows although the reference would use
    its candidate-list driver (prefix / suffix literals): the compile-time proof of that equivalence (compile.cpp, `brute_equiv`)
    against the oracle, with the host harness forced onto the brute-force path (FX_HW_FAST=1)."""
    import random
    import fuzz_prefilter
    monkeypatch.setenv("FX_HW_FAST", "1")
    rng = random.Random(71)
    cases = [fuzz_prefilter.gen_case(rng) for _ in range(6000)]
    a = golden.run_protocol(HW, cases)
    b = golden.run_protocol(golden.ORACLE_CLI, cases)
    diffs = [(c, x, y) for c, x, y in zip(cases, a, b) if not x.startswith("U") and x != y]
    assert not diffs, diffs[:5]
    # round 4: literals with non-ASCII characters (texts with overlong encodings and structure errors; brute force and byte-level
    # tables) and prefix / suffix literals that overlap in the shortest match (`ab{2,}`)
    for mode, extra in (("FX_FUZZ_UTF8", None), ("FX_FUZZ_UTF8", "FX_HW_BYTES"), ("FX_FUZZ_OVERLAP", None)):
        monkeypatch.setenv(mode, "1")
        if extra:
            monkeypatch.setenv(extra, "1")
        rng = random.Random(72)
        cases = [fuzz_prefilter.gen_case(rng) for _ in range(3000)]
        a = golden.run_protocol(HW, cases)
        b = golden.run_protocol(golden.ORACLE_CLI, cases)
        diffs = [(c, x, y) for c, x, y in zip(cases, a, b) if not x.startswith("U") and x != y]
        assert not diffs, (mode, extra, diffs[:5])
        monkeypatch.delenv(mode)
        if extra:
            monkeypatch.delenv(extra)
    # `literal.*literal` shapes are among the programs the proof admits
    lib = ctypes.CDLL(os.path.join(golden.ROOT, "tests", "support", "libhostwalk.so"))
    lib.hw_info.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    # ... and so are prefixes with a border (overlap detector in R), with or without a proven suffix; a suffix literal the proof
    # does not cover keeps the pattern on the general kernel
    # (round 4: non-ASCII literals and literals that overlap in the shortest match are admitted; a match that may BE its suffix is not)
    for pat, want in ((b"abc.*xyz", True), (b"id=\\d+;", True), (b"aa[bc]", True), (b"aa.*bb", True), (b"foo.a b", False),
                      ("夢.{1,7}胡蝶".encode(), True), ("α.*β".encode(), True), (b"ab{2,}", True), (b"A{1,2}bb", False)):
        # (round 6: what the proofs do not cover is admitted with a per-ROW check instead -- FXP_F_PREFIX_CHECK / FXP_F_SUFFIX_CHECK, bits 21 / 22: rows
        #  that fail it take the general procedure inside the launch -- `want`: admitted by proof alone)
        info = (ctypes.c_int32 * 8)()
        lib.hw_info(pat, len(pat), 0, info)
        assert bool(info[1] & (8 | 256 | 0x2000)), (pat, hex(info[1]))
        assert bool(info[1] & ((1 << 21) | (1 << 22))) == (not want), (pat, hex(info[1]))


def test_config_rows_tables_vs_oracle(built):
    """Small slices of the five BASELINE configs through the host walker (one compile per batch) and the oracle."""
    import torch
    import oracle_lib
    from forgex_amd import synth
    lib = ctypes.CDLL(os.path.join(golden.ROOT, "tests", "support", "libhostwalk.so"))
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.hw_batch.argtypes = [ctypes.c_char_p, i64, ctypes.c_int, vp, i64, i64, vp, vp, vp]
    for cfg, n, op in (("cfg1", 1000, 1), ("cfg2", 3000, 0), ("cfg3", 400, 0), ("cfg4", 600, 0), ("cfg5", 800, 0)):
        pat = synth.PATTERNS[cfg].encode()
        rows = synth.batch(cfg, 0, n, torch.device("cpu")).numpy()
        f = np.zeros(n, np.uint8)
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        st = lib.hw_batch(pat, len(pat), op, rows.ctypes.data_as(vp), n, rows.shape[1], f.ctypes.data_as(vp), a.ctypes.data_as(vp), b.ctypes.data_as(vp))
        assert st == 0
        of, oa, ob = oracle_lib.batch(1 if op == 1 else 2, pat, rows, os.cpu_count() or 1)
        assert np.array_equal(f, of), cfg
        if op == 0:
            assert np.array_equal(a, oa) and np.array_equal(b, ob), cfg
        assert 0 < int(f.sum()) < n


def test_c_abi_exports_every_declared_symbol(built):
    """Both headers: the drop-in boundary (forgex_amd.h) and the measurement hooks (forgex_amd_bench.h) -- every declared symbol is
    exported, and nothing measurement-only is declared in the boundary header."""
    import forgex_amd
    from forgex_amd import _lib
    header = open(os.path.join(golden.ROOT, "include", "forgex_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(fxamd_[a-z_]+)\s*\(", header)))
    assert declared == sorted(_lib.EXPORTED_SYMBOLS)
    bench = open(os.path.join(golden.ROOT, "include", "forgex_amd_bench.h")).read()
    hooks = sorted(set(re.findall(r"\b(fxamd_[a-z_]+)\s*\(", bench)))
    assert hooks == sorted(_lib.BENCH_SYMBOLS) and not set(hooks) & set(declared)
    L = forgex_amd.lib()
    for name in declared + hooks:
        assert hasattr(L, name), name


def _hostwalk():
    lib = ctypes.CDLL(os.path.join(golden.ROOT, "tests", "support", "libhostwalk.so"))
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.hw_validate.argtypes = [ctypes.c_char_p, i64, ctypes.c_int]
    lib.hw_validate_blob.argtypes = [vp, i64]
    lib.hw_blob_checksum.argtypes = [vp, i64]
    lib.hw_blob_checksum.restype = ctypes.c_uint32
    lib.hw_dump_nfa.argtypes = [ctypes.c_char_p, i64, ctypes.c_int, vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp]
    return lib


def _golden_patterns():
    pats = []
    for _, kind, f in golden.load_ref_tests():
        if kind in ("in", "match", "regex"):
            pats.append((golden.unhx(f[0]), 1 if kind == "match" else 0))
    return sorted(set(pats))


def test_every_compiled_program_passes_the_wire_format_check(built):
    """fxc::validate_blob (what fxamd_program_from_blob applies to images from outside) accepts every program this compiler emits
    for the patterns of the reference's own tests, both operators, and for the BASELINE patterns."""
    from forgex_amd import synth
    lib = _hostwalk()
    pats = _golden_patterns() + [(p.encode(), 0) for p in synth.PATTERNS.values()] + [(p.encode(), 1) for p in synth.PATTERNS.values()]
    pats += [(rb"[ab]*a[ab]{20}", 0), (rb"aa[bc]", 0), (rb"abc", 0), (rb"a(", 0), (rb"--[a-z]+", 0), (rb"[\w.]+@[\w.]+\.[a-z]{2,4}", 0)]
    assert len(pats) > 300
    for pat, op in pats:
        assert lib.hw_validate(pat, len(pat), op) == 0, (pat, op)


def test_corrupt_blobs_are_rejected(built):
    """Wire format: a patched offset, count, table entry or flag -- with the checksum repaired, so that the STRUCTURAL check is what
    has to catch it -- is refused with FXAMD_E_BLOB, as are truncation and a stale checksum."""
    import struct
    import forgex_amd as fx
    from forgex_amd import synth
    lib = _hostwalk()
    L = fx.lib()

    def load(blob):
        h = ctypes.c_void_p()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        rc = L.fxamd_program_from_blob(buf, len(blob), ctypes.byref(h))
        if rc == 0:
            L.fxamd_program_free(h)
        return rc

    def fixed(blob):
        b = bytearray(blob)
        off = FIELDS["checksum"]
        b[off:off + 4] = b"\0\0\0\0"
        arr = (ctypes.c_char * len(b)).from_buffer(b)
        struct.pack_into("<I", b, off, lib.hw_blob_checksum(arr, len(b)))
        return bytes(b)

    # field offsets of FxpHeader (all uint32), read from program.h so that the test follows the wire format
    src = open(os.path.join(golden.ROOT, "forgex_amd", "csrc", "program.h")).read()
    body = src[src.index("struct FxpHeader {"):src.index("};", src.index("struct FxpHeader {"))]
    body = re.sub(r"//[^\n]*", "", body)
    names = []
    for decl in re.findall(r"uint32_t\s+([^;]+);", body):
        for item in decl.split(","):
            m = re.match(r"\s*(\w+)(?:\[(\d+)\])?\s*$", item)
            names += [m.group(1)] * int(m.group(2) or 1) if m.group(2) else [m.group(1)]
    FIELDS = {}
    for i, nm in enumerate(names):
        FIELDS.setdefault(nm, 4 * i)
    n_bad = 0
    for cfg in ("cfg2", "cfg3", "cfg4", "cfg1"):
        good = fx.Program(synth.PATTERNS[cfg], fx.OP_SEARCH).blob()
        assert struct.unpack_from("<I", good, FIELDS["total_bytes"])[0] == len(good)
        assert load(good) == 0 and load(fixed(good)) == 0
        assert load(good[:-16]) == -4                                  # truncated
        stale = bytearray(good)
        stale[len(stale) // 2] ^= 0x40
        assert load(bytes(stale)) == -4                                # checksum
        for field, value in (("off_cls_page", 0x7FFFFFF0), ("n_classes", 60000), ("off_fastA", len(good) - 8), ("nA", 0x7000), ("nR", 0x7000),
                             ("off_TA", len(good) - 2), ("n_pages", 1000), ("off_byte_TR", len(good) - 4), ("byte_TR_bytes", 60000),
                             ("off_bw16R", len(good) - 64), ("off_w16A", 0xFFFFFFF0), ("chain_TA_bytes", 65000), ("A_init", 0x6000),
                             ("byte_R_start", 0xFFF0), ("mode", 9), ("flags", 0x80000000), ("cls_ffff", 0x4000), ("n_bounds", 1 << 21),
                             ("byte_n_classes", 300), ("byte_row_bytes", 6)):
            b = bytearray(good)
            old = struct.unpack_from("<I", b, FIELDS[field])[0]
            if field == "flags":
                value |= old
            if old == value:
                continue
            struct.pack_into("<I", b, FIELDS[field], value)
            rc = load(fixed(bytes(b)))
            flags = struct.unpack_from("<I", good, FIELDS["flags"])[0]
            table_present = {"off_fastA": 8, "off_byte_TR": 0x1000, "byte_TR_bytes": 0x1000, "off_bw16R": 0x8000, "off_w16A": 0x2000,
                             "chain_TA_bytes": 0x100, "byte_R_start": 0x1000, "byte_n_classes": 0x1000, "byte_row_bytes": 0x1000}.get(field)
            if table_present is not None and not (flags & table_present):
                continue   # the table is absent from this program: its fields are not read
            assert rc == -4, (cfg, field, rc)
            n_bad += 1
        # a table ENTRY that points outside its table (byte-level chain table of R)
        flags = struct.unpack_from("<I", good, FIELDS["flags"])[0]
        if flags & 0x1000:
            b = bytearray(good)
            off = struct.unpack_from("<I", b, FIELDS["off_byte_TR"])[0]
            struct.pack_into("<H", b, off + 2, 0xFFF0)
            assert load(fixed(bytes(b))) == -4
            n_bad += 1
    assert n_bad > 40
    # the LATCHED format of the reverse automaton (FXP_F_R_LATCH, program.h; round 6): configs 3 and 4 carry it (R of 3 / 2 states), config 2 (7) does not;
    # it is the plain table doubled -- a step from a state below 4 goes where the plain table goes (+ 4 when that is a hit state), a step from 4 + s where the
    # plain table sends s, + 4 -- and an image whose latched state unlatches, or whose flag claims a table that is not there, is refused
    for cfg, want in (("cfg3", True), ("cfg4", True), ("cfg2", False)):
        good = fx.Program(synth.PATTERNS[cfg], fx.OP_SEARCH).blob()
        flags = struct.unpack_from("<I", good, FIELDS["flags"])[0]
        assert bool(flags & (1 << 23)) == want, (cfg, hex(flags))
        if not want:
            continue
        orl, orp = struct.unpack_from("<I", good, FIELDS["off_fastRL"])[0], struct.unpack_from("<I", good, FIELDS["off_fastR"])[0]
        hitmin, nR = struct.unpack_from("<I", good, FIELDS["fast_hitR_min"])[0], struct.unpack_from("<I", good, FIELDS["nR"])[0]
        assert nR <= 4
        for sym in range(256):
            for st in range(4):
                d = good[orp + 8 * sym + st]
                assert good[orl + 8 * sym + st] == (d + 4 if hitmin <= d < nR else d), (cfg, sym, st)
                assert good[orl + 8 * sym + 4 + st] == d + 4, (cfg, sym, st)
        b = bytearray(good)
        b[orl + 8 * ord("a") + 5] = 1                                  # a latched state that unlatches
        assert load(fixed(bytes(b))) == -4
        b = bytearray(good)
        b[orl + 8 * ord("a") + 1] = 9                                  # not a state
        assert load(fixed(bytes(b))) == -4
        b = bytearray(good)
        struct.pack_into("<I", b, FIELDS["off_fastRL"], len(good) - 100)   # the table does not fit
        assert load(fixed(bytes(b))) == -4
    # NFA-simulation program: nfa_words = 0 must not reach the launch code (it divides by the per-row scratch size)
    good = fx.Program(r"[ab]*a[ab]{20}", fx.OP_SEARCH).blob()
    assert load(good) == 0
    b = bytearray(good)
    struct.pack_into("<I", b, FIELDS["nfa_words"], 0)
    assert load(fixed(bytes(b))) == -4


def test_compile_nfa_equals_compile_on_the_golden_patterns(built):
    """INTEGRATION.md route B: the range-NFA and the literals the front end builds, flattened the way a Fortran host would flatten
    the reference's nfa_graph_t, handed to fxamd_compile_nfa -- the program must be byte-identical to fxamd_compile's."""
    import forgex_amd as fx
    from forgex_amd import synth
    lib = _hostwalk()
    L = fx.lib()
    vp = ctypes.c_void_p
    n_checked = 0
    pats = _golden_patterns() + [(p.encode(), 0) for p in synth.PATTERNS.values()] + [(rb"aa[bc]", 0), (rb"abc.*xyz", 0), (rb"[ab]*a[ab]{20}", 0)]
    for pat, op in pats:
        counts = (ctypes.c_int64 * 6)()
        lit_len = (ctypes.c_int64 * 3)()
        rc = lib.hw_dump_nfa(pat, len(pat), op, counts, 0, 0, None, None, None, None, None, None, 0, lit_len)
        if rc != 0:
            continue   # invalid pattern: no NFA to hand over
        nt, ns = counts[3], counts[4]
        src, dst = np.zeros(max(nt, 1), np.int32), np.zeros(max(nt, 1), np.int32)
        sb = np.zeros(nt + 1, np.int64)
        smin, smax = np.zeros(max(ns, 1), np.int32), np.zeros(max(ns, 1), np.int32)
        lits = ctypes.create_string_buffer(int(sum(lit_len)) + 1)
        rc = lib.hw_dump_nfa(pat, len(pat), op, counts, nt, ns, src.ctypes.data_as(vp), dst.ctypes.data_as(vp), sb.ctypes.data_as(vp),
                             smin.ctypes.data_as(vp), smax.ctypes.data_as(vp), lits, len(lits), lit_len)
        assert rc == 0 and counts[3] == nt and counts[4] == ns
        raw = lits.raw
        la, lp, ls = int(lit_len[0]), int(lit_len[1]), int(lit_len[2])
        h = ctypes.c_void_p()
        st = ctypes.c_int32(-1)
        rc = L.fxamd_compile_nfa(int(counts[0]), int(counts[1]), int(counts[2]), nt, src.ctypes.data_as(vp), dst.ctypes.data_as(vp),
                                 sb.ctypes.data_as(vp), smin.ctypes.data_as(vp), smax.ctypes.data_as(vp), raw[:la], la, raw[la:la + lp], lp,
                                 raw[la + lp:la + lp + ls], ls, op, ctypes.byref(h), ctypes.byref(st))
        assert rc == 0, (pat, rc)
        n = L.fxamd_program_blob_size(h)
        buf = (ctypes.c_char * n)()
        assert L.fxamd_program_blob(h, buf, n) == 0
        L.fxamd_program_free(h)
        want = fx.Program(pat, op)
        assert st.value == want.status and bytes(buf) == want.blob(), (pat, op)
        n_checked += 1
    assert n_checked > 300
    # argument checks: nothing crosses the boundary as an exception or an out-of-bounds read
    one = np.array([1], np.int32)
    two = np.array([2], np.int32)
    h = ctypes.c_void_p()
    st = ctypes.c_int32(0)
    bad_begin = np.array([-1, 1], np.int64)
    lo, hi = np.array([98], np.int32), np.array([97], np.int32)
    ok_begin = np.array([0, 1], np.int64)
    for sb_, mn, mx in ((bad_begin, lo, lo), (ok_begin, lo, hi)):
        rc = L.fxamd_compile_nfa(2, 1, 2, 1, one.ctypes.data_as(vp), two.ctypes.data_as(vp), sb_.ctypes.data_as(vp), mn.ctypes.data_as(vp),
                                 mx.ctypes.data_as(vp), b"", 0, b"", 0, b"", 0, 0, ctypes.byref(h), ctypes.byref(st))
        assert rc == -1


def test_compile_status_and_blob_roundtrip(built):
    import forgex_amd as fx
    p = fx.Program(r"[a-z]+\d+", fx.OP_SEARCH)
    info = p.info()
    assert p.status == 0 and info["mode"] == 1 and info["nA"] <= 8 and info["nR"] <= 8 and (info["flags"] & 8)
    q = fx.Program.from_blob(p.blob(), fx.OP_SEARCH)
    assert q.info() == info and q.blob() == p.blob()
    with pytest.raises(ValueError):
        fx.Program.from_blob(p.blob()[:-16], fx.OP_SEARCH)
    bad = fx.Program("a(", fx.OP_SEARCH)
    assert bad.status == 2 and not bad.valid and fx.strerror(2) == "ERROR: Closing parenthesis is expected."
    assert fx.is_valid_regex("a{2,1}") is False and fx.is_valid_regex(r"\d{3}-\d{4}") is True
    big = fx.Program(r"[ab]*a[ab]{20}", fx.OP_SEARCH)      # 2^21 DFA states: falls back to on-device NFA state-set simulation
    assert big.valid and big.supported and big.status == 0 and (big.info()["flags"] & 128)
    assert fx.Program("foo(bar|baz)", fx.OP_SEARCH).info()["flags"] & 2   # prefilter literal `fooba`
    assert fx.Program("abc", fx.OP_SEARCH).info()["mode"] == 2            # whole-pattern literal -> INDEX path


def test_program_images_equal_the_committed_fixture(built, monkeypatch):
    """The program image is the wire format between ranks: the compiler's output for a fixed pattern list (bench configs, the GPU
    tests' patterns, shapes of the reference's tests) must stay byte-identical to tests/golden/program_images.tsv -- status, size,
    SHA-256.  (A deliberate table change bumps FXP_VERSION and regenerates the file: tests/golden/make_program_images.py.)"""
    import hashlib
    import forgex_amd as fx
    monkeypatch.setenv("FXAMD_NO_CACHE", "1")
    n = 0
    with open(os.path.join(golden.GOLDEN, "program_images.tsv")) as f:
        for ln in f:
            if ln.startswith("#") or not ln.strip():
                continue
            hx, op, status, size, digest = ln.rstrip("\n").split("\t")
            pat = b"" if hx == "-" else bytes.fromhex(hx)
            q = fx.Program(pat, int(op))
            assert q.status == int(status), (pat, op)
            if q.status == 0:
                image = q.blob()
                assert len(image) == int(size) and hashlib.sha256(image).hexdigest()[:16] == digest, (pat, op)
            n += 1
    assert n >= 80


def test_error_codes_and_messages_match_reference(built):
    import forgex_amd as fx
    n = 0
    for prog, kind, f in golden.load_ref_tests():
        if kind != "error":
            continue
        p = fx.Program(golden.unhx(f[0]), fx.OP_SEARCH)
        assert p.status == int(f[2]), (f, p.status)
        assert fx.strerror(p.status).encode() == golden.unhx(f[3]), f
        n += 1
    assert n == 125


def test_match_path_fails_loudly_without_gpu(built):
    import forgex_amd as fx
    if fx.lib().fxamd_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        fx.in_("abc", [b"xabc"])


def test_host_compiler_under_address_and_ub_sanitizers(built, tmp_path):
    """The host side of the product (front end, table compiler, byte-level automata builders) and the row procedure, compiled with
    -fsanitize=address,undefined (CPU build only; the GPU pool has no sanitizer support) and driven with fuzzed patterns and texts."""
    import random
    import subprocess
    import fuzz_bytes
    import fuzz_prefilter
    exe = str(tmp_path / "host_walk_asan")
    src = [os.path.join(golden.ROOT, "tests", "support", "host_walk.cpp"), os.path.join(golden.ROOT, "forgex_amd", "csrc", "frontend.cpp"),
           os.path.join(golden.ROOT, "forgex_amd", "csrc", "compile.cpp")]
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-DHW_MAIN"] + src + ["-o", exe],
                        capture_output=True)
    if cc.returncode != 0:
        pytest.skip("sanitizer build not available: " + cc.stderr.decode()[:200])
    rng = random.Random(3)
    cases = [fuzz_prefilter.gen_case(rng) for _ in range(250)]
    for _ in range(250):
        pat = rng.choice(fuzz_bytes.EXTRA_PATTERNS) if rng.random() < 0.4 else fuzz_diff.gen_pattern(rng)
        cases.append((rng.choice(["I", "M", "R"]), pat.encode(), fuzz_bytes.gen_text(rng)))
    cases += fuzz_diff.gen_cases(5, 300)
    inp = "".join("%s %s %s\n" % (c[0], golden.hx(c[1]), golden.hx(c[2])) for c in cases)
    env = dict(os.environ, FX_HW_BYTES="1", FX_HW_FAST="1", ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe], input=inp.encode(), capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-2000:]
    assert len(r.stdout.decode().splitlines()) == len(cases)


# kernels of the default dispatch paths: the BASELINE configs' dominant kernels, their packed / follow-up variants, and the ragged
# instantiations that carried scratch after round 5.  (header, template declaration of the kernel, template arguments)
_ONE_SIG = "(const uint8_t*, int64_t, const uint8_t*, FastParams, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t, uint32_t, uint32_t, const uint32_t*, const uint8_t*)"
_SPAN_SIG = "(const uint8_t*, const int64_t, const uint8_t*, const FastParams, uint8_t*, int32_t*, int32_t*, uint32_t*, uint32_t*, uint8_t*, const uint32_t)"
_FAST_SIG = "(const uint8_t*, int64_t, const uint8_t*, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*)"
_DEFAULT_PATH_KERNELS = [
    ("fx_tile.hpp", "fx_search_fast", "8, true, 0, 0, false, true", _FAST_SIG),              # config 3: half-row first pass (plain format of R: FXAMD_NO_LATCH, R of 5..8 states)
    ("fx_tile.hpp", "fx_search_fast", "8, true, 0, 0, false, true, false, true", _FAST_SIG), # ... on the LATCHED format of R (round 6: what config 3 runs)
    ("fx_one.hpp", "fx_search_one", "16, true, 0, 3, false, false, true", _ONE_SIG),         # ... its gated follow-up
    ("fx_one.hpp", "fx_search_one", "4, true, 0, 3, false, true", _ONE_SIG),                 # config 2
    ("fx_one.hpp", "fx_search_one", "12, true, 0, 3, false, false", _ONE_SIG),               # config 4
    ("fx_span.hpp", "fx_search_span", "128, 0, false, false", _SPAN_SIG),                    # config 5's shard (plain format of R)
    ("fx_span.hpp", "fx_search_span", "128, 0, false, false, true", _SPAN_SIG),              # ... on the latched format (what config 5 runs)
    ("fx_span.hpp", "fx_search_span", "128, 0, true, false", _SPAN_SIG),                     # ... packed (what an 8-GPU run gathers)
    ("fx_span.hpp", "fx_search_span", "128, 0, true, false, true", _SPAN_SIG),
    ("fx_span.hpp", "fx_search_span", "16, 0, false, false, true", _SPAN_SIG),               # K = 8 rows per lane, latched
    ("fx_span.hpp", "fx_search_span", "32, 0, false, true, true", _SPAN_SIG),                # ragged rows, latched
    ("fx_span.hpp", "fx_search_span", "16, 0, false, false", _SPAN_SIG),                     # K = 8 rows per lane (scratch in round 5)
    ("fx_span.hpp", "fx_search_span", "16, 0, true, false", _SPAN_SIG),
    ("fx_span.hpp", "fx_search_span", "16, 2, false, false", _SPAN_SIG),
    ("fx_span.hpp", "fx_search_span", "32, 0, true, false", _SPAN_SIG),
    ("fx_span.hpp", "fx_search_span", "32, 0, false, true", _SPAN_SIG),                      # ragged rows: character(20)
    ("fx_span.hpp", "fx_search_span", "128, 0, false, true", _SPAN_SIG),                     # character(80), (100)
    ("fx_one.hpp", "fx_search_one", "8, true, 0, 3, true, false", _ONE_SIG),                 # ragged 65..127 bytes, UTF-8 programs (utf8_100)
    ("fx_one.hpp", "fx_search_one", "8, false, 0, 1, true, false", _ONE_SIG),                # ... flags only
    ("fx_one.hpp", "fx_search_one", "8, true, 1, 2, true, false", _ONE_SIG),                 # ... chain / nibble programs (83 spilled VGPRs in round 5)
    ("fx_one.hpp", "fx_search_one", "8, false, 0, 0, true, false, false, true", _ONE_SIG),   # `.match.` on ragged rows
    ("fx_one.hpp", "fx_search_one", "16, true, 0, 3, true, false", _ONE_SIG),                # ragged 129..255 bytes
]


def _kernel_resource_usage(header, kernel, args, sig, extra=()):
    """VGPRs / spills / scratch of ONE instantiation (hipcc cross-compiles gfx950 device code without a GPU; ~5 s)."""
    import subprocess
    import tempfile
    csrc = os.path.join(golden.ROOT, "forgex_amd", "csrc")
    src = '#include "%s"\ntemplate __global__ void %s<%s>%s;\nconst FxEnv& fx_env() { static FxEnv e{}; return e; }\n' % (header, kernel, args, sig)
    with tempfile.NamedTemporaryFile("w", suffix=".hip", delete=False) as f:
        f.write(src)
        path = f.name
    try:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + csrc, "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull] + list(extra), capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for key, short in (("VGPRs", "vgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("VGPRs Spill", "vspill"), ("SGPRs Spill", "sspill"), ("Occupancy [waves/SIMD]", "occ")):
        m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", r.stderr)
        assert m, (kernel, args, key, r.stderr[-500:])
        out[short] = int(m.group(1))
    return out


def test_default_path_kernels_use_no_scratch_memory():
    """round 5's last-tile patch in the ragged loaders cost 10..40 live VGPRs on the hot path and 25 of 58 one-launch kernels of the 128-byte
    instantiations spilled to scratch memory -- unnoticed, because nothing looked.  The kernels of the default dispatch paths are compiled here (device code only)
    with the resource-usage remarks: none may carry scratch, and the occupancy each is designed for must hold.  `make -C forgex_amd/csrc resource-usage-all` +
    tools/summarize_ru.py do the same for all ~1500 instantiations (profiles/r06_resource_usage.txt); when that output is present it is checked as well."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not present")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda k: _kernel_resource_usage(*k), _DEFAULT_PATH_KERNELS))
    bad = [(k[1], k[2], r) for k, r in zip(_DEFAULT_PATH_KERNELS, res) if r["scratch"] != 0 or r["vspill"] != 0]
    assert not bad, bad
    occ = {(k[1], k[2]): r["occ"] for k, r in zip(_DEFAULT_PATH_KERNELS, res)}
    assert occ[("fx_search_fast", "8, true, 0, 0, false, true")] >= 4 and occ[("fx_search_span", "128, 0, false, false")] >= 4, occ
    assert occ[("fx_search_fast", "8, true, 0, 0, false, true, false, true")] >= 4 and occ[("fx_search_span", "128, 0, false, false, true")] >= 4, occ
    assert occ[("fx_search_span", "16, 0, false, false")] >= 4 and occ[("fx_search_span", "32, 0, false, true")] >= 4, occ
    assert occ[("fx_search_one", "8, true, 0, 3, true, false")] >= 3, occ
    ru_dir = os.path.join(golden.ROOT, "forgex_amd", "csrc", "build_ru")
    csrc = os.path.join(golden.ROOT, "forgex_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hpp", ".hip", ".h", ".cpp"))]
    txts = [os.path.join(ru_dir, f) for f in os.listdir(ru_dir) if f.endswith(".txt")] if os.path.isdir(ru_dir) else []
    if len(txts) >= 20 and min(os.path.getmtime(t) for t in txts) >= max(os.path.getmtime(f) for f in srcs):   # (the remarks of THESE sources, every object)
        sys.path.insert(0, os.path.join(golden.ROOT, "tools"))
        import summarize_ru
        rows = [r for r in summarize_ru.load(ru_dir) if "scratch" in r]
        assert not [(r["obj"], r["kernel"], r["scratch"]) for r in rows if r["scratch"] != 0 and "fx_search_fast<3," not in r["kernel"]]


def test_c_abi_host_side_under_thread_sanitizer(built, tmp_path):
    """SURVEY section 5 asks for a ThreadSanitizer run of the C ABI's host side (VERDICT r05 missing 7): fxamd.hip's HOST code (--cuda-host-only: the
    compile cache and its mutex, refcounted handles, program images), the front end and the table compiler are compiled with -fsanitize=thread and
    linked with the product's own launcher objects; tests/support/tsan_abi.cpp then compiles / inspects / serialises / frees fourteen patterns from eight
    threads at once, with cache trims in between.  CPU build only (the GPU pool has no sanitizer support); nothing is enqueued on a device."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = "/opt/rocm/bin/hipcc"
    csrc = os.path.join(golden.ROOT, "forgex_amd", "csrc")
    objs_dir = os.path.join(csrc, "build")
    tiles = sorted(os.path.join(objs_dir, f) for f in os.listdir(objs_dir) if f.startswith("tile_") and f.endswith(".o")) if os.path.isdir(objs_dir) else []
    if not os.path.exists(hipcc) or len(tiles) < 18 or not os.path.exists(os.path.join(objs_dir, "span.o")):
        pytest.skip("needs hipcc and the product's objects (forgex_amd/csrc/build)")
    common = ["-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=thread", "-fno-omit-frame-pointer"]
    jobs = [[hipcc, "--offload-arch=gfx950", "--cuda-host-only"] + common + ["-c", os.path.join(csrc, "fxamd.hip"), "-o", str(tmp_path / "fxamd.o")],
            [hipcc, "-x", "c++"] + common + ["-c", os.path.join(csrc, "frontend.cpp"), "-o", str(tmp_path / "frontend.o")],
            [hipcc, "-x", "c++"] + common + ["-c", os.path.join(csrc, "compile.cpp"), "-o", str(tmp_path / "compile.o")]]
    with ThreadPoolExecutor(max_workers=3) as ex:
        done = list(ex.map(lambda c: subprocess.run(c, capture_output=True), jobs))
    for d in done:
        if d.returncode != 0:
            pytest.skip("thread sanitizer build not available: " + d.stderr.decode()[-300:])
    # (a host-only object still refers to the device image it would embed, `__hip_fatbin_<hash>`: an empty image stands in -- the HIP runtime loads code
    #  objects lazily, and this run never launches a kernel)
    nm = subprocess.run(["nm", str(tmp_path / "fxamd.o")], capture_output=True, text=True).stdout
    fat = sorted({w for line in nm.splitlines() for w in line.split() if w.startswith("__hip_fatbin_")})
    stub = tmp_path / "fatbin_stub.c"
    stub.write_text("".join('__attribute__((visibility("default"), aligned(4096))) const char %s[4096] = {0};\n' % f for f in fat))
    assert subprocess.run(["gcc", "-fPIC", "-c", str(stub), "-o", str(tmp_path / "fatbin_stub.o")], capture_output=True).returncode == 0
    lib = str(tmp_path / "libforgex_amd_tsan.so")
    ld = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=thread"] + tiles +
                        [os.path.join(objs_dir, "span.o"), str(tmp_path / "fxamd.o"), str(tmp_path / "frontend.o"), str(tmp_path / "compile.o"), str(tmp_path / "fatbin_stub.o"),
                         "-o", lib], capture_output=True)
    assert ld.returncode == 0, ld.stderr.decode()[-2000:]
    exe = str(tmp_path / "tsan_abi")
    cc = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++"] + common + ["-I" + os.path.join(golden.ROOT, "include"), os.path.join(golden.ROOT, "tests", "support", "tsan_abi.cpp"), lib,
                         "-Wl,-rpath," + str(tmp_path), "-lpthread", "-o", exe], capture_output=True)
    assert cc.returncode == 0, cc.stderr.decode()[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", FXAMD_LIB=lib)
    r = subprocess.run([exe], capture_output=True, env=env, timeout=900)
    err = r.stderr.decode()
    if "FATAL: ThreadSanitizer" in err and ("unexpected memory mapping" in err or "failed to" in err):
        pytest.skip("ThreadSanitizer cannot run in this container: " + err[:200])
    assert b"THREADS OK" in r.stdout, (r.stdout.decode()[-500:], err[-2000:])
    assert "WARNING: ThreadSanitizer" not in err, err[-4000:]


def test_compile_cache_shares_and_releases_programs(built, monkeypatch):
    """fxamd_compile hands an identical (op, pattern) the cached, refcounted program (a loop of scalar calls compiles once); every
    handle is still released with fxamd_program_free; FXAMD_NO_CACHE=1 turns the cache off; programs from blobs are never shared."""
    import forgex_amd as fx
    L = fx.lib()

    def compile_(pat, op):
        h = ctypes.c_void_p()
        st = ctypes.c_int32(0)
        assert L.fxamd_compile(pat, len(pat), op, ctypes.byref(h), ctypes.byref(st)) == 0
        return h, st.value
    monkeypatch.delenv("FXAMD_NO_CACHE", raising=False)
    a, sa = compile_(rb"cache[a-z]+\d+", 0)
    b, sb = compile_(rb"cache[a-z]+\d+", 0)
    c, _ = compile_(rb"cache[a-z]+\d+", 1)          # the other operator: its own program
    d, _ = compile_(rb"cache[a-z]+\d+ ", 0)         # (`.in.` trims the pattern, but the cache key is the text as passed)
    assert a.value == b.value and sa == sb == 0
    assert c.value != a.value and d.value != a.value
    for h in (a, b, c, d):
        L.fxamd_program_free(h)
    e, _ = compile_(rb"cache[a-z]+\d+", 0)          # still cached after every handle was released
    assert e.value == a.value
    L.fxamd_program_free(e)
    # more distinct patterns than the cache holds: the oldest are evicted, nothing breaks, statuses stay right
    hs = [compile_(b"evict%d[a-z]" % i, 0) for i in range(100)]
    assert all(st == 0 for _, st in hs)
    bad, st = compile_(rb"a(", 0)
    assert st == 2
    for h, _ in hs + [(bad, st)]:
        L.fxamd_program_free(h)
    monkeypatch.setenv("FXAMD_NO_CACHE", "1")
    x, _ = compile_(rb"cache[a-z]+\d+", 0)
    y, _ = compile_(rb"cache[a-z]+\d+", 0)
    assert x.value != y.value
    L.fxamd_program_free(x)
    L.fxamd_program_free(y)
    monkeypatch.delenv("FXAMD_NO_CACHE", raising=False)
    p = fx.Program(rb"cache[a-z]+\d+", fx.OP_SEARCH)
    q = fx.Program.from_blob(p.blob(), fx.OP_SEARCH)
    assert q._h.value != p._h.value and q.blob() == p.blob()


def test_compile_cache_survives_concurrent_compile_free_churn(built):
    """Handles are refcounted and shared through the compile cache; fxamd_program_free of the last user handle pins the cached program
    while it looks at its scratch, and a concurrent fxamd_compile may evict it meanwhile (round 2's advisor finding: use after free).
    Eight threads compile and free 150 distinct patterns -- more than the cache holds -- in different orders; every handle must report
    its own pattern's status, and nothing may crash."""
    import threading
    import forgex_amd as fx
    L = fx.lib()
    pats = [(b"churn%d[a-z]+\\d{%d}" % (i, 1 + i % 5), i % 2) for i in range(150)] + [(b"(bad%d" % i, 0) for i in range(10)]
    want = {}
    for pat, op in pats:
        h, st = ctypes.c_void_p(), ctypes.c_int32(0)
        assert L.fxamd_compile(pat, len(pat), op, ctypes.byref(h), ctypes.byref(st)) == 0
        want[(pat, op)] = st.value
        L.fxamd_program_free(h)
    errors = []

    def worker(seed):
        import random
        rng = random.Random(seed)
        held = []
        for _ in range(3000):
            pat, op = rng.choice(pats)
            h, st = ctypes.c_void_p(), ctypes.c_int32(0)
            if L.fxamd_compile(pat, len(pat), op, ctypes.byref(h), ctypes.byref(st)) != 0 or st.value != want[(pat, op)]:
                errors.append((pat, op, st.value))
                return
            if L.fxamd_program_status(h) != want[(pat, op)]:
                errors.append((pat, op, "status of the handle"))
                return
            held.append(h)
            if len(held) > rng.randint(0, 6):
                L.fxamd_program_free(held.pop(rng.randrange(len(held))))
        for h in held:
            L.fxamd_program_free(h)

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
