"""Config-scale parity inputs: rows of the five BASELINE configs, invalid-UTF-8 mutations of config-4 rows
and the Appendix A quirk probes at row lengths 64..256 -- every row a pure function of a section name and an index, so that the
container (where the REAL reference answers them: tests/golden/make_config_goldens.py) and the GPU box (where the HIP path is
checked against those answers: tests/test_gpu_parity.py) regenerate identical bytes.  tests/golden/config_rows.tsv holds the
reference's answers and a CRC of every section's bytes.

Test infrastructure: nothing under forgex_amd/ imports this module.
"""
import zlib

import numpy as np
import torch

from forgex_amd import synth

MASK = (1 << 64) - 1


def _mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


class _Rng:
    """counter-based: value k of stream (salt, index)"""

    def __init__(self, salt, index):
        self.base = _mix64((salt << 32) ^ index)
        self.k = 0

    def next(self, mod):
        self.k += 1
        return _mix64(self.base + self.k * 0xD1342543DE82EF95) % mod


def _cfg_rows(cfg, idx):
    return synth.rows(cfg, torch.as_tensor(np.asarray(idx, dtype=np.int64))).numpy()


# ---- sections of config rows: (name, op, pattern, row_len, list of row indices) ----------------------------------------------------
def _cfg4_corrupt_indices(count, scan=160000):
    idx = torch.arange(0, scan, dtype=torch.int64)
    r = synth._rowhash(idx, synth.SEEDS["cfg4"], 0)
    corrupt = (synth._lsr(r, 20) % 100) == 0
    return idx[corrupt][:count].tolist()


def config_sections():
    out = []
    n1, _ = synth.SHAPES["cfg1"]
    out.append(("cfg1", "M", synth.PATTERNS["cfg1"], 8, list(range(0, 2000))))   # the config's 1000 rows + 1000 more of the same generator
    n2, _ = synth.SHAPES["cfg2"]
    out.append(("cfg2", "R", synth.PATTERNS["cfg2"], 64, list(range(0, 2048)) + list(range(n2 - 2048, n2))))
    n3, _ = synth.SHAPES["cfg3"]
    out.append(("cfg3", "R", synth.PATTERNS["cfg3"], 256, list(range(0, 4096)) + list(range(n3 - 2048, n3))))
    n4, _ = synth.SHAPES["cfg4"]
    out.append(("cfg4", "R", synth.PATTERNS["cfg4"], 192, list(range(0, 2048)) + list(range(n4 - 2048, n4))))
    out.append(("cfg4_corrupt", "R", synth.PATTERNS["cfg4"], 192, _cfg4_corrupt_indices(1024)))
    n5, _ = synth.SHAPES["cfg5"]
    idx5 = []
    for shard in range(8):   # the first 512 rows of every shard of the 8-GPU partition
        idx5 += list(range(shard * (n5 // 8), shard * (n5 // 8) + 512))
    out.append(("cfg5", "R", synth.PATTERNS["cfg5"], 128, idx5))
    return out


def config_section_rows(name):
    for nm, op, pat, L, idx in config_sections():
        if nm == name:
            cfg = "cfg4" if nm.startswith("cfg4") else nm
            return _cfg_rows(cfg, idx)
    raise KeyError(name)


# ---- invalid-UTF-8 mutations -------------------------------------------------------------------------------------------------------
# patterns whose answer depends on how bytes >= 0x80 are read (reference utf8_m.f90:168-246: structural validity only, every byte of an
# invalid sequence is one U+FFFF symbol; :338-430: arithmetic decode of overlong / surrogate / out-of-range forms)
MUTATION_PATTERNS = [
    synth.PATTERNS["cfg4"],          # [α-ωぁ-ん]+
    "[ぁ-ん][^α-ω]",                 # a negated class right after a 3-byte character
    "\\x{FFFF}+",                    # the replacement symbol itself, repeated
    "[α-ω][ぁ-ん].",                  # `.` = U+0020..U+10FFFF takes U+FFFF
]
MUTATION_LENGTHS = [64, 128, 192, 256]
MUTATION_ROWS_PER_SECTION = 256      # 4 patterns x 4 lengths x 256 = 4096 records

_INSERTS = [bytes([0xC0, 0xAF]), bytes([0xE0, 0x80, 0xAF]), bytes([0xF0, 0x80, 0x80, 0xAF]),   # overlong '/'
            bytes([0xED, 0xA0, 0x80]), bytes([0xED, 0xBF, 0xBF]),                                # surrogates
            bytes([0xF5, 0x80, 0x80, 0x80]), bytes([0xF7, 0xBF, 0xBF, 0xBF]),                    # beyond U+10FFFF
            bytes([0xF8]), bytes([0xFC, 0x80]), bytes([0xFE]), bytes([0xFF]),                    # bytes that never start a character
            bytes([0xEF, 0xBF, 0xBF]), bytes([0xE3, 0x81]), bytes([0xCE]),                       # U+FFFF itself; truncated leads
            bytes([0x80]), bytes([0xBF, 0xBF]), bytes([0x00]), bytes([0x0A]), bytes([0x20])]     # stray continuations, NUL, LF, blank


def _mutate(row, rng):
    b = bytearray(row)
    L = len(b)
    for _ in range(1 + rng.next(4)):
        kind = rng.next(6)
        pos = rng.next(L)
        if kind == 0:     # overwrite with a byte >= 0x80
            b[pos] = 0x80 + rng.next(0x80)
        elif kind == 1:   # overwrite with an ASCII letter
            b[pos] = 97 + rng.next(26)
        elif kind == 2:   # delete a byte (the rest moves left, a blank fills the end)
            del b[pos]
            b.append(32)
        elif kind == 3:   # insert a crafted sequence (the tail falls off the row)
            ins = _INSERTS[rng.next(len(_INSERTS))]
            b[pos:pos] = ins
            del b[L:]
        elif kind == 4:   # cut the row's last character in half: the row ends inside a sequence
            ins = _INSERTS[rng.next(3)] if rng.next(2) else bytes([0xE3, 0x81, 0x82])
            k = 1 + rng.next(len(ins) - 1) if len(ins) > 1 else 1
            b[L - k:L] = ins[:k]
        else:             # a run of continuation bytes
            k = 1 + rng.next(5)
            for j in range(pos, min(L, pos + k)):
                b[j] = 0x80 + rng.next(0x40)
    return bytes(b)


def mutation_sections():
    out = []
    for pi, pat in enumerate(MUTATION_PATTERNS):
        for L in MUTATION_LENGTHS:
            out.append(("mut_p%d_L%d" % (pi, L), "R", pat, L, list(range(MUTATION_ROWS_PER_SECTION))))
    return out


def mutation_section_rows(name):
    pi = int(name.split("_")[1][1:])
    L = int(name.split("_")[2][1:])
    n = MUTATION_ROWS_PER_SECTION
    # base text: two consecutive config-4 rows glued (192 bytes of text + blanks each), cut to L
    base_idx = [100000 + 1000 * pi + 2 * i for i in range(n)]
    a = _cfg_rows("cfg4", base_idx)
    b = _cfg_rows("cfg4", [j + 1 for j in base_idx])
    rows = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        text = bytes(a[i][:190]) + bytes(b[i])   # (190: the first row's multi-byte body without its two pad blanks)
        rng = _Rng(0x6D75 + pi * 16 + MUTATION_LENGTHS.index(L), i)
        rows[i] = np.frombuffer(_mutate(text[:L], rng), dtype=np.uint8)
    return rows


# ---- Appendix A quirk probes, embedded in rows of 64..256 bytes --------------------------------------------------------------------
# (pattern, short text) pairs taken from SURVEY.md Appendix A items 1-10 (anchors and NUL sentinels, empty matches, longest-from-
# leftmost, the non-overlapping prefix list, all-literal patterns, the .match. gate, trim, negation quirks, UTF-8 stepping)
PROBES = [
    ("^abc$", b"def\nabc"), ("abc$", b"abc\ndef"), ("^abc", b"abc"), ("^", b"abc"), ("$", b"abc"), ("x*$", b"abc"),
    ("b*", b"aaa"), ("a*", b"baaa"), ("foo(bar|baz)", b"xxfoobarbaz"), ("[a-z]+\\d+", b"ab12  cd345"),
    ("aa[bc]", b"aaab"), ("aa[bc]", b"xaab"), ("--[a-z]+", b"---ab"), ("abab\\d", b"ababab1"), ("zz\\d+", b"zzz9"),
    ("/", b"\xc0\xaf"), (".", b"\xc0\xaf"), ("\\x{FFFF}", b"\xff"), ("ab[cd]", b"ab"), ("ab(c|d)e", b"ab"), ("a{2}[xy]", b"aa"),
    ("abc", b"abc "), ("[^a-z]", b"\t"), ("[^a-z]", b"\x1f"), ("[^\\t]", b"\n"), ("[^\\t]", b"\x08"), ("\\S", b"\x0b"), ("\\S", b"\x08"),
    ("\\D", b"\t"), ("\\W+", b"\x1f\x1f"), ("\\n", b"\r\n"), ("[\\n]", b"\r"), ("\\s", "　".encode()), ("\\s+x", b" \t\n\r\x0c x"),
    ("(|^)a", b"a"), ("(^|)a", b"a"), ("id=\\d+;", b"id=42;"), ("abc.*xyz", b"abcabcxyzxyz"), ("aa.*bb", b"aaabbb"),
    ("\\d{3}-\\d{4}", b"123-4567"), ("[い]{6}", "いいいいいい".encode()), ("[α-ω]+", "xαβγx".encode()), ("ん+$", "あんん".encode()),
]
PROBE_LENGTHS = [64, 100, 128, 255, 256]
PROBE_FILLERS = [b"x", b" ", b"\n", b"q\xe3\x81\x82"]   # filler text the probe is embedded in (repeated)
PROBE_PLACES = ["start", "middle", "end"]


def probe_cases():
    """-> list of (pattern str, op, row bytes): `R` for every probe, plus `M` for the .match.-gate probes."""
    out = []
    for pat, txt in PROBES:
        for L in PROBE_LENGTHS:
            for fi, fill in enumerate(PROBE_FILLERS):
                for place in PROBE_PLACES:
                    room = L - len(txt)
                    if place == "start":
                        before = 0
                    elif place == "end":
                        before = room
                    else:
                        before = room // 2
                    # whole filler units in front of the probe (the rest is 'q'): the probe never starts inside a character;
                    # behind it the filler is cut where the row ends, possibly inside one
                    pre = b"q" * (before % len(fill)) + fill * (before // len(fill))
                    post_n = room - before
                    post = (fill * (post_n // len(fill) + 1))[:post_n]
                    row = pre + txt + post
                    assert len(row) == L
                    out.append((pat, "R", row))
        if pat in ("ab[cd]", "ab(c|d)e", "a{2}[xy]", "abc", "\\d{3}-\\d{4}", "[い]{6}"):
            for L in PROBE_LENGTHS:
                row = txt + b" " * (L - len(txt))
                out.append((pat, "M", row))
                row2 = (txt * (L // len(txt) + 1))[:L]
                out.append((pat, "M", row2))
    return out


def crc_of(rows):
    return zlib.crc32(np.ascontiguousarray(rows).tobytes()) & 0xFFFFFFFF


# ---- round 5: the kernels round 4 added, pinned to the REAL reference (VERDICT r04 "Next round" item 3) --------------------------------
# (a) rows of 2..32 bytes (`fx_match_tiny` / `fx_search_tiny`, last_path 17; with spans: the one-launch kernel's ragged instantiations),
# (b) chain- and nibble-table programs over 256- and 128-byte rows (half-row first passes on those tables), (c) a chain-table program
# over rows of 400 and 1024 bytes (128-byte segments), (d) the speculative forward pass at 64 / 128 / 256 bytes: the match at the row's
# first character, at its second, later, behind a broken sequence, nowhere.
TINY_LENGTHS = [2, 3, 5, 8, 12, 20, 31, 32]
TINY_ROWS = 576
TINY_PATTERNS = [("M", "\\d+-?\\d*"), ("R", "[a-z]+\\d+"), ("R", "\\d+$")]


def _tiny_rows(L, pi, n):
    if L == 8 and pi == 0:   # BASELINE config 1's generator (its pattern is the section's)
        return _cfg_rows("cfg1", list(range(5000, 5000 + n)))
    alpha = [b"0123456789", b"0123456789-", b"abcdefghijklmnopqrstuvwxyz0123456789 ", b"abc  019", b"0123456789abc\n"]
    rows = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        rng = _Rng(0x7469 + 64 * pi + L, i)
        a = alpha[rng.next(len(alpha))]
        b = bytearray(a[rng.next(len(a))] for _ in range(L))
        k = rng.next(8)
        if k == 0:
            b[L - 1] = 48 + rng.next(10)           # a digit at the row's last byte
        elif k == 1:
            b[0] = 97 + rng.next(26)               # a letter at its first
        elif k == 2 and L >= 3:
            b[rng.next(L - 1)] = 45                # a '-'
        elif k == 3:
            b[rng.next(L)] = [0, 10, 13, 32, 9][rng.next(5)]
        rows[i] = np.frombuffer(bytes(b), dtype=np.uint8)
    return rows


def tiny_sections():
    out = []
    for L in TINY_LENGTHS:
        for pi, (op, pat) in enumerate(TINY_PATTERNS):
            if L == 8 and pi == 0:
                pat = synth.PATTERNS["cfg1"]
            out.append(("tiny_%s%d_L%d" % (op, pi, L), op, pat, L, TINY_ROWS, lambda L=L, pi=pi: _tiny_rows(L, pi, TINY_ROWS)))
    return out


# chain tables only (17 / 23 states), nibble tables (`\d{3}-\d{4}` 10 / 9 states; the e-mail pattern's class-level automata fit them)
TABLE_PATTERNS = [("R", "[a-z]{6}\\d{1,3}[a-z ]{6}"), ("M", "[a-z ]{6}[a-z ]*\\d{0,3}[a-z ]{6}[a-z ]*"), ("R", "\\d{3}-\\d{4}"),
                  ("R", "[a-z0-9]+@[a-z0-9]+\\.[a-z]{2,4}")]
TABLE_ROWS = 192
_PLANTS = [b"123-4567", b"12-34567", b"999-0000x", b"ab@cd.com", b"x1@y2.org9", b"@a.bc", b"abcdef12ghijkl", b"abcdefg123 hijkl ", b"abcde1fghijk"]


def _table_rows(L, pi, n):
    cfg = "cfg3" if L == 256 else "cfg5"
    base = _cfg_rows(cfg, [777000 + 1000 * pi + i for i in range(n)])
    rows = base.copy()
    for i in range(n):
        rng = _Rng(0x7462 + 16 * pi + (L >> 7), i)
        if pi == 1 and rng.next(2) == 0:      # `.match.` of the 23-state pattern: letters and blanks, at most one short digit run
            b = bytearray(97 + rng.next(26) if rng.next(8) else 32 for _ in range(L))
            if rng.next(3):
                k = 6 + rng.next(L - 16)
                for j in range(1 + rng.next(3 + rng.next(2))):
                    b[k + j] = 48 + rng.next(10)
            rows[i] = np.frombuffer(bytes(b), dtype=np.uint8)
        elif rng.next(3) == 0:
            pl = _PLANTS[rng.next(len(_PLANTS))]
            k = rng.next(L - len(pl) + 1)
            rows[i, k:k + len(pl)] = np.frombuffer(pl, dtype=np.uint8)
    return rows


def table_sections():
    out = []
    for pi, (op, pat) in enumerate(TABLE_PATTERNS):
        for L in (256, 128):
            out.append(("tab_p%d_L%d" % (pi, L), op, pat, L, TABLE_ROWS, lambda L=L, pi=pi: _table_rows(L, pi, TABLE_ROWS)))
    return out


LONG_CHAIN = [(400, 96), (1024, 48)]


def _long_rows(L, n):
    per = (L + 255) // 256
    base = _cfg_rows("cfg3", [888000 + i for i in range(n * per)]).reshape(n, per * 256)[:, :L].copy()
    for i in range(n):
        rng = _Rng(0x6C6F + L, i)
        if rng.next(2) == 0:   # a match of the 17-state pattern somewhere, also across the 128-byte segment borders
            pl = _PLANTS[6 + rng.next(2)]
            k = [rng.next(L - len(pl) + 1), 128 * (1 + rng.next((L >> 7) - 1)) - rng.next(len(pl)), L - len(pl)][rng.next(3)]
            k = max(0, min(L - len(pl), k))
            base[i, k:k + len(pl)] = np.frombuffer(pl, dtype=np.uint8)
    return base


def long_sections():
    return [("long_chain_L%d" % L, "R", TABLE_PATTERNS[0][1], L, n, lambda L=L, n=n: _long_rows(L, n)) for L, n in LONG_CHAIN]


SPEC_LENGTHS = [64, 128, 256]
SPEC_ROWS = 256


def _spec_rows(L, n):
    base_idx = [300000 + 2 * i for i in range(n)]
    a = _cfg_rows("cfg4", base_idx)
    b = _cfg_rows("cfg4", [j + 1 for j in base_idx])
    junk = [bytes([0x80]), bytes([0xE3, 0x81]), bytes([0xCE]), bytes([0xFF]), bytes([0xC0, 0xAF]), bytes([0xF0, 0x9F]), bytes([0xBF, 0xBF])]
    rows = np.empty((n, L), dtype=np.uint8)
    for i in range(n):
        rng = _Rng(0x7370 + L, i)
        text = bytes(a[i][:190]) + bytes(b[i][:190])
        kind = rng.next(6)
        if kind == 0:     # the match starts at the row's first character
            row = text
        elif kind == 1:   # ... at its second
            row = bytes([97 + rng.next(26)]) + text
        elif kind == 2:   # later
            row = bytes(97 + rng.next(26) if rng.next(5) else 32 for _ in range(2 + rng.next(L // 2))) + text
        elif kind == 3:   # behind a broken sequence
            row = junk[rng.next(len(junk))] + text
        elif kind == 4:   # a broken sequence ends it early; another match follows
            k = 2 * (1 + rng.next(6))
            row = text[:k] + junk[rng.next(len(junk))] + b"x" + text[k:]
        else:             # nowhere
            row = bytes(97 + rng.next(26) if rng.next(6) else [32, 10, 0][rng.next(3)] for _ in range(L))
        rows[i] = np.frombuffer((row + b" " * L)[:L], dtype=np.uint8)
    return rows


def spec_sections():
    return [("spec_L%d" % L, "R", synth.PATTERNS["cfg4"], L, SPEC_ROWS, lambda L=L: _spec_rows(L, SPEC_ROWS)) for L in SPEC_LENGTHS]


# ---- round 5: OVERLONG encodings.  The reference's decoder is arithmetic (src/essential/utf8_m.f90:338-430): `C0 80` is U+0000 (a line start for
# `^`), `C1 A1` is `a`, `E0 80 B1` and `F0 80 80 B1` are `1` -- bytes >= 0x80 that decode to code points BELOW 0x80.  Rows of config-3 text with such
# sequences (and valid characters, broken leads, LF) planted at the row's start, its end, across the 128- and 256-byte borders and anywhere. ----
OVL_LENGTHS = [64, 128, 256, 400]
OVL_ROWS = 160
OVL_PATTERNS = [("caret", r"^[a-z]+"), ("cfg3", r"[a-z]+\d+"), ("a", r"a+")]
_OVL = [b"\xc0\x80", b"\xc1\xa1", b"\xc0\xb1", b"\xc1\xa1\xc1\xa2\xc0\xb7", b"\xe0\x80\x80", b"\xe0\x81\xa1", b"\xe0\x80\xb1", b"\xf0\x80\x80\x80", b"\xf0\x80\x81\xa1",
        b"\xf0\x80\x80\xb1", b"\xc0", b"\xc1", b"\xe0\x80", b"\xf0\x80\x80", b"\xc2\x80", b"\xe0\xa0\x80", "あ".encode(), b"\n", b"\xc0\x8a", b"\xc1\xa1\xe0\x80\xb1"]


def _ovl_rows(L, n):
    per = (L + 255) // 256
    base = _cfg_rows("cfg3", [777000 + i for i in range(n * per)]).reshape(n, per * 256)[:, :L].copy()
    for i in range(n):
        rng = _Rng(0x6F76 + L, i)
        for _ in range(1 + rng.next(3)):
            pl = _OVL[rng.next(len(_OVL))]
            where = rng.next(5)
            k = [0, L - len(pl), 128 * (1 + rng.next(max(1, (L >> 7) - 1))) - rng.next(len(pl) + 1), rng.next(L - len(pl) + 1), rng.next(L - len(pl) + 1)][where]
            k = max(0, min(L - len(pl), k))
            base[i, k:k + len(pl)] = np.frombuffer(pl, dtype=np.uint8)
            if rng.next(3) == 0 and k + len(pl) + 3 <= L:   # letters and a digit right behind it: a match that starts at the planted character
                base[i, k + len(pl):k + len(pl) + 3] = np.frombuffer(b"ab7", dtype=np.uint8)
    return base


def ovl_sections():
    return [("ovl_%s_L%d" % (nm, L), "R", pat, L, OVL_ROWS, lambda L=L: _ovl_rows(L, OVL_ROWS)) for nm, pat in OVL_PATTERNS for L in OVL_LENGTHS]


_ROWS_CACHE = {}


def _cached(name, getter):
    def get():
        if name not in _ROWS_CACHE:
            _ROWS_CACHE[name] = getter()
        return _ROWS_CACHE[name]
    return get


def all_sections():
    """(name, op, pattern, row_len, n_rows, rows-getter) of every batch section (the probes are per-record cases)."""
    out = []
    for nm, op, pat, L, idx in config_sections():
        out.append((nm, op, pat, L, len(idx), lambda nm=nm: config_section_rows(nm)))
    for nm, op, pat, L, idx in mutation_sections():
        out.append((nm, op, pat, L, len(idx), lambda nm=nm: mutation_section_rows(nm)))
    out += tiny_sections() + table_sections() + long_sections() + spec_sections() + ovl_sections()
    return [(nm, op, pat, L, n, _cached(nm, g)) for nm, op, pat, L, n, g in out]


def load_fixture(path):
    """-> (sections: name -> int64 array [n, 3] of flag/from/to, crcs: name -> crc, probes: int64 array [m, 3])"""
    sections, crcs = {}, {}
    with open(path) as f:
        for ln in f:
            if ln.startswith("#crc"):
                _, nm, v = ln.split()
                crcs[nm] = int(v, 16)
                continue
            if ln.startswith("#") or not ln.strip():
                continue
            nm, i, fl, a, b = ln.rstrip("\n").split("\t")
            sections.setdefault(nm, []).append((int(i), int(fl), int(a), int(b)))
    res = {}
    for nm, recs in sections.items():
        recs.sort()
        assert [r[0] for r in recs] == list(range(len(recs))), nm
        res[nm] = np.array([r[1:] for r in recs], dtype=np.int64)
    return res, crcs
