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


def all_sections():
    """(name, op, pattern, row_len, n_rows, rows-getter) of every batch section (the probes are per-record cases)."""
    out = []
    for nm, op, pat, L, idx in config_sections():
        out.append((nm, op, pat, L, len(idx), lambda nm=nm: config_section_rows(nm)))
    for nm, op, pat, L, idx in mutation_sections():
        out.append((nm, op, pat, L, len(idx), lambda nm=nm: mutation_section_rows(nm)))
    return out


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
