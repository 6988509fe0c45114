#!/usr/bin/env python3
"""Differential fuzz of the compile-time proof "candidate-list search == brute-force search": patterns with literal prefixes and
suffixes; the host walker searches by brute force wherever the program carries tile-kernel tables (FX_HW_FAST=1, what the tile
kernels do on pure-ASCII rows) and must agree with the oracle, which follows the reference's candidate-list driver."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden import ORACLE_CLI, run_protocol  # noqa: E402
from fuzz_diff import gen_pattern  # noqa: E402

HOST_WALK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_walk")
LITS = ["a", "ab", "abc", "aa", "aba", "abab", "foo", "x", "xy", "0", "12", "a b", "-", "ba", "b", "c", "ca", "id=", "zz", "aab"]
MIDS = [".*", ".+", "[a-c]*", "\\d+", "[a-z]+", "(b|c)*", "x?", "\\s*", "[^a]*", "(ab)*", ".", "\\w{1,3}", "(a|b)+", ""]
ALPH = list("abcxyz012 -=") + ["ab", "aa", "foo", "id=", "aba", "zz", "12"]


# round 4: literals with non-ASCII characters (the proofs used to be restricted to ASCII prefix / suffix literals); the texts mix the
# literals' own characters with other multi-byte characters, overlong encodings of them (C1 A1 = 'a', E0 8E B1 = alpha: the reference
# decodes those arithmetically to the SAME code point, but the driver's INDEX works on bytes) and structure errors
U_LITS = ["α", "αβ", "ぁ", "é", "αa", "aα", "夢", "胡蝶", "αα", "ああ", "aé", "é ", "x", "ab", "-", "ぁa", "βα"]
U_MIDS = [".{1,7}", ".*", ".+", "[α-ω]*", "[ぁ-ん]+", "(α|β)*", "x?", "\\s*", "[^a]*", "", ".", "\\w{1,3}", "[a-zα-ω]+"]
U_ALPH = ["α", "β", "ぁ", "あ", "é", "夢", "胡", "蝶", "胡蝶", "a", "b", "x", " ", "-", "αβ", "αα", "ああ"]
U_JUNK = [b"\xc1\xa1", b"\xe0\x8e\xb1", b"\xce", b"\xb1", b"\xe3\x81", b"\xff", b"\xf0\x9f\x98\x80", b"\xc0\xaf", b"\xe5\xa4"]


def gen_case_utf8(rng):
    r = rng.random()
    if r < 0.5:
        pat = rng.choice(U_LITS) + rng.choice(U_MIDS) + rng.choice(U_LITS)
    elif r < 0.7:
        pat = rng.choice(U_LITS) + rng.choice(U_MIDS)
    else:
        pat = rng.choice(U_LITS) + rng.choice(U_MIDS) + rng.choice(U_LITS) + rng.choice(U_MIDS) + rng.choice(U_LITS)
    lits = [c for c in pat if ord(c) > 127 or c.isalnum() or c in " -"]
    pieces = []
    for _ in range(rng.randint(0, 12)):
        q = rng.random()
        if q < 0.55 and lits:
            pieces.append(rng.choice(lits).encode())
        elif q < 0.9:
            pieces.append(rng.choice(U_ALPH).encode())
        else:
            pieces.append(rng.choice(U_JUNK))
    return (rng.choice(["I", "R", "R"]), pat.encode(), b"".join(pieces))


REPS = ["{2,}", "{1,2}", "+", "{2}", "{3,}", "{1,3}"]   # a repeated tail makes prefix and suffix literals overlap in the shortest match


def gen_case(rng):
    if os.environ.get("FX_FUZZ_UTF8"):
        return gen_case_utf8(rng)
    if os.environ.get("FX_FUZZ_OVERLAP") and rng.random() < 0.7:
        body = rng.choice(LITS)
        if rng.random() < 0.5:
            body = "(" + body + ")"
        pat = rng.choice(["", rng.choice(LITS)]) + body + rng.choice(REPS) + rng.choice(["", rng.choice(LITS), rng.choice(MIDS) + rng.choice(LITS)])
        bits = [c for c in pat if c.isalnum() or c in " =-"]
        txt = "".join(rng.choice(bits + ALPH[:4]) for _ in range(rng.randint(1, 14))) if bits else "ab"
        return (rng.choice(["I", "R", "R"]), pat.encode(), txt.encode())
    r = rng.random()
    if r < 0.45:
        pat = rng.choice(LITS) + rng.choice(MIDS) + rng.choice(LITS)
    elif r < 0.65:
        pat = rng.choice(LITS) + rng.choice(MIDS)
    elif r < 0.8:
        pat = rng.choice(LITS) + rng.choice(MIDS) + rng.choice(LITS) + rng.choice(MIDS) + rng.choice(LITS)
    else:
        pat = gen_pattern(rng)
    txt = "".join(rng.choice(ALPH) for _ in range(rng.randint(0, 14)))
    if rng.random() < 0.5:   # bias towards near-matches: pieces of the pattern's literals
        bits = [c for c in pat if c.isalnum() or c in " =-"]
        txt = "".join(rng.choice(bits + ALPH[:6]) for _ in range(rng.randint(1, 16))) if bits else txt
    return (rng.choice(["I", "R", "R"]), pat.encode(), txt.encode())


# round 6: programs with FXP_F_PREFIX_CHECK -- a prefix literal the compile-time proof does NOT cover (not a necessary beginning, or bordered without an overlap
# state); the tile kernels run their tables all the same and check per row that the brute-force start is a candidate (row_engine.hpp prefix_start_ok), every
# other row with a hit goes to the statement-level driver.  Patterns: random ones (fuzz_diff.gen_pattern) and hand-made shapes whose program carries the flag;
# texts: the pattern's own literal characters repeated and overlapped, so that several prefix occurrences, overlapping ones and matches that do NOT start at an
# occurrence all happen.
CHECK_SHAPES = [r"(}[abc]){2}\d*c{2,}", r"(\t{3}[a-z]){2}", r" {3}\\{2}[a-z]{2,}", r"(ab|abc)x", r"(a|ab)(c|bcd)", r"a?ab+", r"(aa|a)b", r"x*yz", r"(ab)*abc", r"a{1,2}ab",
                r"(-|--)a", r"(\t\t|\t)x+", r"aa(a|b)", r"(xy){1,2}z", r"ab?ab", r"(zz|z)\d", r"\d?12", r"(ab){2,}c?", r"a*aab", r"(b|)aba",
                r"A{1,2}bb", r"(ab{2}){1,2}-a{2}", r"a{1,2}b{2}", r"ab+c{0,1}bc", r"x(ab|b)b", r"(ab{2}){1,2}-α{2}", r"\x41{1,2}α{2}"]   # (the last rows: suffix literals the proof does not cover)
_info = None


def prefix_check_flag(pat):
    """does the compiled search program carry FXP_F_PREFIX_CHECK (bit 21)?  (tests/support/libhostwalk.so, hw_info)"""
    global _info
    import ctypes
    if _info is None:
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhostwalk.so"))
        lib.hw_info.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
        _info = lib.hw_info
    info = (ctypes.c_int32 * 8)()
    _info(pat, len(pat), 0, info)
    return info[5] == 0 and (info[1] & (1 << 21)) != 0


def gen_case_check(rng, pool):
    pat = rng.choice(pool)
    lits = sorted({c for c in pat.decode("utf-8", "replace") if c.isalnum() or c in " -}\t"} | set("ab1c "))
    lits = [("\t" if c == "t" and "\\t" in pat.decode("utf-8", "replace") else c) for c in lits]
    pieces = []
    for _ in range(rng.randint(0, 14)):
        q = rng.random()
        if q < 0.7:
            c = rng.choice(lits)
            pieces.append((c * rng.randint(1, 4)).encode())
        elif q < 0.9:
            pieces.append(rng.choice(["ab", "aab", "abab", "}a}b", "12", "\t\t\t\ta", "   \\\\ab", "--a", "xyxyz", "zz1"]).encode())
        else:
            pieces.append(rng.choice([b"\xce\xb1", b"\xff", b"\n", b"\0"]))
    return (rng.choice(["I", "R", "R"]), pat, b"".join(pieces))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    rng = random.Random(seed)
    if os.environ.get("FX_FUZZ_CHECK"):
        pool = [p.encode() for p in CHECK_SHAPES if prefix_check_flag(p.encode())]
        tried = 0
        while len(pool) < 60 and tried < 20000:   # random patterns whose program carries the flag
            tried += 1
            p = gen_pattern(rng).encode()
            if prefix_check_flag(p):
                pool.append(p)
        print("FX_FUZZ_CHECK: %d patterns with FXP_F_PREFIX_CHECK (%d hand-made shapes carry it)" % (len(pool), sum(1 for p in CHECK_SHAPES if prefix_check_flag(p.encode()))))
        cases = [gen_case_check(rng, pool) for _ in range(n)]
    else:
        cases = [gen_case(rng) for _ in range(n)]
    a = run_protocol(ORACLE_CLI, cases)
    os.environ["FX_HW_FAST"] = "1"
    b = run_protocol(HOST_WALK, cases)
    bad = 0
    for c, x, y in zip(cases, a, b):
        if y.startswith("U "):
            continue
        if x != y:
            bad += 1
            if bad <= 30:
                print("DIFF %s pat=%r txt=%r\n   oracle: %s\n   brute : %s" % (c[0], c[1].decode(), c[2], x, y))
    print("seed %d: %d cases, %d differences" % (seed, n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
