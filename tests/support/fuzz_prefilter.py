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


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    rng = random.Random(seed)
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
