#!/usr/bin/env python3
"""Differential fuzzing: the same (op, pattern, text) cases through two protocol speakers
(by default oracle/oracle_cli vs the real reference oracle/_ref/ref_driver).  Development aid and
the generator behind tests/golden/fuzz_cases.tsv."""
import random
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden import ORACLE_CLI, REF_DRIVER, run_protocol, hx  # noqa: E402

ATOMS = ["a", "b", "c", "ab", "x", "0", "1", "9", " ", ".", "\\d", "\\D", "\\w", "\\W", "\\s", "\\S", "\\t", "\\n", "\\r",
         "[a-c]", "[^a]", "[abc]", "[0-9]", "[a-z]", "[^0-9a]", "[\\d]", "[\\w-]", "[a\\]]", "[b-]", "[\\t ]", "^", "$",
         "\\.", "\\\\", "\\x41", "\\x{3042}", "[\\x61-\\x63]", "あ", "[ぁ-ん]", "α", "[α-ω]", "é", "\\(", "\\|", "-", ",", "}", "]"]
SUFFIX = ["", "", "", "*", "+", "?", "{2}", "{1,2}", "{0,1}", "{2,}", "{,2}", "{0}", "{3}"]
TEXT_ALPH = [b"a", b"b", b"c", b"x", b"0", b"1", b"9", b" ", b"\n", b"\r", b"\t", b".", b"-", b"A", "あ".encode(), "ん".encode(),
             "α".encode(), "ω".encode(), "é".encode(), b"\x00", b"\x80", b"\xc0", b"\xe3\x81", b"\xff", b"\xf0\x9f\x98\x80",
             b"\\", b"(", b"|", b"]", b"}", b",", b"\x1f", b"\x0b"]


def gen_pattern(rng, depth=0):
    n = rng.choice([1, 1, 2, 2, 3, 4])
    parts = []
    for _ in range(n):
        r = rng.random()
        if r < 0.18 and depth < 3:
            inner = gen_pattern(rng, depth + 1)
            if rng.random() < 0.5:
                inner = inner + "|" + gen_pattern(rng, depth + 1)
            atom = "(" + inner + ")"
        else:
            atom = rng.choice(ATOMS)
        parts.append(atom + rng.choice(SUFFIX))
    p = "".join(parts)
    if depth == 0 and rng.random() < 0.15:
        p = p + "|" + gen_pattern(rng, 1)
    return p


def gen_garbage_pattern(rng):
    chars = list("ab01 .*+?|()[]{}^$\\-,xdDwWsSnt") + ["あ", "{1,2}", "[a-", "\\x4", "\\x{", "--", "[^", "]]"]
    return "".join(rng.choice(chars) for _ in range(rng.randint(0, 8)))


def gen_text(rng, maxlen=12):
    return b"".join(rng.choice(TEXT_ALPH) for _ in range(rng.randint(0, maxlen)))


def gen_cases(seed, n):
    rng = random.Random(seed)
    cases = []
    for _ in range(n):
        pat = gen_garbage_pattern(rng) if rng.random() < 0.2 else gen_pattern(rng)
        pb = pat.encode()
        if rng.random() < 0.05:
            pb = pb + b" " * rng.randint(1, 2)
        if rng.random() < 0.03:
            pb = b" " + pb
        txt = gen_text(rng)
        # bias texts towards things the pattern could match
        if rng.random() < 0.5:
            lits = [a.encode() for a in ATOMS if not a.startswith(("\\", "[", "^", "$", "."))]
            txt = b"".join(rng.choice(lits + TEXT_ALPH[:12]) for _ in range(rng.randint(0, 10)))
        op = rng.choice(["I", "M", "R", "R", "L", "V"])
        cases.append((op, pb, b"" if op in ("L", "V") else txt))
    return cases


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    a_exe = sys.argv[3] if len(sys.argv) > 3 else ORACLE_CLI
    b_exe = sys.argv[4] if len(sys.argv) > 4 else REF_DRIVER
    cases = gen_cases(seed, n)
    a = run_protocol(a_exe, cases)
    b = run_protocol(b_exe, cases)
    bad = 0
    for c, x, y in zip(cases, a, b):
        if x != y:
            bad += 1
            if bad <= 40:
                print("DIFF %s pat=%r txt=%r\n   A: %s\n   B: %s" % (c[0], c[1], c[2], x, y))
    print("seed %d: %d cases, %d differences" % (seed, n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
