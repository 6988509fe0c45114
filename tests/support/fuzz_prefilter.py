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


def gen_case(rng):
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
