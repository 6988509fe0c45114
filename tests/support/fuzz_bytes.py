#!/usr/bin/env python3
"""Differential fuzz of the byte-level tables (FXP_F_BYTE_DFA): the host walker answering from them (FX_HW_BYTES=1, exceptions
falling through to the decode path, exactly the device pipeline) against the oracle, on texts rich in multi-byte, overlong and
structurally invalid UTF-8.  Development aid; tests/test_host_logic.py runs a bounded slice."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden import ORACLE_CLI, run_protocol  # noqa: E402
from fuzz_diff import gen_pattern  # noqa: E402

HOST_WALK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_walk")
PIECES = [b"a", b"b", b"c", b"x", b"z", b"0", b"1", b"9", b" ", b"\n", b".", b"-", "あ".encode(), "ん".encode(), "ぁ".encode(), "ア".encode(),
          "α".encode(), "ω".encode(), "é".encode(), "　".encode(), "\U0001F600".encode(), "￿".encode(), "߿".encode(),
          "ࠀ".encode(), "\U00010000".encode(), "\U0010FFFF".encode(), b"\xc0\xaf", b"\xc1\xa1", b"\xe0\x80\xaf", b"\xf0\x80\x80\xaf",
          b"\xf4\x90\x80\x80", b"\xf7\xbf\xbf\xbf", b"\xed\xa0\x80", b"\x80", b"\xbf", b"\xc3", b"\xe3\x81", b"\xe3", b"\xf0\x9f\x98", b"\xf0\x9f",
          b"\xf8", b"\xff", b"\xfe", b"\x00", b"\x1f", b"\x7f"]
EXTRA_PATTERNS = ["[α-ωぁ-ん]+", "[ぁ-ん]+[0-9]*", ".", ".+", "[^a]+", "\\S+", "\\W+x", "[a-z]+\\d+", "(あ|ア)+", "[^ぁ-ん]+", "\\x{FFFF}", "[\\x{10000}-\\x{10FFFF}]+",
                  "a.c", "é+", "[à-ÿ]+", "[\\x{80}-\\x{7FF}]+", "x*[ -~]+", "\\s+"]


def gen_text(rng):
    n = rng.randint(1, 14)
    return b"".join(rng.choice(PIECES) for _ in range(n))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rng = random.Random(seed)
    cases = []
    for _ in range(n):
        pat = rng.choice(EXTRA_PATTERNS) if rng.random() < 0.4 else gen_pattern(rng)
        txt = gen_text(rng)
        if rng.random() < 0.3:   # valid-only text: the rows the byte tables answer themselves
            txt = b"".join(rng.choice(PIECES[:26]) for _ in range(rng.randint(1, 14)))
        cases.append((rng.choice(["I", "M", "R", "R"]), pat.encode(), txt))
    a = run_protocol(ORACLE_CLI, cases)
    os.environ.setdefault("FX_HW_BYTES", "1")
    b = run_protocol(HOST_WALK, cases)
    bad = unsup = 0
    for c, x, y in zip(cases, a, b):
        if y.startswith("U "):
            unsup += 1
            continue
        if x != y:
            bad += 1
            if bad <= 30:
                print("DIFF %s pat=%r txt=%r\n   oracle: %s\n   bytes : %s" % (c[0], c[1].decode(), c[2], x, y))
    print("seed %d: %d cases, %d unsupported, %d differences" % (seed, n, unsup, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
