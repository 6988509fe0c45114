"""Helpers shared by the tests: golden-vector loading and the I/M/R/V/L line protocol spoken by
oracle/oracle_cli (the C++ restatement) and oracle/_ref/ref_driver (the real reference)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_CLI = os.path.join(ORACLE_DIR, "oracle_cli")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")


def hx(b):
    if isinstance(b, str):
        b = b.encode("utf-8")
    return b.hex().upper() if b else "-"


def unhx(h):
    return b"" if h == "-" else bytes.fromhex(h)


def load_ref_tests():
    """Yield (program, kind, fields) from tests/golden/ref_tests.tsv."""
    out = []
    with open(os.path.join(GOLDEN, "ref_tests.tsv")) as f:
        for ln in f:
            if ln.startswith("#") or not ln.strip():
                continue
            parts = ln.rstrip("\n").split("\t")
            out.append((parts[0], parts[1], parts[2:]))
    return out


def run_protocol(exe, cases, timeout=600):
    """cases: list of (op, pattern_bytes, text_bytes) -> list of output lines."""
    inp = "".join("%s %s %s\n" % (op, hx(p), hx(t)) for op, p, t in cases)
    r = subprocess.run([exe], input=inp.encode(), capture_output=True, timeout=timeout)
    lines = r.stdout.decode().splitlines()
    if len(lines) != len(cases):
        raise RuntimeError("%s answered %d lines for %d cases (rc=%d, stderr=%s)" % (
            exe, len(lines), len(cases), r.returncode, r.stderr.decode()[-500:]))
    return lines


def expected_lines_from_golden(records):
    """Translate golden records into (case, expected protocol line) pairs."""
    pairs = []
    for prog, kind, f in records:
        if kind == "in":
            pat, txt, _exp, got, frm, to, length, status = f
            pairs.append((("I", unhx(pat), unhx(txt)), "I " + got))
            pairs.append((("R", unhx(pat), unhx(txt)), None if False else ("R %s %s %s %s" % (frm, to, length, status))))
        elif kind == "match":
            pat, txt, _exp, got = f
            pairs.append((("M", unhx(pat), unhx(txt)), "M " + got))
        elif kind == "regex":
            pat, txt, _exp, got, frm, to, length, status = f
            pairs.append((("R", unhx(pat), unhx(txt)), "R %s %s %s %s %s" % (frm, to, length, status, got)))
        elif kind == "validate":
            pat, _exp, got = f
            pairs.append((("V", unhx(pat), b""), "V " + got))
        elif kind == "error":
            pat, _exp, got, _msg = f
            pairs.append((("R", unhx(pat), b""), "R* status=" + got))
        elif kind in ("prefix", "suffix"):
            pat, _exp, got = f
            pairs.append((("L", unhx(pat), b""), ("L* %s=" % kind) + got))
    return pairs


def line_matches(expected, actual):
    if expected.startswith("R* status="):
        return actual.split()[4] == expected.split("=")[1]
    if expected.startswith("L* prefix="):
        a = actual.split()
        return a[1] == "T" and a[3] == expected.split("=")[1] or (a[1] == "F" and expected.split("=")[1] == "-")
    if expected.startswith("L* suffix="):
        a = actual.split()
        return a[1] == "T" and a[4] == expected.split("=")[1] or (a[1] == "F" and expected.split("=")[1] == "-")
    if expected.startswith("R ") and len(expected.split()) == 5:
        return actual.split()[:5] == expected.split()
    return expected == actual
