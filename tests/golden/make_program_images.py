"""Writes tests/golden/program_images.tsv: SHA-256 (first 16 hex digits) and size of the program image the host compiler emits for a fixed list
of patterns (the bench configs, the GPU tests' patterns, shapes from the reference's own tests).  The images are the wire format
between ranks: a refactor of the compiler must leave them byte-identical; a deliberate change of the tables bumps
FXP_VERSION and regenerates this file:

    python tests/golden/make_program_images.py
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ["FXAMD_NO_CACHE"] = "1"

PATTERNS = [r"[a-z]+\d+", r"foo(bar|baz)", r"\d{3}-\d{4}", "[α-ωぁ-ん]+", "[ぁ-ん]+", r"zz+", r"[0-9]$", "needle", r"(ab|cd)+\d", r"aa[bc]",
            "ω[α-ω]", r"\w+x", r"q[u-z]+", "foobar", r"(ab|cd)+e", r"x[yz]+\d", r"[a-f]+ [g-z]", r"q\d", r"[0-9]+", r"k+ ", r"a.c\d", r"${2,}",
            r"[a-z]+", r"\s+", r"[^a-z]+", r".*", r"^abc$", r"\x{3042}+", r"[\x{1F600}-\x{1F64F}]", r"a{2,5}b?", r"(a|b|c|d|e|f|g)+h",
            r"\w+@\w+\.(com|org|net)", r"[A-Z][a-z]{2,8} [0-9]{1,3}", r"[ab]*a[ab]{20}", "a(", r"a{2,1}", "", " abc  ", r"\n", r"[\n]", r"\S+\s\S+"]


def rows():
    import forgex_amd as fx
    out = []
    for p in PATTERNS:
        for op in (fx.OP_SEARCH, fx.OP_MATCH):
            q = fx.Program(p, op)
            image = q.blob() if q.status == 0 else b""
            out.append((p.encode().hex() or "-", op, q.status, len(image), hashlib.sha256(image).hexdigest()[:16] if image else "-"))
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "program_images.tsv"), "w") as f:
        f.write("# pattern (hex, - = empty)\top\tstatus\timage bytes\tsha256 of the image (16 hex digits)\n")
        for r in rows():
            f.write("\t".join(str(x) for x in r) + "\n")
    print("wrote", len(rows()), "records")
