"""CPU tests (no GPU): the oracle is pinned against every golden vector recorded from the REAL reference
(tests/golden/ref_tests.tsv = all 1496 assertions of the reference's own test programs, plus the reference's actual
outputs incl. from/to), and -- when oracle/_ref exists (this container) -- against the real reference on fuzzed cases."""
import os

import pytest

import golden
import fuzz_diff


@pytest.fixture(scope="module")
def pairs():
    return golden.expected_lines_from_golden(golden.load_ref_tests())


def test_golden_file_is_complete():
    recs = golden.load_ref_tests()
    kinds = {}
    for _, k, _ in recs:
        kinds[k] = kinds.get(k, 0) + 1
    assert len(recs) == 1496
    assert kinds == {"in": 88, "match": 857, "regex": 52, "prefix": 65, "suffix": 102, "validate": 207, "error": 125}
    # the reference passes its own tests: expected == recorded actual
    for prog, kind, f in recs:
        if kind in ("in", "match"):
            assert f[2] == f[3], (prog, f)
        elif kind == "validate":
            assert f[1] == f[2], (prog, f)
        elif kind == "error":
            assert f[1] == f[2], (prog, f)


def test_oracle_reproduces_every_golden_vector(built, pairs):
    out = golden.run_protocol(golden.ORACLE_CLI, [c for c, _ in pairs])
    bad = [(c, e, a) for (c, e), a in zip(pairs, out) if not golden.line_matches(e, a)]
    assert not bad, bad[:5]
    assert len(pairs) > 1500


def test_baseline_config1_cases_verbatim(built):
    """BASELINE.json config 1 (`\\d{3}-\\d{4}` .match.) -- the two cases of reference test_case_003.f90:27-28."""
    out = golden.run_protocol(golden.ORACLE_CLI, [("M", rb"\d{3}-\d{4}", b"100-1002"), ("M", rb"\d{3}-\d{4}", b"1234567")])
    assert out == ["M T", "M F"]


def test_appendix_a_quirks(built):
    """Reference quirks the build must keep (each line was observed on the real reference)."""
    cases = [
        (("M", b"ab[cd]", b"ab"), "M T"), (("I", b"aa[bc]", b"aaab"), "I F"), (("I", b"aa[bc]", b"xaab"), "I T"),
        (("R", b"^abc$", b"def\nabc"), "R 4 7 4 0 0A616263"), (("R", b"abc$", b"abc\ndef"), "R 1 4 4 0 6162630A"),
        (("I", b"b*", b"aaa"), "I F"), (("I", b"b*", b""), "I T"), (("I", b"$", b"abc"), "I F"), (("I", b"^", b"abc"), "I F"),
        (("R", b"foo(bar|baz)", b"xxfoobarbaz"), "R 3 8 6 0 666F6F626172"), (("R", b"a*", b"baaa"), "R 2 4 3 0 616161"),
        (("M", b"abc", b"abc "), "M F"), (("M", b"(|^)", b""), "M F"), (("M", b"(^|)", b""), "M T"),
        (("I", b"/", b"\xc0\xaf"), "I F"), (("M", b".", b"\xc0\xaf"), "M T"), (("I", rb"\s", b" "), "I F"),
    ]
    out = golden.run_protocol(golden.ORACLE_CLI, [c for c, _ in cases])
    assert out == [e for _, e in cases]


@pytest.mark.skipif(not os.path.exists(golden.REF_DRIVER), reason="the real reference (oracle/_ref, built from /root/reference by oracle/Makefile) is absent here; NOTE the oracle and the product share frontend.cpp, so without this test a front-end bug is caught only by the golden vectors (validate / error / prefix / suffix records)")
def test_oracle_equals_real_reference_on_fuzz(built):
    cases = fuzz_diff.gen_cases(101, 3000)
    a = golden.run_protocol(golden.ORACLE_CLI, cases)
    b = golden.run_protocol(golden.REF_DRIVER, cases)
    diffs = [(c, x, y) for c, x, y in zip(cases, a, b) if x != y]
    assert not diffs, diffs[:5]
