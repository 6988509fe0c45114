#!/usr/bin/env python3
"""Census of kernel paths: over the patterns of the reference's own tests (tests/golden/ref_tests.tsv) and of
the fuzz generator (tests/support/fuzz_diff.py), which VALID patterns land on the general kernel (`last_path` 2: one lane per
row, ~0.2 TB/s) on 16-byte-aligned rows, and why?  CPU only: the path is decided from the compiled program's mode and flags exactly
as forgex_amd/csrc/fxamd.hip does (fast_scheme / enqueue_batch).

    python tools/census_paths.py [--fuzz 4000] > profiles/r03_census_paths.md
"""
import argparse
import collections
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import forgex_amd   # noqa: E402
import golden       # noqa: E402
import fuzz_diff    # noqa: E402

F = dict(INIT_ACC=1 << 0, PREFILTER=1 << 1, HAS_SUFFIX=1 << 2, FAST_OK=1 << 3, HAS_R=1 << 4, MATCH_LIT=1 << 5, FAST_UTF8=1 << 6, NFA_SIM=1 << 7,
         CHAIN_OK=1 << 8, CHAIN_UTF8=1 << 9, RAW_BYTES=1 << 10, RAGGED_OK=1 << 11, BYTE_DFA=1 << 12, W16_OK=1 << 13, W16_UTF8=1 << 14,
         BYTE_W16=1 << 15, PREFIX_NEC=1 << 16, OVERLAP_SINK=1 << 17)
MODE = {0: "invalid", 1: "search", 2: "literal", 3: "match"}


def classify(pat, op):
    """-> (category, reason) for 16-byte aligned rows of whole chunks (and ragged rows when they differ)"""
    p = forgex_amd.Program(pat, op)
    if p.status >= 100:
        return "unsupported", "status %d" % p.status
    if p.status != 0:
        return "invalid", "status %d" % p.status
    i = p.info()
    fl, mode = i["flags"], i["mode"]
    if fl & F["NFA_SIM"]:
        return "nfa", "DFA beyond the state limit: NFA simulation"
    if fl & (F["FAST_OK"] | F["W16_OK"] | F["CHAIN_OK"]):
        tab = "8-state v_perm" if fl & F["FAST_OK"] else ("nibble" if fl & F["W16_OK"] else "chain")
        ragged = "" if fl & F["RAGGED_OK"] else " (ragged rows -> general)"
        if fl & (1 << 21):   # FXP_F_PREFIX_CHECK (round 6): rows of up to 256 bytes on the one-launch kernel, the start checked per row; longer rows: general kernel
            tab += " + per-row prefix check (rows > 256 B: general kernel)"
        return "tile", tab + ragged
    # general kernel: why?
    why = []
    if mode == 1:
        if not (fl & F["HAS_R"]):
            why.append("reverse DFA beyond the state limit (restart loop)")
        if fl & F["PREFILTER"]:
            if not (fl & F["PREFIX_NEC"]):
                why.append("prefix literal not proven a necessary beginning")
            elif fl & F["HAS_SUFFIX"]:
                why.append("prefix necessary, but suffix literal not proven a necessary ending / bordered prefix without sink")
            else:
                why.append("bordered prefix literal without overlap sink")
        elif fl & F["HAS_R"]:
            why.append("automata too large for the LDS tables (nA=%d nR=%d)" % (i["nA"], i["nR"]))
    elif mode == 2:
        why.append("literal with a NUL byte / too long for the tables")
    elif mode == 3:
        why.append("`.match.` automaton too large for the LDS tables (nA=%d)" % i["nA"])
    return "general", "; ".join(why) or "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=4000)
    args = ap.parse_args()
    sets = collections.OrderedDict()
    gold = set()
    for _, kind, f in golden.load_ref_tests():
        if kind in ("in", "regex"):
            gold.add((golden.unhx(f[0]), forgex_amd.OP_SEARCH))
        elif kind == "match":
            gold.add((golden.unhx(f[0]), forgex_amd.OP_MATCH))
    sets["patterns of the reference's own tests (in / regex / match records)"] = sorted(gold)
    rng = random.Random(20261003)
    fz = set()
    while len(fz) < args.fuzz:
        pat = fuzz_diff.gen_pattern(rng).encode()
        fz.add((pat, rng.choice([forgex_amd.OP_SEARCH, forgex_amd.OP_SEARCH, forgex_amd.OP_MATCH])))
    sets["fuzz generator (tests/support/fuzz_diff.py gen_pattern, seed 20261003)"] = sorted(fz)
    print("# r03: which valid patterns leave the tile kernels? (census, CPU: program mode + flags as fxamd.hip dispatches them)\n")
    for title, pats in sets.items():
        cat = collections.Counter()
        reasons = collections.Counter()
        examples = {}
        for pat, op in pats:
            c, why = classify(pat, op)
            cat[c] += 1
            if c == "general":
                key = ("search: " if op == forgex_amd.OP_SEARCH else "match: ") + why
                reasons[key] += 1
                examples.setdefault(key, []).append(pat)
        valid = cat["tile"] + cat["general"] + cat["nfa"]
        print("## %s\n" % title)
        print("%d (pattern, operator) pairs: %d invalid, %d beyond the builder's limits, %d valid.\n" % (len(pats), cat["invalid"], cat["unsupported"], valid))
        print("| path on 16-byte-aligned rows | pairs | share of valid |")
        print("|---|---|---|")
        for c, label in (("tile", "tile kernels (`last_path` 1, 3, 5-16)"), ("general", "general kernel (`last_path` 2)"), ("nfa", "NFA simulation (`last_path` 4)")):
            print("| %s | %d | %.1f %% |" % (label, cat[c], 100.0 * cat[c] / max(valid, 1)))
        print("\nWhy the general kernel:\n")
        print("| reason | pairs | examples |")
        print("|---|---|---|")
        for key, k in reasons.most_common():
            ex = ", ".join("`%s`" % e.decode("utf-8", "replace").replace("|", "\\|").replace("\n", "\\n").replace("\r", "\\r").replace("\t", "\\t") for e in examples[key][:4])
            print("| %s | %d | %s |" % (key, k, ex))
        print()


if __name__ == "__main__":
    main()
