#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh: per-kernel stats (count, avg/min/max ns) and, for the dominant
kernel, HBM traffic per launch from the PMC passes with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts
half of a wide coalesced read stream -> doubled; counters are in KiB)."""
import csv
import glob
import json
import os
import sys

out, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]


def find(sub, pattern):
    # (the NEWEST file: gpurun merges every call's output into the same scratch directory, so older runs' files may sit next to it)
    hits = sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


summary = {"tag": tag, "config": cfg}
kt = find("kt", "*kernel_stats.csv")
if kt:
    rows = list(csv.DictReader(open(kt)))
    summary["kernel_stats_csv"] = os.path.relpath(kt, out)   # tools/collect_profiles.py copies exactly this file
    summary["kernel_stats"] = [{k: r.get(k) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows]
    print("== kernel stats (%s)" % kt)
    for r in rows:
        print("  %-70s calls=%s avg=%s ns  min=%s max=%s  %s%%" % (r.get("Name", "")[:70], r.get("Calls"), r.get("AverageNs"), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))


def pmc(sub):
    f = find(sub, "*counter_collection.csv")
    acc = {}
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        key = (name, r.get("Counter_Name"))
        acc.setdefault(key, []).append(float(r.get("Counter_Value", 0)))
    return acc


fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
dom = None   # dominant kernel = the fx_* kernel with the largest fetched volume
best = -1.0
for (name, ctr), vals in fetch.items():
    if ("fx_search_fast" in name or "fx_search_one" in name or "fx_search_span" in name or "fx_search_tiny" in name or "fx_match" in name or "fx_general" in name or
            "fx_nfa" in name) and sum(vals) > best:
        dom, best = name, sum(vals)
if dom:
    f = fetch.get((dom, "FETCH_SIZE"), [])
    w = write.get((dom, "WRITE_SIZE"), [])
    if f and w:
        fetch_kib = sum(f) / len(f)
        write_kib = sum(w) / len(w)
        traffic = (2.0 * fetch_kib + write_kib) * 1024.0
        summary["dominant_kernel"] = dom
        summary["FETCH_SIZE_KiB_per_launch_raw"] = fetch_kib
        summary["WRITE_SIZE_KiB_per_launch"] = write_kib
        summary["traffic_bytes_per_launch"] = traffic
        print("== traffic of %s: FETCH_SIZE raw %.0f KiB (x2 gfx950 correction), WRITE_SIZE %.0f KiB -> %.3f GB per launch" % (
            dom[:60], fetch_kib, write_kib, traffic / 1e9))
    sqv = {c: sum(v) / len(v) for (n, c), v in sq.items() if n == dom}
    if sqv:
        summary["sq_counters_per_launch"] = sqv
        print("== SQ counters per launch:", json.dumps(sqv))
        wc = sqv.get("SQ_WAVE_CYCLES") or 1.0
        print("   active %.2f  wait_any %.2f  wait_inst %.2f of the wave cycles; LDS bank conflicts %.2f of the LDS-array cycles" % (
            sqv.get("SQ_ACTIVE_INST_ANY", 0) / wc, sqv.get("SQ_WAIT_ANY", 0) / wc, sqv.get("SQ_WAIT_INST_ANY", 0) / wc,
            sqv.get("SQ_LDS_BANK_CONFLICT", 0) / (sqv.get("SQ_LDS_IDX_ACTIVE") or 1.0)))
    inst = {c: sum(v) / len(v) for (n, c), v in pmc("pmc_inst").items() if n == dom}
    if inst:
        summary["inst_counters_per_launch"] = inst
        # input bytes per launch from the bench line printed under the profiler
        nbytes = None
        try:
            for ln in open(os.path.join(out, "bench_under_rocprof.log")):
                if ln.startswith("{"):
                    c = json.loads(ln)["config"]
                    nbytes = c["rows_per_gpu"] * c["row_len"]
        except Exception:
            pass
        print("== instructions per launch: " + "  ".join("%s %.3g%s" % (c, v, (" (%.2f per input byte x 64 lanes)" % (v * 64.0 / nbytes)) if nbytes else "")
                                                          for c, v in sorted(inst.items())))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
