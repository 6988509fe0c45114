#!/usr/bin/env python3
"""Copy the condensed rocprofv3 output of tools/profile_round.sh / tools/profile_shapes.sh from gpurun_out/ (scratch) into profiles/
(tracked): per run the summary (.txt / .json), the kernel-stats CSV and the bench line printed under the profiler; refresh
profiles/pmc_traffic.json from the config summaries.   python tools/collect_profiles.py r03"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
prof = os.path.join(ROOT, "profiles")
traffic_path = os.path.join(prof, "pmc_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_*" % tag))):
    name = os.path.basename(d)[len("prof_"):]   # r03_cfg4, r03_match_cfg3, ...
    for ext in ("txt", "json"):
        src = os.path.join(d, "summary." + ext)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(prof, "%s_summary.%s" % (name, ext)))
    # the kernel-stats CSV the summary was computed from (summary.json names it); without that, the newest one.  Then the check that
    # the tracked pair agrees: every kernel's average duration in the summary must be the CSV's.
    sj0 = os.path.join(d, "summary.json")
    named = json.load(open(sj0)).get("kernel_stats_csv") if os.path.exists(sj0) else None
    ks = [os.path.join(d, named)] if named and os.path.exists(os.path.join(d, named)) else sorted(
        glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if ks:
        dst = os.path.join(prof, "%s_kernel_stats.csv" % name)
        shutil.copy(ks[-1], dst)
        if os.path.exists(sj0):
            import csv
            want = {r.get("Name"): r.get("AverageNs") for r in json.load(open(sj0)).get("kernel_stats", [])}
            have = {r.get("Name"): r.get("AverageNs") for r in csv.DictReader(open(dst))}
            bad = [k for k in want if have.get(k) != want[k]]
            if bad:
                sys.exit("collect_profiles: %s: the kernel-stats CSV and the summary disagree on %r" % (name, bad[:3]))
    log = os.path.join(d, "bench_under_rocprof.log")
    if os.path.exists(log):
        lines = [ln for ln in open(log) if ln.startswith("{")]
        if lines:
            open(os.path.join(prof, "%s_bench_under_rocprof.json" % name), "w").write(lines[-1])
    sj = os.path.join(d, "summary.json")
    if os.path.exists(sj):
        s = json.load(open(sj))
        cfg = s.get("config")
        if cfg and s.get("traffic_bytes_per_launch"):
            traffic[cfg] = s["traffic_bytes_per_launch"]
            traffic.setdefault("_kernels", {})[cfg] = s.get("dominant_kernel", "")[:120]
    print("collected", name)
traffic["_source"] = ("profiles/%s_<cfg>_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --config <cfg>`; FETCH_SIZE x2 per "
                      "MI355X_MICROARCH.md); dominant kernel of each config" % tag)
json.dump(traffic, open(traffic_path, "w"), indent=1)
