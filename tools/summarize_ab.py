#!/usr/bin/env python3
"""One line per run of the interleaved A/B experiments of a round: gpurun_out/c<N>/*.json (tools/r05_job.sh ab: / abs: / abl: steps, tools/exp_*.sh)
-> a text table for profiles/.   python tools/summarize_ab.py <first call> <last call> > profiles/r05_ab_runs.txt"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for c in range(lo, hi + 1):
    d = os.path.join(ROOT, "gpurun_out", "c%d" % c)
    files = sorted(glob.glob(os.path.join(d, "*.json")))
    rows = []
    for f in files:
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        name = os.path.basename(f)[:-5]
        if "shape" in j:
            rows.append("  %-44s %-18s L %4d rows %9d  %9.2f us  %5.0f GB/s of input  frac %.3f  path %s" % (
                name, j["shape"], j["row_len"], j["rows"], j["ms_per_step"] * 1e3, j["input_gbs"], j["frac_of_hbm_peak"], j["last_path"]))
        elif "metric" in j:
            r = j.get("roofline") or {}
            cfg = (j.get("config") or {}).get("workload", "")[:5]
            rows.append("  %-44s %-18s step %9.2f us  kernel %9.2f us  value %5.0f GB/s  kernel frac %.3f  parity %s" % (
                name, cfg, j["ms_per_step"] * 1e3, (r.get("kernel_ms") or 0) * 1e3, j["value"], r.get("frac") or 0, (j.get("parity") or {}).get("mismatches")))
    if rows:
        print("== gpurun call r05_c%d (one allocation: the arms of an experiment alternate; `on` / `base` / `default` = the build or setting that ships,"
              " `off` / an ENV name = the other arm)" % c)
        print("\n".join(rows))
