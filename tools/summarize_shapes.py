#!/usr/bin/env python3
"""Condense one shape's rocprofv3 output (tools/profile_shapes.sh): per-kernel stats, and -- summed over the fx_* kernels of one step --
HBM traffic per step from the PMC passes with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide coalesced
read stream -> doubled; counters are in KiB), next to the step's algorithmic bytes."""
import csv
import glob
import json
import os
import sys

out, tag, shape = sys.argv[1], sys.argv[2], sys.argv[3]


def find(sub, pattern):
    # (the NEWEST file: gpurun merges every call's output into the same scratch directory, so older runs' files may sit next to it)
    hits = sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


bench = None
try:
    for ln in open(os.path.join(out, "bench_under_rocprof.log")):
        if ln.startswith("{"):
            bench = json.loads(ln)
except Exception:
    pass
summary = {"tag": tag, "shape": shape, "bench_under_rocprof": bench}
kt = find("kt", "*kernel_stats.csv")
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if "fx_" in r.get("Name", "")]
    summary["kernel_stats"] = [{k: r.get(k) for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")} for r in rows]
    print("== %s: fx_* kernels (%s)" % (shape, os.path.basename(kt)))
    for r in rows:
        print("  %-90s calls=%s avg=%s ns min=%s max=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"), r.get("MinNs"), r.get("MaxNs")))


def pmc(sub, counter):
    f = find(sub, "*counter_collection.csv")
    per_kernel = {}
    if not f:
        return per_kernel
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter or "fx_" not in r.get("Kernel_Name", ""):
            continue
        per_kernel.setdefault(r["Kernel_Name"], []).append(float(r.get("Counter_Value", 0)))
    return per_kernel


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
# steps of a PMC pass = dispatches of its most-dispatched fx_* kernel (--steps 3 --warmup 1, plus the call that allocates the result buffer in the
# packed shapes: dividing those five calls by four is what made rounds 3..5 report 1.25 x for packed results)
steps = max([len(v) for v in fetch.values()] + [1])
wsteps = max([len(v) for v in write.values()] + [1])
if fetch and write and bench:
    # per step: every fx_* kernel's dispatches / steps
    f_kib = sum(sum(v) for v in fetch.values()) / steps
    w_kib = sum(sum(v) for v in write.values()) / wsteps
    traffic = (2.0 * f_kib + w_kib) * 1024.0
    alg = bench["algorithmic_bytes_per_step"]
    summary.update({"FETCH_SIZE_KiB_per_step_raw": f_kib, "WRITE_SIZE_KiB_per_step": w_kib, "traffic_bytes_per_step": traffic,
                    "algorithmic_bytes_per_step": alg, "traffic_ratio": traffic / alg})
    print("== traffic per step: FETCH_SIZE raw %.0f KiB (x2 gfx950 correction) + WRITE_SIZE %.0f KiB = %.4f GB = %.3f x algorithmic (%.4f GB)" % (
        f_kib, w_kib, traffic / 1e9, traffic / alg, alg / 1e9))
# SQ counters of the dominant fx_* kernel (the pass is optional: tools/profile_shapes.sh <tag> "<shapes>" sq)
sq = {}
fsq = find("pmc_sq", "*counter_collection.csv")
if fsq:
    acc = {}
    for r in csv.DictReader(open(fsq)):
        if "fx_" in r.get("Kernel_Name", ""):
            acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r.get("Counter_Value", 0)))
    names = sorted({k for k, _ in acc}, key=lambda k: -sum(acc.get((k, "SQ_WAVE_CYCLES"), [0])))
    if names:
        dom = names[0]
        sq = {c: sum(v) / len(v) for (k, c), v in acc.items() if k == dom}
        summary["sq_counters_per_launch"] = {"kernel": dom[:160], **sq}
        wc = sq.get("SQ_WAVE_CYCLES") or 1.0
        print("== SQ counters per launch of %s" % dom[:100])
        print("   " + "  ".join("%s %.3g" % (c, v) for c, v in sorted(sq.items())))
        print("   active %.2f  wait_any %.2f  wait_inst %.2f of the wave cycles; LDS bank conflicts %.2f of the LDS-array cycles" % (
            sq.get("SQ_ACTIVE_INST_ANY", 0) / wc, sq.get("SQ_WAIT_ANY", 0) / wc, sq.get("SQ_WAIT_INST_ANY", 0) / wc,
            sq.get("SQ_LDS_BANK_CONFLICT", 0) / (sq.get("SQ_LDS_IDX_ACTIVE") or 1.0)))
fi = find("pmc_inst", "*counter_collection.csv")
if fi:
    acc = {}
    for r in csv.DictReader(open(fi)):
        if "fx_" in r.get("Kernel_Name", ""):
            acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r.get("Counter_Value", 0)))
    names = sorted({k for k, _ in acc}, key=lambda k: -sum(acc.get((k, "SQ_INSTS_VALU"), [0])))
    if names and bench:
        dom = names[0]
        ins = {c: sum(v) / len(v) for (k, c), v in acc.items() if k == dom}
        nbytes = bench["rows"] * bench["row_len"]
        summary["inst_counters_per_launch"] = {"kernel": dom[:160], **ins}
        print("== instructions per launch of %s" % dom[:100])
        print("   " + "  ".join("%s %.3g (%.2f per input byte x64 lanes)" % (c, v, v * 64.0 / nbytes) for c, v in sorted(ins.items())))
if bench:
    print("== step under rocprof: %.4f ms, %.0f GB/s of input, frac of HBM peak %.3f, last_path %s" % (
        bench["ms_per_step"], bench["input_gbs"], bench["frac_of_hbm_peak"], bench["last_path"]))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
