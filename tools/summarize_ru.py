#!/usr/bin/env python3
"""Condense `make -C forgex_amd/csrc resource-usage-all` (the -Rpass-analysis=kernel-resource-usage remarks of every device object, build_ru/*.txt)
into one table: kernel, VGPRs, AGPRs, SGPR spills, VGPR spills, scratch bytes per lane, occupancy, static LDS.

    python tools/summarize_ru.py [--dir forgex_amd/csrc/build_ru] [--spills-only] > profiles/rNN_resource_usage.txt

`load(dir)` is what tests/test_host_logic.py uses: a kernel on a default dispatch path with scratch fails the CPU suite.
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "occ"),
          ("SGPRs Spill", "sspill"), ("VGPRs Spill", "vspill"), ("LDS Size [bytes/block]", "lds"))


def demangle(names):
    """c++filt in one go; keeps only `kernel<template arguments>`."""
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    except (OSError, subprocess.CalledProcessError):
        return list(names)
    res = []
    for s in out[:len(names)]:
        s = re.sub(r"^void ", "", s)
        depth, cut = 0, len(s)
        for i, ch in enumerate(s):   # cut the argument list: the first '(' outside the template brackets
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        res.append(s[:cut].replace("(bool)", "").replace("(int)", "").replace(" ", ""))
    return res


def load(ru_dir):
    """[{obj, kernel, vgpr, agpr, sgpr, sspill, vspill, scratch, occ, lds}] over every build_ru/*.txt"""
    rows = []
    for path in sorted(glob.glob(os.path.join(ru_dir, "*.txt"))):
        obj = os.path.basename(path)[:-4]
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"obj": obj, "mangled": m.group(1)}
                rows.append(cur)
                continue
            if cur is None:
                continue
            for key, short in FIELDS:
                m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
                if m:
                    cur[short] = int(m.group(1))
    for r, k in zip(rows, demangle([r["mangled"] for r in rows])):
        r["kernel"] = k
    return rows


def main():
    ru_dir = os.path.join(ROOT, "forgex_amd", "csrc", "build_ru")
    spills_only = "--spills-only" in sys.argv
    if "--dir" in sys.argv:
        ru_dir = sys.argv[sys.argv.index("--dir") + 1]
    rows = load(ru_dir)
    kernels = [r for r in rows if "scratch" in r]
    with_scratch = [r for r in kernels if r["scratch"] > 0]
    with_sspill = [r for r in kernels if r.get("sspill", 0) > 0]
    print("# kernel resource usage, gfx950 (hipcc -O3 -Rpass-analysis=kernel-resource-usage; `make -C forgex_amd/csrc resource-usage-all`)")
    print("# %d kernels in %d objects; %d with scratch (VGPR spills or stack), %d with SGPR spills" %
          (len(kernels), len({r["obj"] for r in kernels}), len(with_scratch), len(with_sspill)))
    print("# %-12s %5s %5s %6s %6s %7s %4s %6s  %s" % ("object", "VGPR", "AGPR", "Sspill", "Vspill", "scratch", "occ", "LDS", "kernel"))
    for r in sorted(kernels, key=lambda r: (r["obj"], r["kernel"])):
        if spills_only and r["scratch"] == 0 and r.get("vspill", 0) == 0:
            continue
        print("  %-12s %5d %5d %6d %6d %7d %4d %6d  %s" % (r["obj"], r.get("vgpr", -1), r.get("agpr", 0), r.get("sspill", 0), r.get("vspill", 0), r["scratch"],
                                                          r.get("occ", 0), r.get("lds", 0), r["kernel"]))


if __name__ == "__main__":
    main()
