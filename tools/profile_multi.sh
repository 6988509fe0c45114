#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of fx_search_multi (6 patterns, one pass) against one pipeline per pattern, on 2 M rows of config 3 (512 MB of rows)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_multi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fused -- python3 $REPO/tools/exp_multi.py cfg3 2000000 > $OUT/fused.log 2>&1
FXAMD_NO_MULTI=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/seq -- python3 $REPO/tools/exp_multi.py cfg3 2000000 > $OUT/seq.log 2>&1
cd $REPO && python3 - <<'PY' > $OUT/summary.txt
import csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "prof_multi")
for tag in ("fused", "seq"):
    f = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        name = r["Kernel_Name"].split("(")[0]
        if "fx_" not in name:
            continue
        acc.setdefault(name, []).append(float(r["Counter_Value"]))
    print("== %s: FETCH_SIZE per launch (KiB raw; x2 per the gfx950 note = bytes/1024), rows = 2 000 000 x 256 B = 500 000 KiB" % tag)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print("   %-72s launches %4d  avg raw %9.0f KiB  -> %.3f x the rows" % (k[:72], len(v), sum(v) / len(v), 2 * sum(v) / len(v) / 500000.0))
PY
cat $OUT/summary.txt
