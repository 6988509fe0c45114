#!/bin/bash
# round 3, GPU call 8: where the time of the many-pattern pass on UTF-8 rows goes (kernel trace of tools/exp_multi.py cfg4)
OUT=$PWD/gpurun_out/r03_c8
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_multi -- python3 $REPO/tools/exp_multi.py cfg4 > $OUT/multi.log 2>&1
FXAMD_NO_MULTI=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_seq -- python3 $REPO/tools/exp_multi.py cfg4 > $OUT/seq.log 2>&1
cd $REPO
for d in kt_multi kt_seq; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d"; python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "fx_" in r["Name"]:
        print("  %-100s calls=%s avg_us=%.1f total_ms=%.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
grep -v amdgpu $OUT/multi.log | tail -3; grep -v amdgpu $OUT/seq.log | tail -3
