#!/bin/bash
# small batches on the span kernel: rounds of the resident blocks (FXAMD_HALF_ROUNDS) against the built-in rule, interleaved.
#   bash tools/exp_rounds.sh <tag> "<shape> ..." "<rows> ..." "<rounds> ..."
TAG=${1:-rounds}; SHAPES=${2:-"rows_64"}; ROWS=${3:-"1000000 4000000"}; ROUNDS=${4:-"0 1 2 3 4"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for sh in $SHAPES; do for n in $ROWS; do for rep in 1 2; do for r in $ROUNDS; do
  FXAMD_HALF_ROUNDS=$r python tools/bench_shapes.py --shape $sh --rows $n --steps 200 --warmup 50 > $OUT/${sh}_${n}_${r}_$rep.json 2> $OUT/${sh}_${n}_${r}_$rep.err
  python3 - $OUT/${sh}_${n}_${r}_$rep.json "$sh n=$n rounds=$r rep $rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s L %4d  %8.2f us  %.0f GB/s  frac %.3f  path %s" % (sys.argv[2], d["row_len"], d["ms_per_step"] * 1e3, d["input_gbs"], d["frac_of_hbm_peak"], d["last_path"]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
done; done; done; done
