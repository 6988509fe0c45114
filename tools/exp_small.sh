#!/bin/bash
# config 3 at the per-GPU row counts of the north star's split batch (10 M rows over 2, 4, 8 GPUs): the half-row pipeline (first pass +
# gated follow-up) against the one-launch kernel (FXAMD_NO_HALF=1), interleaved.   bash tools/exp_small.sh <tag> "<rows> ..." "<ENV or ->  ..."
TAG=${1:-small}; ROWS=${2:-"1250000 2500000 5000000"}; ARMS=${3:-"- FXAMD_NO_HALF=1"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for n in $ROWS; do for rep in 1 2; do for arm in $ARMS; do
  if [ "$arm" = "-" ]; then python bench.py --config cfg3 --rows $n --steps 200 --warmup 50 --no-cpu-baseline --no-extras --no-parity > $OUT/cfg3_${n}_base_$rep.json 2> $OUT/cfg3_${n}_base_$rep.err; f=$OUT/cfg3_${n}_base_$rep.json
  else env $arm python bench.py --config cfg3 --rows $n --steps 200 --warmup 50 --no-cpu-baseline --no-extras --no-parity > $OUT/cfg3_${n}_${arm%%=*}_$rep.json 2> $OUT/cfg3_${n}_${arm%%=*}_$rep.err; f=$OUT/cfg3_${n}_${arm%%=*}_$rep.json; fi
  python3 - $f "cfg3 n=$n $arm rep $rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("%-44s step %8.2f us  kernel %8.2f us  value %.0f GB/s  path %s" % (sys.argv[2], d["ms_per_step"] * 1e3, r.get("kernel_ms", 0) * 1e3, d["value"], d.get("config", {}).get("last_path", d.get("last_path"))))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
done; done; done
