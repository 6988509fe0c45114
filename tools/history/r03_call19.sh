#!/bin/bash
# round 3, GPU call 19: the grid rules (one-launch kernel: one round of resident blocks per FXAMD_ONE_ROUND_MB of rows; half-row kernel: one round per
# 225 MB) against fixed grids, interleaved repetitions; then the GPU suite
OUT=gpurun_out/r03_c19
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
one() {   # tag cfg
  $B --config $2 > $OUT/$2_$1.json 2> $OUT/$2_$1.err
  python3 -c "
import json
d=json.loads(open('$OUT/$2_$1.json').read().strip().splitlines()[-1]); print('$2 $1', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac', d['roofline']['frac'])"
}
for rep in 1 2 3; do
  for mb in 0 225 800 100000; do
    if [ $mb = 0 ]; then unset FXAMD_ONE_ROUND_MB; else export FXAMD_ONE_ROUND_MB=$mb; fi
    one mb${mb}_$rep cfg5
  done
  unset FXAMD_ONE_ROUND_MB
  one default_$rep cfg2
  one default_$rep cfg4
  for r in 0 3 8 12; do
    if [ $r = 0 ]; then unset FXAMD_HALF_ROUNDS; else export FXAMD_HALF_ROUNDS=$r; fi
    one rounds${r}_$rep cfg3
  done
  unset FXAMD_HALF_ROUNDS
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "pytest rc $?"
