#!/bin/bash
# round 3, GPU call 18: grid size of the one-launch kernel (FXAMD_ONE_GRID = blocks per CU in the grid; default: what is resident when the launch has
# exception queues, else 8), interleaved repetitions, configs 5 / 4 / 2
OUT=gpurun_out/r03_c18
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for cfg in cfg5 cfg4 cfg2; do
for rep in 1 2 3; do
  for g in 0 6 12 24 36 48; do
    if [ $g = 0 ]; then unset FXAMD_ONE_GRID; else export FXAMD_ONE_GRID=$g; fi
    $B --config $cfg > $OUT/${cfg}_g${g}_$rep.json 2> $OUT/${cfg}_g${g}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/${cfg}_g${g}_$rep.json').read().strip().splitlines()[-1]); print('$cfg grid=$g rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2))"
  done
done
done
unset FXAMD_ONE_GRID
