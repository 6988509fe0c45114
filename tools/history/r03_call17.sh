#!/bin/bash
# round 3, GPU call 17: grid rounds of the four-wave half-row kernel, interleaved repetitions (box drift is +-3 % within a call)
OUT=gpurun_out/r03_c17
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3 4 5; do
  for r in 3 8 12 16; do
    FXAMD_HALF_ROUNDS=$r $B > $OUT/rounds_${r}_$rep.json 2> $OUT/rounds_${r}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/rounds_${r}_$rep.json').read().strip().splitlines()[-1]); print('rounds=$r rep$rep', 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'cold', round(d['roofline']['cold_kernel_ms'],4))"
  done
done
