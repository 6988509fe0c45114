#!/bin/bash
# round 3, GPU call 16: tuning of the four-wave half-row kernel in one allocation: grid rounds (FXAMD_HALF_ROUNDS), load cache policy (nt vs default build)
OUT=gpurun_out/r03_c16
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
show() { python3 - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2", "step_ms", round(d["ms_per_step"],4), "settled", round(d["settled"]["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), "cold", round(d["roofline"]["cold_kernel_ms"],4))
except Exception as e:
    print("$2", "FAILED", e)
PY
}
for rep in 1 2; do
  for r in 0 2 3 4 6 8; do
    FXAMD_HALF_ROUNDS=$r $B > $OUT/rounds_${r}_$rep.json 2> $OUT/rounds_${r}_$rep.err; show $OUT/rounds_${r}_$rep.json "rounds=$r rep$rep"
  done
  FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_aux0.so $B > $OUT/aux0_$rep.json 2> $OUT/aux0_$rep.err; show $OUT/aux0_$rep.json "aux0 (default policy loads) rep$rep"
done
