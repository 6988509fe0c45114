#!/bin/bash
OUT=${1:-gpurun_out/r02d}
mkdir -p $OUT
B="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-extras"
$B --config cfg3 > $OUT/one_g2.json 2> $OUT/one_g2.err
FXAMD_ONE_GRID=8 $B --config cfg3 > $OUT/one_g8.json 2> $OUT/one_g8.err
FXAMD_MULTIPASS=1 $B --config cfg3 > $OUT/multi.json 2> $OUT/multi.err
FXAMD_HALF=1 $B --config cfg3 > $OUT/half.json 2> $OUT/half.err
FXAMD_HALF=1 $B --config cfg3 --flags-only > $OUT/half_flags.json 2> $OUT/half_flags.err
$B --config cfg4 > $OUT/one_cfg4.json 2> $OUT/one_cfg4.err
FXAMD_ONE_GRID=8 $B --config cfg4 > $OUT/one_cfg4_g8.json 2> $OUT/one_cfg4_g8.err
for f in one_g2 one_g8 multi half half_flags one_cfg4 one_cfg4_g8; do python3 - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", "step_us", round(d["ms_per_step"]*1e3,1), "settled", round(d["settled"]["ms_per_step"]*1e3,1), "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,1), "parity", d["parity"]["mismatches"], d["roofline"]["kernel"])
except Exception as e:
    print("$f ERR", e)
PY
done
FXAMD_HALF=1 python -m pytest tests -m gpu -x -q -k "full_size or one_launch or fast_kernel_fuzz or batch_shapes" > $OUT/pytest_half.log 2>&1; tail -3 $OUT/pytest_half.log
