#!/bin/bash
# round 3, GPU call 14: 128-byte rows on the half-row kernel (config 5; A/B with FXAMD_NO_HALF128=1 in the same build), tests
OUT=gpurun_out/r03_c14
mkdir -p $OUT
show() { python3 - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2", "step_us", round(d["ms_per_step"]*1e3,2), "settled_us", round(d["settled"]["ms_per_step"]*1e3,2), "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "flags_only_us", round(d["flags_only"]["ms_per_step"]*1e3,2), "parity", d["parity"]["mismatches"], "path", d["roofline"]["kernel"])
except Exception as e:
    print("$2", "FAILED", e)
PY
}
for rep in 1 2 3; do
  python bench.py --config cfg5 --steps 100 --warmup 30 --no-cpu-baseline > $OUT/cfg5_half_$rep.json 2> $OUT/cfg5_half_$rep.err; show $OUT/cfg5_half_$rep.json "cfg5 half128 rep$rep"
  FXAMD_NO_HALF128=1 python bench.py --config cfg5 --steps 100 --warmup 30 --no-cpu-baseline > $OUT/cfg5_one_$rep.json 2> $OUT/cfg5_one_$rep.err; show $OUT/cfg5_one_$rep.json "cfg5 one-launch rep$rep"
done
python bench.py --steps 100 --warmup 30 --no-cpu-baseline > $OUT/cfg3.json 2> $OUT/cfg3.err; show $OUT/cfg3.json "cfg3"
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
