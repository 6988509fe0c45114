#!/bin/bash
# round 3, GPU call 25: the final build -- GPU suite, rocprofv3 (kernel trace + stats, FETCH_SIZE / WRITE_SIZE / SQ passes) of configs 3, 4, 5, 2,
# bench lines of every config, the driver's protocol twice, smoke
OUT=gpurun_out/r03_c25
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
for cfg in cfg3 cfg4 cfg5 cfg2; do
  bash tools/profile_round.sh r03_$cfg $cfg > $OUT/prof_$cfg.log 2>&1; tail -3 $OUT/prof_$cfg.log | cut -c1-300
done
for cfg in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 200 --warmup 30 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", "step_us", round(d["ms_per_step"]*1e3,2), "settled_us", round(d["settled"]["ms_per_step"]*1e3,2), "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "cold_us", round(d["roofline"]["cold_kernel_ms"]*1e3,2), "flags_only_us", round(d["flags_only"]["ms_per_step"]*1e3,2), "parity", d["parity"]["mismatches"], "cpu", d["cpu_baseline"]["value"], (d["cpu_baseline"].get("gpu_vs_reference_on_sample") or {}).get("mismatches"))
PY
done
for rep in 1 2; do python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_$rep.json 2> $OUT/bench_driver_$rep.err; python3 -c "
import json
d=json.loads(open('$OUT/bench_driver_$rep.json').read().strip().splitlines()[-1]); print('driver protocol rep$rep', 'value', round(d['value']), 'step_ms', round(d['ms_per_step'],4), 'settled', round(d['settled']['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"; done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"
