#!/bin/bash
# round 3, GPU call 36: last check of the shipped library (rebuilt after a comment-only change): GPU suite, smoke, one driver-protocol bench line
OUT=gpurun_out/r03_c36
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python3 -c "
import json
d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); print('driver protocol', 'value', round(d['value']), 'step_ms', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'parity', d['parity']['mismatches'], 'cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"
