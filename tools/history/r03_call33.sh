#!/bin/bash
# round 3, GPU call 33: the final build once more after the last change (`.match.` three-buffer pass from 96-byte rows on): GPU suite, smoke, `.match.` shapes,
# the driver's protocol
OUT=gpurun_out/r03_c33
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
for s in match_cfg3 match_cfg5 match_utf8 match_cfg1x; do
  python tools/bench_shapes.py --shape $s --steps 60 --warmup 20 > $OUT/$s.json 2> $OUT/$s.err
  python3 -c "
import json
d=json.loads(open('$OUT/$s.json').read().strip().splitlines()[-1]); print('$s us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])"
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; python3 -c "
import json
d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); print('driver protocol', 'value', round(d['value']), 'step_ms', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'parity', d['parity']['mismatches'])"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"
