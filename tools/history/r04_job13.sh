OUT=gpurun_out/r04_c13; mkdir -p $OUT
for g in 0 2 3 4 6 8; do
  for rep in 1 2; do
    FXAMD_ONE_GRID=$g python bench.py --config cfg2 --no-cpu-baseline --no-extras --no-parity > $OUT/cfg2_g${g}_$rep.json 2>/dev/null
    python3 -c "
import json;d=json.loads(open('$OUT/cfg2_g${g}_$rep.json').read().strip().splitlines()[-1]);print('cfg2 grid blocks/CU', $g, 'rep', $rep, 'step', round(d['ms_per_step']*1e3,2),'us kernel', round(d['roofline']['kernel_ms']*1e3,2), 'settled', round(d['settled']['ms_per_step']*1e3,2))"
  done
done
