#!/bin/bash
# round 3, GPU call 23: the three-buffer aligned forward walk (FX_FWD_PIPE3) with and without the direct entry for clustered starts (FX_FWD_DIRECT),
# against the two-buffer loop behind the window (libforgex_amd_few.so) and the previous commit; the left-half re-walk skip of the half-row kernel
# (config 3); interleaved repetitions; then the GPU suite
OUT=gpurun_out/r03_c23
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for lib in libforgex_amd_prev.so libforgex_amd_few.so libforgex_amd_nd.so libforgex_amd.so; do
    FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg4 > $OUT/cfg4_${lib}_$rep.json 2> $OUT/cfg4_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/cfg4_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg4 $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
    FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape utf8_192_clean --steps 60 --warmup 20 > $OUT/clean_${lib}_$rep.json 2> $OUT/clean_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/clean_${lib}_$rep.json').read().strip().splitlines()[-1]); print('utf8_192_clean $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3))"
  done
  for cfg in cfg3 cfg5; do
    for lib in libforgex_amd_prev.so libforgex_amd.so; do
      FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config $cfg > $OUT/${cfg}_${lib}_$rep.json 2> $OUT/${cfg}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/${cfg}_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$cfg $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
    done
  done
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "pytest rc $?"
