#!/bin/bash
# round 3, GPU call 27: the aligned forward walk with its steady-state trips specialised (no "not yet joined" selects once every walking lane has joined;
# libforgex_amd_j.so: tile_12 only) against the committed library; parity tests on the new library first
OUT=gpurun_out/r03_c27
mkdir -p $OUT
FXAMD_LIB=$(pwd)/forgex_amd/libforgex_amd_j.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exception_queues or few_exception or utf8 or config_rows or config_scale or packed or fuzz" > $OUT/tests_j.log 2>&1
echo "tests on j: rc $?"; tail -3 $OUT/tests_j.log
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for lib in libforgex_amd.so libforgex_amd_j.so; do
    FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg4 > $OUT/cfg4_${lib}_$rep.json 2> $OUT/cfg4_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/cfg4_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg4 $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
    FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape utf8_192_clean --steps 60 --warmup 20 > $OUT/clean_${lib}_$rep.json 2> $OUT/clean_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/clean_${lib}_$rep.json').read().strip().splitlines()[-1]); print('utf8_192_clean $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3))"
  done
done
