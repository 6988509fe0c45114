#!/bin/bash
# round 3, GPU call 24: the backward pass of 192-byte rows with three lookup buffers (FX_BACK_PIPE3; libforgex_amd_bp3.so) against the library of call 23
# (libforgex_amd_p3.so: three-buffer forward walk + direct entry) and the previous commit; parity tests on the new library first
OUT=gpurun_out/r03_c24
mkdir -p $OUT
FXAMD_LIB=$(pwd)/forgex_amd/libforgex_amd_bp3.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exception_queues or few_exception or utf8 or config_rows or config_scale or many_patterns or packed or fuzz" > $OUT/tests_bp3.log 2>&1
echo "tests on bp3: rc $?"; tail -3 $OUT/tests_bp3.log
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for lib in libforgex_amd_prev.so libforgex_amd_p3.so libforgex_amd_bp3.so; do
    FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg4 > $OUT/cfg4_${lib}_$rep.json 2> $OUT/cfg4_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/cfg4_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg4 $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
    for s in utf8_192_clean utf8_192_flags match_utf8; do
      FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape $s --steps 60 --warmup 20 > $OUT/${s}_${lib}_$rep.json 2> $OUT/${s}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/${s}_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$s $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3))"
    done
  done
done
