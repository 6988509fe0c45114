#!/bin/bash
# round 3, GPU call 22: the few-rows exception scan (fx_few.hpp; libforgex_amd_few.so = the previous call's library + that) and the clustered-start
# shortcut into the aligned forward walk (libforgex_amd.so = both) against the previous commit's library, interleaved repetitions; then the GPU suite
OUT=gpurun_out/r03_c22
mkdir -p $OUT
for lib in libforgex_amd_prev.so libforgex_amd.so; do
  FXAMD_LIB=$(pwd)/forgex_amd/$lib python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "few_exception_rows or exception_queues" > $OUT/few_test_$lib.log 2>&1
  echo "few-rows tests with $lib: rc $?"; tail -3 $OUT/few_test_$lib.log
done
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for lib in libforgex_amd_prev.so libforgex_amd_few.so libforgex_amd.so; do
    FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg4 > $OUT/cfg4_${lib}_$rep.json 2> $OUT/cfg4_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/cfg4_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg4 $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
    for s in utf8_192_clean utf8_128 nibble_cfg3 match_utf8; do
      FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape $s --steps 60 --warmup 20 > $OUT/${s}_${lib}_$rep.json 2> $OUT/${s}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/${s}_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$s $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])"
    done
  done
  for cfg in cfg2 cfg5 cfg3; do
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
