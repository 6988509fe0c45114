#!/bin/bash
# round 3, GPU call 30: the ROLLED three-buffer backward pass on the 8-state tables (three chunks per trip; rows of 96 bytes and longer) against the committed
# library: config 5's shard (CH 8), config 3 on the one-launch kernel (FXAMD_NO_HALF=1, CH 16) and by default, config 2; parity tests first, full suite last
OUT=gpurun_out/r03_c30
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config_rows or config_scale or fuzz or exception_queues or few_exception" > $OUT/tests_new.log 2>&1
echo "tests on the new library: rc $?"; tail -3 $OUT/tests_new.log
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for lib in libforgex_amd_prev.so libforgex_amd.so; do
    for cfg in cfg5 cfg3 cfg2 cfg4; do
      FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config $cfg > $OUT/${cfg}_${lib}_$rep.json 2> $OUT/${cfg}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/${cfg}_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$cfg $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
    done
    FXAMD_NO_HALF=1 FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg3 > $OUT/cfg3nh_${lib}_$rep.json 2> $OUT/cfg3nh_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/cfg3nh_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg3 no_half=1 $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
  done
  FXAMD_LIB=$(pwd)/forgex_amd/libforgex_amd_hp.so $B --config cfg3 > $OUT/cfg3_hp_$rep.json 2> $OUT/cfg3_hp_$rep.err
  python3 -c "
import json
d=json.loads(open('$OUT/cfg3_hp_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg3 libforgex_amd_hp.so (half-row kernel with the rolled three-buffer backward pass) rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
done
FXAMD_LIB=$(pwd)/forgex_amd/libforgex_amd_hp.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_cfg3 or config_scale or config_rows" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "pytest rc $?"
