#!/bin/bash
# round 3, GPU call 21: start-up overlap (table entries read first, the first tile's loads issued before the tables are staged) and the aligned forward
# loop without register copies, against the previous commit's library (forgex_amd/libforgex_amd_prev.so), interleaved repetitions; then the GPU suite
# and a soak of the fuzz test (more seeds / patterns than the suite's default)
OUT=gpurun_out/r03_c21
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for cfg in cfg2 cfg4 cfg5 cfg3; do
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
for seed in 1 2 3; do
  FX_FUZZ_SEED=$seed FX_FUZZ_PATTERNS=200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast_kernel_fuzz_patterns_and_row_lengths" 2>&1 | tail -2
  echo "soak seed $seed rc $?"
done
