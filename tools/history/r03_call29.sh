#!/bin/bash
# round 3, GPU call 29: three lookup buffers for the BACKWARD pass on the 8-state tables at two waves per SIMD (fx_search_one CH 12 / 16; fx_search_fast CH 16:
# flags-only whole rows, rows longer than 256 bytes) against the committed library; config 3 also on the one-launch kernel (FXAMD_NO_HALF=1); then the GPU suite
OUT=gpurun_out/r03_c29
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-parity --steps 100 --warmup 30"
for rep in 1 2; do
  for lib in libforgex_amd_prev.so libforgex_amd.so; do
    for nh in 0 1; do
      if [ $nh = 1 ]; then export FXAMD_NO_HALF=1; else unset FXAMD_NO_HALF; fi
      FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg3 > $OUT/cfg3_nh${nh}_${lib}_$rep.json 2> $OUT/cfg3_nh${nh}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/cfg3_nh${nh}_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg3 no_half=$nh $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4), 'flags_only_us', round(d['flags_only']['ms_per_step']*1e3,2), r['kernel'][:40])"
    done
    unset FXAMD_NO_HALF
    for s in long_1024 long_4096 match_cfg3 nibble_cfg3; do
      FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape $s --steps 40 --warmup 15 > $OUT/${s}_${lib}_$rep.json 2> $OUT/${s}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/${s}_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$s $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])"
    done
  done
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "pytest rc $?"
