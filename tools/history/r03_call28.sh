#!/bin/bash
# round 3, GPU call 28: `.match.` on the 8-state tables with three lookup buffers at 192 / 256-byte rows (FX_MATCH_PIPE3) + the specialised steady-state
# trips of the aligned walk (libforgex_amd.so) against the committed library (libforgex_amd_prev.so); then the GPU suite on the new library
OUT=gpurun_out/r03_c28
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 30"
for rep in 1 2 3; do
  for lib in libforgex_amd_prev.so libforgex_amd.so; do
    for s in match_cfg3 match_utf8; do
      FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape $s --steps 60 --warmup 20 > $OUT/${s}_${lib}_$rep.json 2> $OUT/${s}_${lib}_$rep.err
      python3 -c "
import json
d=json.loads(open('$OUT/${s}_${lib}_$rep.json').read().strip().splitlines()[-1]); print('$s $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])"
    done
    FXAMD_LIB=$(pwd)/forgex_amd/$lib $B --config cfg4 > $OUT/cfg4_${lib}_$rep.json 2> $OUT/cfg4_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/cfg4_${lib}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg4 $lib rep$rep', 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r['kernel_ms']*1e3,2), 'frac', round(r['frac'],4))"
  done
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "pytest rc $?"
