#!/bin/bash
# round 3, GPU call 32: the three-buffer `.match.` pass at rows of 96 / 128 bytes too (libforgex_amd_m6.so: tile_6 / tile_8 built with -DFX_MATCH_P3_MINCH=6)
# against the final library; `.match.` over config 5's shard; parity tests of `.match.` on the variant first
OUT=gpurun_out/r03_c32
mkdir -p $OUT
FXAMD_LIB=$(pwd)/forgex_amd/libforgex_amd_m6.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "match or fuzz or golden" > $OUT/tests_m6.log 2>&1
echo "tests on m6: rc $?"; tail -3 $OUT/tests_m6.log
for rep in 1 2 3; do
  for lib in libforgex_amd.so libforgex_amd_m6.so; do
    FXAMD_LIB=$(pwd)/forgex_amd/$lib python tools/bench_shapes.py --shape match_cfg5 --steps 60 --warmup 20 > $OUT/match_cfg5_${lib}_$rep.json 2> $OUT/match_cfg5_${lib}_$rep.err
    python3 -c "
import json
d=json.loads(open('$OUT/match_cfg5_${lib}_$rep.json').read().strip().splitlines()[-1]); print('match_cfg5 $lib rep$rep us', round(d['ms_per_step']*1e3,2), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])"
  done
done
