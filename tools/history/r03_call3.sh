#!/bin/bash
# round 3, GPU call 3: match compaction in fx_search_one (sparse tiles only) A/B on configs 2 / 4 / 5 / 3, GPU tests (misaligned bases on the
# tile kernels, FXAMD_FORCE_GENERAL), shapes of DESIGN 4.1d
OUT=gpurun_out/r03_c3
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2; do
for cfg in cfg2 cfg5 cfg4 cfg3; do
  for lib in libforgex_amd.so libforgex_amd_nodefer.so; do
    FXAMD_LIB=$PWD/forgex_amd/$lib python bench.py --config $cfg --steps 100 --warmup 30 --no-cpu-baseline > $OUT/${cfg}_${lib}_$rep.json 2> $OUT/${cfg}_${lib}_$rep.err
    python3 - <<PY
import json
try:
    d=json.loads(open("$OUT/${cfg}_${lib}_$rep.json").read().strip().splitlines()[-1])
    print("$cfg $lib rep$rep", "step_us", round(d["ms_per_step"]*1e3,2), "settled_us", round(d["settled"]["ms_per_step"]*1e3,2), "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "flags_only_us", round(d["flags_only"]["ms_per_step"]*1e3,2), "parity", d["parity"]["mismatches"], "path", d["roofline"]["kernel"])
except Exception as e:
    print("$cfg $lib", "FAILED", e)
PY
  done
done
done
for sh in match_cfg3 match_cfg1x match_utf8 long_1024 long_4096 long_400 nibble_cfg3 chain_cfg3 literal_cfg2 multi6_cfg3 packed_cfg5 packed_cfg3 ragged_255; do
  python tools/bench_shapes.py --shape $sh > $OUT/shape_$sh.json 2> $OUT/shape_$sh.err || echo "shape $sh failed"
  python3 -c "
import json
d=json.loads(open('$OUT/shape_$sh.json').read().strip().splitlines()[-1])
print('$sh', 'ms', round(d['ms_per_step'],4), 'input_GBs', round(d['input_gbs']), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])" 2>/dev/null || tail -2 $OUT/shape_$sh.err
done
