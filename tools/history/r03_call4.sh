#!/bin/bash
# round 3, GPU call 4: `.match.` in one launch (A/B against FXAMD_MULTIPASS=1 in the same build), aligned forward loop (A/B against the
# build without it), GPU tests
OUT=gpurun_out/r03_c4
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2; do
for cfg in cfg4 cfg5 cfg2; do
  for lib in libforgex_amd.so libforgex_amd_noalign.so; do
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
for sh in match_cfg3 match_cfg1x match_utf8; do
  for mp in 0 1; do
    if [ $mp = 1 ]; then export FXAMD_MULTIPASS=1; else unset FXAMD_MULTIPASS; fi
    python tools/bench_shapes.py --shape $sh > $OUT/shape_${sh}_mp$mp.json 2> $OUT/shape_${sh}_mp$mp.err || echo "shape $sh failed"
    python3 -c "
import json
d=json.loads(open('$OUT/shape_${sh}_mp$mp.json').read().strip().splitlines()[-1])
print('$sh multipass=$mp', 'ms', round(d['ms_per_step'],4), 'input_GBs', round(d['input_gbs']), 'frac', round(d['frac_of_hbm_peak'],3), 'path', d['last_path'], 'matches', d['matches'])" 2>/dev/null || tail -2 $OUT/shape_${sh}_mp$mp.err
  done
done
unset FXAMD_MULTIPASS
