#!/bin/bash
# A/B of library variants over several configs in ONE gpurun call: tools/exp_variants_cfg.sh "cfg2 cfg4 cfg5" libA.so libB.so ...
# per run: step µs (as given), settled step µs, kernel µs (HIP events), whole-batch parity mismatches
CFGS=$1; shift
for rep in 1 2; do
for cfg in $CFGS; do
for lib in "$@"; do
  echo -n "== $cfg $lib: "
  FXAMD_LIB=$PWD/forgex_amd/$lib python bench.py --config $cfg --steps ${FX_AB_STEPS:-200} --warmup ${FX_AB_WARMUP:-30} --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"]*1e3,2), round(d[\"settled\"][\"ms_per_step\"]*1e3,2), round(d[\"roofline\"][\"kernel_ms\"]*1e3,2), d[\"parity\"][\"mismatches\"])"
done
done
done
