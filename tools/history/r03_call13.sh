#!/bin/bash
# round 3, GPU call 13: the half-row kernel tuned for four waves per SIMD (FX_HALF4: libforgex_amd_half4.so) against three (libforgex_amd.so)
OUT=gpurun_out/r03_c13
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-extras"
show() { python3 - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2", "value", round(d["value"]), "step_ms", round(d["ms_per_step"],4), "settled", round(d["settled"]["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), "cold", round(d["roofline"]["cold_kernel_ms"],4), "frac", round(d["roofline"]["frac"],3), "parity", (d.get("parity") or {}).get("mismatches"))
except Exception as e:
    print("$2", "FAILED", e)
PY
}
for rep in 1 2 3; do
  for lib in libforgex_amd.so libforgex_amd_half4.so; do
    FXAMD_LIB=$PWD/forgex_amd/$lib $B --steps 100 --warmup 30 > $OUT/long_${lib}_$rep.json 2> $OUT/long_${lib}_$rep.err; show $OUT/long_${lib}_$rep.json "long $lib rep$rep"
    FXAMD_LIB=$PWD/forgex_amd/$lib $B --steps 20 --warmup 5 > $OUT/drv_${lib}_$rep.json 2> $OUT/drv_${lib}_$rep.err; show $OUT/drv_${lib}_$rep.json "drv $lib rep$rep"
  done
done
FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_half4.so python -m pytest tests/test_gpu_parity.py -x -q -k "config_scale or full_size or golden or fast_and_general or quirk or non_ascii or utf8" > $OUT/pytest_half4.log 2>&1; echo "pytest half4 rc $?"; tail -2 $OUT/pytest_half4.log
