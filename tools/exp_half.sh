#!/bin/bash
# A/B inside ONE GPU allocation: memory path only (no-compute build, FX_EXP_NOCOMPUTE) with full-row and with half-row staging, then
# the real kernels.  Usage (through gpurun): bash tools/exp_half.sh <outdir>
OUT=${1:-gpurun_out/exp_half}
mkdir -p $OUT
B="python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-parity --no-extras"
FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_nc.so $B > $OUT/nc_full.json 2> $OUT/nc_full.err
FXAMD_HALF=1 FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_nc.so $B > $OUT/nc_half.json 2> $OUT/nc_half.err
$B > $OUT/real_full.json 2> $OUT/real_full.err
FXAMD_HALF=1 $B > $OUT/real_half.json 2> $OUT/real_half.err
for f in nc_full nc_half real_full real_half; do python3 - <<PY
import json
d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
print("$f", "step_ms", round(d["ms_per_step"],4), "settled", round(d["settled"]["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4))
PY
done
