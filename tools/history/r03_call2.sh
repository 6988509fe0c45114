#!/bin/bash
# round 3, GPU call 2: match compaction in the half-row kernel (A/B against the build without it, one allocation), bench init order,
# the config-scale fixture test, two RCCL ranks on one GPU
OUT=gpurun_out/r03_c2
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-parity --no-extras"
for rep in 1 2; do
  $B --steps 20 --warmup 5 > $OUT/drv_defer_$rep.json 2> $OUT/drv_defer_$rep.err
  FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_nodefer.so $B --steps 20 --warmup 5 > $OUT/drv_nodefer_$rep.json 2> $OUT/drv_nodefer_$rep.err
  $B --steps 100 --warmup 30 > $OUT/long_defer_$rep.json 2> $OUT/long_defer_$rep.err
  FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_nodefer.so $B --steps 100 --warmup 30 > $OUT/long_nodefer_$rep.json 2> $OUT/long_nodefer_$rep.err
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
timeout 200 python tools/rccl_two_on_one.py > $OUT/rccl_two_on_one.json 2> $OUT/rccl_two_on_one.err
for f in drv_defer_1 drv_nodefer_1 drv_defer_2 drv_nodefer_2 long_defer_1 long_nodefer_1 long_defer_2 long_nodefer_2 bench_full; do python3 - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", "value", round(d["value"]), "step_ms", round(d["ms_per_step"],4), "settled", round(d["settled"]["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), "cold", round(d["roofline"]["cold_kernel_ms"],4), "parity", (d.get("parity") or {}).get("mismatches"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
cat $OUT/rccl_two_on_one.json
