# A/B of library variants in ONE gpurun call (box-to-box variance is several percent): tools/exp_variants.sh libA.so libB.so ...
for rep in 1 2; do
for lib in "$@"; do
  for extra in "" "--flags-only"; do
    echo -n "== $lib $extra: "
    FXAMD_LIB=$PWD/forgex_amd/$lib python bench.py --steps ${FX_AB_STEPS:-200} --warmup ${FX_AB_WARMUP:-30} --no-cpu-baseline $extra 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"kernel_ms\"],4), round(d[\"roofline\"][\"frac\"],4))"
  done
done
done
