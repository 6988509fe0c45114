#!/bin/bash
# round 3, GPU call 5: byte-level forward automaton in the v_perm format (FXP_F_BYTE_A8; A/B with FXAMD_NO_A8=1 in the same build),
# regression check of configs 2 / 5 / 3 against the round's first build, GPU tests, rocprofv3 of config 4
OUT=gpurun_out/r03_c5
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
show() { python3 - <<PY
import json
try:
    d=json.loads(open("$1").read().strip().splitlines()[-1])
    print("$2", "step_us", round(d["ms_per_step"]*1e3,2), "settled_us", round(d["settled"]["ms_per_step"]*1e3,2), "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "flags_only_us", round(d["flags_only"]["ms_per_step"]*1e3,2), "parity", d["parity"]["mismatches"], "path", d["roofline"]["kernel"])
except Exception as e:
    print("$2", "FAILED", e)
PY
}
for rep in 1 2; do
  python bench.py --config cfg4 --steps 100 --warmup 30 --no-cpu-baseline > $OUT/cfg4_a8_$rep.json 2> $OUT/cfg4_a8_$rep.err; show $OUT/cfg4_a8_$rep.json "cfg4 A8 rep$rep"
  FXAMD_NO_A8=1 python bench.py --config cfg4 --steps 100 --warmup 30 --no-cpu-baseline > $OUT/cfg4_noa8_$rep.json 2> $OUT/cfg4_noa8_$rep.err; show $OUT/cfg4_noa8_$rep.json "cfg4 NO_A8 rep$rep"
  for cfg in cfg5 cfg2 cfg3; do
    for lib in libforgex_amd.so libforgex_amd_nodefer.so; do
      FXAMD_LIB=$PWD/forgex_amd/$lib python bench.py --config $cfg --steps 100 --warmup 30 --no-cpu-baseline > $OUT/${cfg}_${lib}_$rep.json 2> $OUT/${cfg}_${lib}_$rep.err; show $OUT/${cfg}_${lib}_$rep.json "$cfg $lib rep$rep"
    done
  done
done
bash tools/profile_round.sh r03_cfg4 cfg4 > $OUT/prof_cfg4.log 2>&1; tail -6 $OUT/prof_cfg4.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err; show $OUT/bench_full.json "cfg3 driver protocol"
