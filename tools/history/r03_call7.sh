#!/bin/bash
# round 3, GPU call 7: follow-ups of the shared pass side by side (A/B with FXAMD_MULTI_SERIAL=1), soak runs of the pattern fuzzers on the
# round's kernels (other seeds than the suite's), bench lines of every config
OUT=gpurun_out/r03_c7
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
python tools/exp_multi.py cfg4 > $OUT/multi_cfg4_side.txt 2>&1; grep -v amdgpu.ids $OUT/multi_cfg4_side.txt
FXAMD_MULTI_SERIAL=1 python tools/exp_multi.py cfg4 > $OUT/multi_cfg4_serial.txt 2>&1; grep -v amdgpu.ids $OUT/multi_cfg4_serial.txt
FXAMD_NO_MULTI=1 python tools/exp_multi.py cfg4 > $OUT/multi_cfg4_sequential.txt 2>&1; grep -v amdgpu.ids $OUT/multi_cfg4_sequential.txt
for seed in 31 32 33 34; do
  FX_FUZZ_SEED=$seed FX_FUZZ_PATTERNS=160 FX_FUZZ_GROUPS=40 python -m pytest tests/test_gpu_parity.py -x -q -k "fuzz" > $OUT/soak_$seed.log 2>&1; echo "soak seed $seed rc $?"; tail -1 $OUT/soak_$seed.log
done
for cfg in cfg2 cfg3 cfg4 cfg5; do
  python bench.py --config $cfg --steps 200 --warmup 30 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", "step_us", round(d["ms_per_step"]*1e3,2), "settled_us", round(d["settled"]["ms_per_step"]*1e3,2), "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "frac", round(d["roofline"]["frac"],3), "flags_only_us", round(d["flags_only"]["ms_per_step"]*1e3,2), "parity", d["parity"]["mismatches"], "host", round(d["host_path"]["value"],1), d["host_path"].get("pinned"), "cpu", d["cpu_baseline"]["value"], (d["cpu_baseline"].get("gpu_vs_reference_on_sample") or {}).get("mismatches"))
PY
done
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -c 400 $OUT/bench_driver.json
FXAMD_BENCH_FORCE_DIST=1 python bench.py --config cfg5 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg5_rccl_world1.json 2> $OUT/bench_cfg5_rccl_world1.err; python3 -c "
import json
d=json.loads(open('$OUT/bench_cfg5_rccl_world1.json').read().strip().splitlines()[-1]); print('cfg5 world1 rccl', d['ms_per_step'], d['packed_step_ms'], d['gather_ms'])"
