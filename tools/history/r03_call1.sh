#!/bin/bash
# round 3, GPU call 1: tests after the ADVICE fixes, the cold-penalty attribution (copy / no-compute / real kernel), driver-protocol bench
OUT=gpurun_out/r03_c1
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
python tools/exp_transient.py > $OUT/transient_real.json 2> $OUT/transient_real.err
FXAMD_LIB=$PWD/forgex_amd/libforgex_amd_nc.so python tools/exp_transient.py > $OUT/transient_nc.json 2> $OUT/transient_nc.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $OUT/bench_driver2.json 2> $OUT/bench_driver2.err
tail -c 600 $OUT/bench_driver.json
