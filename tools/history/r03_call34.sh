#!/bin/bash
# round 3, GPU call 34: soak of the FINAL build (fuzz test of the tile kernels incl. `.match.` at every row length, few-exception-rows test), seeds 21, 22
OUT=gpurun_out/r03_c34
mkdir -p $OUT
for seed in 21 22; do
  FX_FUZZ_SEED=$seed FX_FUZZ_PATTERNS=150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast_kernel_fuzz_patterns_and_row_lengths or few_exception_rows" > $OUT/soak_$seed.log 2>&1
  echo "soak seed $seed rc $?"; tail -2 $OUT/soak_$seed.log
done
