#!/bin/bash
# round 3, GPU call 26: soak of the final build -- the fuzz test of the tile kernels and the few-exception-rows test with more seeds / patterns than the
# suite's default, and the exception-queue test
OUT=gpurun_out/r03_c26
mkdir -p $OUT
for seed in 11 12 13; do
  FX_FUZZ_SEED=$seed FX_FUZZ_PATTERNS=150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast_kernel_fuzz_patterns_and_row_lengths or few_exception_rows" > $OUT/soak_$seed.log 2>&1
  echo "soak seed $seed rc $?"; tail -2 $OUT/soak_$seed.log
done
