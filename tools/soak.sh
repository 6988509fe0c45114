#!/bin/bash
# Soak of the fuzz tests under other seeds (GPU box):  bash tools/soak.sh <tag> "<seed> <seed> ..." [patterns per test]
TAG=${1:-soak}; SEEDS=${2:-"1 2 3"}; NP=${3:-120}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for s in $SEEDS; do
  FX_FUZZ_SEED=$s FX_FUZZ_PATTERNS=$NP FX_FUZZ_GROUPS=40 timeout 3000 python -m pytest tests -m gpu -x -q \
     -k "fuzz_patterns_and_row_lengths or many_patterns_fuzz_groups or few_exception_rows or fuzzed_patterns_through_gpu or speculative or tiny_rows_fuzz or span_kernel_fuzz" > $OUT/seed_$s.log 2>&1
  echo "seed $s rc $? $(tail -1 $OUT/seed_$s.log)"
done
