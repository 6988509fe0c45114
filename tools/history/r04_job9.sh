OUT=gpurun_out/r04_c9; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "match_operator or fuzz_patterns_and_row_lengths or match_one_launch or golden_vectors" > $OUT/k_match.log 2>&1; echo "match tests rc $?"; tail -2 $OUT/k_match.log
for sh in match_ragged_200 match_cfg3 match_cfg5; do bash tools/r04_job.sh r04_c9 shape:$sh; done
FXAMD_NO_SPEC=1 true
