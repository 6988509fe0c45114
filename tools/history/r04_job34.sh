bash tools/r04_job.sh r04_c34 'k:long_rows_chain or graph or fuzz_patterns_and_row_lengths or chain_scheme or match_one_launch' shape:chain17_200
FXAMD_HALF_SCH=7 bash tools/r04_job.sh r04_c34b shape:chain17_200
bash tools/r04_job.sh r04_c34c shape:chain17_200
