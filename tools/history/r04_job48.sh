# 128-byte rows: the chain tables on 64-byte half rows (shipped rule) against the one-launch kernel (FXAMD_HALF_SCH=15); the nibble tables the same way (experiment, bit 5)
bash tools/r04_job.sh r04_c48 'k:half_rows or chain_scheme or fuzz_patterns_and_row_lengths or graph' shape:chain17_128 shape:nibble_128
FXAMD_HALF_SCH=15 bash tools/r04_job.sh r04_c48b shape:chain17_128
FXAMD_HALF_SCH=63 bash tools/r04_job.sh r04_c48c shape:nibble_128
bash tools/r04_job.sh r04_c48d shape:chain17_128 shape:nibble_128
FXAMD_HALF_SCH=63 bash tools/r04_job.sh r04_c48e shape:nibble_128
