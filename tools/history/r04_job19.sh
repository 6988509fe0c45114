# half-row staging for chain (bit 1) and nibble (bit 2) tables on 256-byte rows: parity + A/B in one allocation
export FXAMD_HALF_SCH=7
bash tools/r04_job.sh r04_c19 'k:chain_scheme or fuzz_patterns_and_row_lengths or real_reference_fixture or alternating' shape:chain_cfg3 shape:nibble_cfg3 shape:chain17_cfg3
export FXAMD_HALF_SCH=1
bash tools/r04_job.sh r04_c19b shape:chain_cfg3 shape:nibble_cfg3 shape:chain17_cfg3
export FXAMD_HALF_SCH=7
bash tools/r04_job.sh r04_c19c shape:chain_cfg3 shape:nibble_cfg3 shape:chain17_cfg3
