bash tools/r04_job.sh r04_c31 'k:long_rows' shape:match_long_chain_1024 shape:match_long_1024
FXAMD_HALF_SCH=7 bash tools/r04_job.sh r04_c31b shape:match_long_chain_1024
bash tools/r04_job.sh r04_c31c shape:match_long_chain_1024
