bash tools/r04_job.sh r04_c28 'k:long_rows' shape:long_chain_1024 shape:long_1024
FXAMD_HALF_SCH=7 bash tools/r04_job.sh r04_c28b shape:long_chain_1024
bash tools/r04_job.sh r04_c28c shape:long_chain_1024
