# generalised tiny-row kernels: tests + shapes, with and without the kernels
bash tools/r04_job.sh r04_c14 'k:tiny' shape:match_rows_12 shape:in_flags_rows_20 shape:in_flags_rows_10 shape:match_cfg1x shape:in_flags_cfg1x
FXAMD_NO_TINY=1 bash tools/r04_job.sh r04_c14n shape:match_rows_12 shape:in_flags_rows_20 shape:in_flags_rows_10
