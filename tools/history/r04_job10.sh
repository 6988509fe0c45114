OUT=gpurun_out/r04_c10; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "tiny_rows or golden_vectors or config_rows_vs_oracle or match_operator" > $OUT/k_tiny.log 2>&1; echo "tiny tests rc $?"; tail -12 $OUT/k_tiny.log
for rep in 1 2; do bash tools/r04_job.sh r04_c10 shape:match_cfg1x; FXAMD_NO_TINY=1 bash tools/r04_job.sh r04_c10 shape:match_cfg1x; done
bash tools/r04_job.sh r04_c10 dist1
