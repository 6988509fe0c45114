OUT=gpurun_out/r04_c12; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "tiny_rows or golden_vectors or hip_graph or alternating" > $OUT/k_tiny.log 2>&1; echo "tiny tests rc $?"; tail -12 $OUT/k_tiny.log
for rep in 1 2; do bash tools/r04_job.sh r04_c12 shape:in_flags_cfg1x; FXAMD_NO_TINY=1 bash tools/r04_job.sh r04_c12 shape:in_flags_cfg1x; done
