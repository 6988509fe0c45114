# config 4's pattern and text in rows of 256 bytes: the half-row pipeline (its first pass defers every tile) against the one-launch kernel (FXAMD_NO_HALF=1)
for rep in 1 2; do
  bash tools/r04_job.sh r04_c38 shape:utf8_256
  FXAMD_NO_HALF=1 bash tools/r04_job.sh r04_c38n shape:utf8_256
done
