# experiment: the 256-byte segment loads of rows longer than 256 bytes WITHOUT the nt cache policy (libforgex_amd_exp.so: tile_16_1.o built
# with -DFX_LONG_AUX=0) against the shipped build, interleaved in one allocation
for rep in 1 2; do
  bash tools/r04_job.sh r04_c22a shape:long_1024 shape:long_400 shape:long_4096 shape:match_long_1024
  cp forgex_amd/libforgex_amd.so /tmp/lib_keep.so
  cp forgex_amd/libforgex_amd_exp.so forgex_amd/libforgex_amd.so
  echo "--- FX_LONG_AUX=0"
  bash tools/r04_job.sh r04_c22b shape:long_1024 shape:long_400 shape:long_4096 shape:match_long_1024
  cp /tmp/lib_keep.so forgex_amd/libforgex_amd.so
  echo "--- shipped"
done
