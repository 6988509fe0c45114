# FX_ADAPT_CALLS: the half-row pipeline on a batch that is mostly UTF-8 (a general pattern over config 4's text in 256-byte rows), with and without
# the adaptive first pass; the headline shape must not notice
bash tools/r04_job.sh r04_c42 'k:adaptive or alternating or fixture or full_size' shape:utf8_256_any bench_cfg3
FXAMD_NO_ADAPT=1 bash tools/r04_job.sh r04_c42n shape:utf8_256_any bench_cfg3
bash tools/r04_job.sh r04_c42b shape:utf8_256_any bench_cfg3
FXAMD_NO_ADAPT=1 bash tools/r04_job.sh r04_c42m shape:utf8_256_any bench_cfg3
