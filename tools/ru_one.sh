#!/bin/bash
# Resource usage (VGPRs, SGPR spills, occupancy) of ONE instantiation of fx_search_one, in seconds instead of a whole chunk-count object:
#   tools/ru_one.sh "8, true, 0, 3, false, false"        (template arguments CH, SPANS, SCH, BSCH, RAGGED, GEN[, MARKED[, MATCH]])
# Extra compiler flags in $EXTRA (e.g. EXTRA=-DFX_SPEC_FWD=0).
ARGS="$1"
TMP=$(mktemp /tmp/ru_one.XXXXXX.hip)
cat > $TMP <<EOT
#include "fx_one.hpp"
template __global__ void fx_search_one<$ARGS>(const uint8_t*, int64_t, const uint8_t*, FastParams, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t, uint32_t, uint32_t, const uint32_t*, const uint8_t*);
const FxEnv& fx_env() { static FxEnv e{}; return e; }
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $EXTRA -I$(dirname $0)/../forgex_amd/csrc --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $TMP -o /dev/null 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|SGPRs Spill|Occupancy|ScratchSize|LDS Size" | sed 's/.*remark: *//'
rm -f $TMP
