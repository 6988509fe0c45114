/* forgex_amd_bench.h -- measurement hooks of libforgex_amd.so.  NOT part of the drop-in boundary (include/forgex_amd.h): used by
 * bench.py's roofline leg and the tools/ experiment scripts only. */
#ifndef FORGEX_AMD_BENCH_H
#define FORGEX_AMD_BENCH_H
#include "forgex_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Enqueue ONLY the dominant fast kernel, without the fix-up pass, so
 * its launch duration can be bracketed with HIP events.  FXAMD_E_ARG when the fast path does not apply. */
int fxamd_launch_fast_only(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags,
                           int32_t* d_from, int32_t* d_to, void* hip_stream);

/* The FXAMD_* test / experiment hooks (listed at fxamd_last_path in forgex_amd.h) are read from the environment ONCE per process;
 * a test that changes one afterwards calls this to have them read again.  Not synchronised with match calls in flight. */
void fxamd_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif
