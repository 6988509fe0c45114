// Explicit instantiation of the tile-kernel launchers (and with them the kernels) for ONE chunk count: compiled per
// -DFX_INST_CH=<1|2|4|8|12|16> and, so that the few hundred variants build in parallel on all cores, per -DFX_INST_PART=<1|2|3>:
// 1 = the multi-pass kernels (fx_search_fast / fx_match_fast), 2 = the one-launch kernel for programs whose tables decode, 3 = the
// one-launch kernel with the general row procedure for queued rows (GEN), the many-pattern kernel and the gated follow-up
// (forgex_amd/csrc/Makefile).
#include "fx_multi.hpp"

#if FX_INST_PART == 1
#define FX_X(CH, M, S)                                           \
   template hipError_t launch_fast<CH, M, S> FX_TILE_SIG_FAST;   \
   template hipError_t launch_match<CH, M, S> FX_TILE_SIG_MATCH;
FX_TILE_COMBOS(FX_X, FX_INST_CH)
#undef FX_X
#endif

#define FX_Y(CH, S, B, G) template hipError_t launch_one<CH, S, B, G> FX_ONE_SIG;
#if FX_INST_PART == 2
FX_ONE_COMBOS_G(FX_Y, FX_INST_CH, false)
#endif
#if FX_INST_PART == 3
FX_ONE_COMBOS_G(FX_Y, FX_INST_CH, true)
#if FX_INST_CH <= 8   // (the many-pattern pass takes rows of up to 128 bytes: beyond, one pipeline per pattern is faster -- fxamd.hip)
template hipError_t launch_multi<FX_INST_CH> FX_MULTI_SIG;
#endif
#if FX_INST_CH == 16
template hipError_t launch_one_marked<FX_INST_CH, 0, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 1, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 2, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 3, false> FX_ONE_MARKED_SIG;
#endif
#if FX_INST_CH == 8 || FX_INST_CH == 4 || FX_INST_CH == 2 || FX_INST_CH == 1   // (the span kernel's follow-ups)
template hipError_t launch_one_marked<FX_INST_CH, 0, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 1, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 2, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 3, false> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 0, true> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 1, true> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 2, true> FX_ONE_MARKED_SIG;
#endif
#endif
#undef FX_Y

#ifdef FX_STAMP_ONE
// debug builds only (`make stamp-one`): read and clear the per-wave phase sums of this object's fx_search_one kernels (FX_STAMP_MAX_WAVES x FX_STAMP_SLOTS words)
extern "C" __attribute__((visibility("default"))) int fxamd_debug_stamps_one(unsigned long long* out, long long max_words) {
   const size_t bytes = sizeof(unsigned long long) * (size_t)(max_words < (long long)(FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS) ? max_words : FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS);
   if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fx_one_stamp_buf), bytes) != hipSuccess) return 1;
   void* p = nullptr;
   if (hipGetSymbolAddress(&p, HIP_SYMBOL(fx_one_stamp_buf)) != hipSuccess) return 1;
   if (hipMemset(p, 0, sizeof(unsigned long long) * FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS) != hipSuccess) return 1;
   return 0;
}
#endif

#if defined(FX_STAMP) && FX_INST_PART == 1
// debug builds only (`make stamp-fast`): read and clear the per-wave phase sums of this object's fx_search_fast kernels
extern "C" __attribute__((visibility("default"))) int fxamd_debug_stamps_fast(unsigned long long* out, long long max_words) {
   const size_t bytes = sizeof(unsigned long long) * (size_t)(max_words < (long long)(FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS) ? max_words : FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS);
   if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fx_stamp_buf), bytes) != hipSuccess) return 1;
   void* p = nullptr;
   if (hipGetSymbolAddress(&p, HIP_SYMBOL(fx_stamp_buf)) != hipSuccess) return 1;
   if (hipMemset(p, 0, sizeof(unsigned long long) * FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS) != hipSuccess) return 1;
   return 0;
}
#endif
