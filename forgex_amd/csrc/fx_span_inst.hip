// Explicit instantiation of the span-kernel launchers (fx_span.hpp): rows of 128 / 64 bytes, 8-state v_perm and 16-state nibble tables,
// with the gated follow-up (programs whose tables decode UTF-8) or the general row procedure inside the launch (GEN).
#include "fx_span.hpp"

#define FX_S(RL, SCH)                                              \
   template hipError_t launch_span<RL, SCH, false> FX_SPAN_SIG;    \
   template hipError_t launch_span<RL, SCH, true> FX_SPAN_SIG;
FX_S(128, 0)
FX_S(64, 0)
#undef FX_S
