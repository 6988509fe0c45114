// Explicit instantiation of the span-kernel launchers (fx_span.hpp): rows of up to 128 / 64 / 32 / 16 bytes on the 8-state v_perm tables
// (aligned and ragged, plain and packed results); aligned and ragged rows on the nibble tables (all four lengths, plain results).
#include "fx_span.hpp"

template hipError_t launch_span<128, 0> FX_SPAN_SIG;
template hipError_t launch_span<64, 0> FX_SPAN_SIG;
template hipError_t launch_span<32, 0> FX_SPAN_SIG;
template hipError_t launch_span<16, 0> FX_SPAN_SIG;
template hipError_t launch_span<128, 2> FX_SPAN_SIG;
template hipError_t launch_span<64, 2> FX_SPAN_SIG;
template hipError_t launch_span<32, 2> FX_SPAN_SIG;
template hipError_t launch_span<16, 2> FX_SPAN_SIG;
