// Explicit instantiation of the span-kernel launchers (fx_span.hpp): rows of 128 / 64 / 32 / 16 bytes on the 8-state v_perm tables.
#include "fx_span.hpp"

template hipError_t launch_span<128, 0> FX_SPAN_SIG;
template hipError_t launch_span<64, 0> FX_SPAN_SIG;
template hipError_t launch_span<32, 0> FX_SPAN_SIG;
template hipError_t launch_span<16, 0> FX_SPAN_SIG;
