// Explicit instantiation of the tile-kernel launchers (and with them the kernels) for ONE chunk count: compiled once per
// -DFX_INST_CH=<1|2|3|4|6|8|12|16> so that the variants build in parallel (forgex_amd/csrc/Makefile).
#include "fx_multi.hpp"

#define FX_X(CH, M, S)                                           \
   template hipError_t launch_fast<CH, M, S> FX_TILE_SIG_FAST;   \
   template hipError_t launch_match<CH, M, S> FX_TILE_SIG_MATCH;
FX_TILE_COMBOS(FX_X, FX_INST_CH)
#undef FX_X

#define FX_Y(CH, S, B, G) template hipError_t launch_one<CH, S, B, G> FX_ONE_SIG;
FX_ONE_COMBOS(FX_Y, FX_INST_CH)
#undef FX_Y

template hipError_t launch_multi<FX_INST_CH> FX_MULTI_SIG;

#if FX_INST_CH == 16
template hipError_t launch_one_marked<FX_INST_CH, 0> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 1> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 2> FX_ONE_MARKED_SIG;
template hipError_t launch_one_marked<FX_INST_CH, 3> FX_ONE_MARKED_SIG;
#endif
