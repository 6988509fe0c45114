// fx_scan_few_rows: the scan of a gathered tile that holds only a FEW rows, with the lanes of the wave spread over the rows' CELLS.
//
// Why: a wave of fx_search_one ends with one pass over the exception rows it queued (structurally invalid UTF-8: gathered, decoded in
// LDS, scanned with the class-level tables).  That is usually a handful of rows -- config 4: five -- but fx_scan_tile costs the same for
// 5 live lanes as for 64: one lane per row, 2 * 16 * CH dependent steps.  Measured on config 4: 12.6 us of 103.7 (the same batch without
// its 1 % corrupted rows: 91.0 us, profiles/r03_utf8_shapes.txt), most of it this scan.
//
// Here CH lanes share a row, one 16-symbol cell each (64 / CH rows per call), and the two dependent passes become the classic
// chunk-parallel automaton walk on the 8-state v_perm tables (a transition FUNCTION of such an automaton is 8 bytes; composing two of
// them is two v_perm_b32):
//   backward: every lane composes the map of its cell (right to left), a segmented suffix scan over the row's lanes gives every cell
//             the state that enters it, a second walk of the cell from that state finds the cell's leftmost hit, a suffix-min gives
//             the row's;
//   forward:  the same left to right from the row's start (the start cell walks from the start symbol only and publishes a CONSTANT
//             map: everything to its right is then determined), a second walk records each cell's last accept, a max gives the row's.
// About 450 vector instructions per call whatever CH, against 190 * CH for the lane-per-row scan.
//
// Semantics: exactly fx_scan_tile<CH, SPANS, false, 0, false, /*DECODED*/ true, ...> on rows 0 .. take-1 of the tile (symbol ids, pads
// are the inert symbol, the virtual end of row = the trailing NUL then KILL symbols), results through `emit` as for a gathered row
// (ordered = false).  Reference: api_internal_m.F90:108-164 (do_matching_including) as restated in fx_scan_tile.
#pragma once
#include "fx_tile.hpp"

struct FxMap8 {
   uint32_t lo, hi;   // byte q = image of state q (q = 0..3 in lo, 4..7 in hi)
};
__device__ __forceinline__ FxMap8 fx_map_id() { return FxMap8{0x03020100u, 0x07060504u}; }
// first `first`, then `then`
__device__ __forceinline__ FxMap8 fx_map_compose(const FxMap8 then, const FxMap8 first) {
   return FxMap8{__builtin_amdgcn_perm(then.hi, then.lo, first.lo), __builtin_amdgcn_perm(then.hi, then.lo, first.hi)};
}
__device__ __forceinline__ FxMap8 fx_map_after(const uint2 f, const FxMap8 first) {   // the symbol's table row applied after `first`
   return FxMap8{__builtin_amdgcn_perm(f.y, f.x, first.lo), __builtin_amdgcn_perm(f.y, f.x, first.hi)};
}
// a state as the kernels hold it (its id in all four bytes) through a map
__device__ __forceinline__ uint32_t fx_map_apply(const FxMap8 m, const uint32_t st) { return __builtin_amdgcn_perm(m.hi, m.lo, st); }
__device__ __forceinline__ uint32_t fx_lane_read(const uint32_t v, const uint32_t src_lane) {
   return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

template <int CH>
constexpr uint32_t fx_few_rows_max() {
   return 64u / (uint32_t)CH;
}

template <int CH, bool SPANS, class Emit>
__device__ __forceinline__ void fx_scan_few_rows(const uint4* tile, const uint2* __restrict__ tabR, const uint2* __restrict__ tabA, const FastParams& P,
                                                 const uint32_t lane, const uint32_t take, const uint32_t* rowq, Emit& emit, const uint32_t r0 = 0u) {
   static_assert(CH >= 2 && CH <= 16, "cells of a row share a wave");
   constexpr uint32_t L = 16u * CH;
   // rows r0 .. r0 + 64 / CH - 1 of the gathered tile (round 6: a tile of up to three such groups is scanned group by group)
   const uint32_t r = r0 + lane / (uint32_t)CH, k = lane % (uint32_t)CH;
   const bool on = r < take && lane < fx_few_rows_max<CH>() * (uint32_t)CH;
   const uint4 cell = tile[tile_cell(on ? r : 0u, on ? k : 0u)];
   uint2 f[16];
   // ---- backward: the reverse unanchored automaton; the LAST hit seen is the leftmost start ----
   lookup8(&f[0], cell.x, cell.y, tabR);
   lookup8(&f[8], cell.z, cell.w, tabR);
   FxMap8 T = fx_map_id();
#pragma unroll
   for (int i = 15; i >= 0; --i) T = fx_map_after(f[i], T);
   // inclusive suffix scan over the row's lanes: T_k = (cell k) after (cell k+1) after ... after (cell CH-1)
#pragma unroll
   for (uint32_t d = 1; d < (uint32_t)CH; d <<= 1) {
      const FxMap8 q{fx_lane_read(T.lo, lane + d), fx_lane_read(T.hi, lane + d)};
      const FxMap8 c = fx_map_compose(T, q);
      const bool take_it = k + d < (uint32_t)CH;
      T.lo = take_it ? c.lo : T.lo;
      T.hi = take_it ? c.hi : T.hi;
   }
   uint32_t s;
   {
      const FxMap8 right{fx_lane_read(T.lo, lane + 1u), fx_lane_read(T.hi, lane + 1u)};
      uint32_t st = k == (uint32_t)CH - 1u ? P.R_start : fx_map_apply(right, P.R_start);
      uint32_t loc = 16u;
#pragma unroll
      for (int i = 15; i >= 0; --i) {
         st = fxstep(f[i], st, nullptr);
         loc = st >= P.hit_min ? (uint32_t)i : loc;
      }
      uint32_t cand = loc < 16u ? 16u * k + loc : 0xFFFFu;   // text index of this cell's leftmost hit
      // leading NUL (lane k == 0: `st` is the state after the row's first symbol): a hit there is the leftmost start
      const uint2 fz = tabR[0];
      const bool s_nul = fxstep(fz, st, nullptr) >= P.hit_min;
#pragma unroll
      for (uint32_t d = 1; d < (uint32_t)CH; d <<= 1) {
         const uint32_t o = fx_lane_read(cand, lane + d);
         cand = (k + d < (uint32_t)CH && o < cand) ? o : cand;
      }
      s = cand != 0xFFFFu ? cand + 2u : 0u;   // wrapped start index (1 = leading NUL, j + 2 for text index j), 0 = none
      s = s_nul ? 1u : s;
      s = fx_lane_read(s, lane - k);          // (lane k == 0 holds the row's)
   }
   // ---- forward from the leftmost start: anchored automaton, longest accept ----
   uint32_t cur0 = (s != 0u && (SPANS || s == 1u) && P.lit_len == 0u) ? P.A_init : 0u;
   uint32_t mm = (P.lit_len != 0u && s != 0u) ? s + P.lit_len : 0u;   // max_match (wrapped index of the symbol after the match)
   const uint32_t j = s >= 2u ? s - 2u : 0u;
   if (s == 1u) {
      const uint2 fz = tabA[0];
      cur0 = fxstep(fz, cur0, nullptr);
      mm = cur0 >= P.acc_min ? 2u : 0u;
   }
   if (__builtin_amdgcn_ballot_w64(on && cur0 != 0u) != 0) {
      const uint32_t ks = j >> 4, o = j & 15u;
      lookup8(&f[0], cell.x, cell.y, tabA);
      lookup8(&f[8], cell.z, cell.w, tabA);
      const bool start_cell = k == ks;
      FxMap8 M = fx_map_id();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
         const FxMap8 nx = fx_map_after(f[i], M);
         const bool skip = start_cell && (uint32_t)i < o;   // (the start cell walks from the start symbol)
         M.lo = skip ? M.lo : nx.lo;
         M.hi = skip ? M.hi : nx.hi;
      }
      if (start_cell) {   // what leaves the start cell is ONE state: a constant map
         const uint32_t e = fx_map_apply(M, cur0);
         M.lo = e;
         M.hi = e;
      }
      if (k < ks) M = fx_map_id();
      // inclusive prefix scan: M_k = (cell k) after (cell k-1) after ... after (cell 0)
#pragma unroll
      for (uint32_t d = 1; d < (uint32_t)CH; d <<= 1) {
         const FxMap8 q{fx_lane_read(M.lo, lane - d), fx_lane_read(M.hi, lane - d)};
         const FxMap8 c = fx_map_compose(M, q);
         const bool take_it = k >= d;
         M.lo = take_it ? c.lo : M.lo;
         M.hi = take_it ? c.hi : M.hi;
      }
      const FxMap8 left{fx_lane_read(M.lo, lane - 1u), fx_lane_read(M.hi, lane - 1u)};
      // (cells right of the start: their left neighbour's map is constant -- its image of any state is the entering state)
      uint32_t st = start_cell ? cur0 : (k > ks ? fx_map_apply(left, 0u) : 0u);
      uint32_t mmc = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
         const uint32_t nx = fxstep(f[i], st, nullptr);
         const bool walk = !(start_cell && (uint32_t)i < o);
         st = walk ? nx : st;
         mmc = (walk && nx >= P.acc_min) ? 16u * k + (uint32_t)i + 3u : mmc;
      }
      {   // what follows the text: the trailing NUL (an accept after it gives max_match = L + 3), then KILL symbols
         const uint2 fz = tabA[0];
         const uint32_t nx = fxstep(fz, st, nullptr);
         mmc = (k == (uint32_t)CH - 1u && nx >= P.acc_min) ? L + 3u : mmc;
      }
#pragma unroll
      for (uint32_t d = 1; d < (uint32_t)CH; d <<= 1) {
         const uint32_t v = fx_lane_read(mmc, lane + d);
         mmc = (k + d < (uint32_t)CH && v > mmc) ? v : mmc;
      }
      mm = mmc > mm ? mmc : mm;   // (lane k == 0 holds the row's; later accepts have larger indices than the leading NUL's 2)
   }
   uint32_t flag = 0;
   int32_t fr = 0, tt = 0;
   if (SPANS) {
      if (s != 0u && mm != 0u) {   // api_internal_m.F90:140-148
         fr = (int32_t)(s - 1u);
         if (fr == 0) fr = 1;
         tt = mm >= L + 2u ? (int32_t)L : (int32_t)mm - 2;
         if (fr > 0 && tt > 0) flag = 1;
         else {
            fr = 0;
            tt = 0;
         }
      }
   } else {
      flag = (s >= 2u || (s == 1u && mm > 2u)) ? 1u : 0u;
   }
   const bool mine = on && k == 0u;
   const int64_t row = (int64_t)(mine ? rowq[r] : 0u);
   emit(row, mine, false, flag, fr, tt, true);
}
