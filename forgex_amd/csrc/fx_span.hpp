// fx_search_span: `.in.` / regex with spans over rows of 128 / 64 / 32 / 16 bytes -- a lane owns a 128-byte SPAN of K = 128 / RL whole rows (round 5).
//
// The headline kernel (fx_search_fast<8, ..., LONG>: 256-byte rows staged as two 128-byte halves, 8 KB of tile per wave, 127 VGPRs, four
// waves per SIMD) runs at 0.72 of the HBM peak; the one-launch kernel that took 128- and 64-byte rows (fx_search_one<8 / 4>) at 0.58 / 0.53:
// a 64-row tile of short rows pays the per-tile fixed work (staging, exact start, forward window, result stores) for fewer bytes, and its
// 163 VGPRs stop at three waves per SIMD.  Rows are independent (reference src/forgex.F90:74: the operators are elemental), so the SAME
// memory path and the same lean loops serve short rows when a lane's 128 contiguous bytes are read as K whole rows:
//   * a wave's tile = 64 spans = 8 KB of contiguous bytes = 64 K rows (whole 128-byte lines, coalesced 16-byte pieces, the swizzled store of
//     the other tile kernels); the next tile's loads are in flight while this one is scanned;
//   * every row is scanned on its own -- the reverse automaton from the row's last byte with a fresh start state (api_internal_m.F90:108-155:
//     the leftmost start with a non-empty match), then the leading NUL; NO state crosses a row boundary;
//   * FINISH (exact start by re-walking ONE 8-byte group, anchored automaton forwards from it: a 32-symbol window, then 8 symbols per
//     trip) is per-ROW work that only rows with a hit need, but a wave pays for it per pass of 64 lanes.  With K >= 2 the rows that need it
//     are COMPACTED: their (lane, row, hit group, entry state) go to a per-wave slot list in LDS, lane q finishes slot q reading that row's
//     bytes out of the tile (any lane can: the tile is in LDS), and the owner reads the result back from the slot -- ceil(hits / 64)
//     passes per tile instead of K (BASELINE config 2: one row in ten matches -- one pass per 256 rows instead of four; the one-launch
//     kernel's match compaction finished such rows from GLOBAL memory, 1.13 x the algorithmic HBM traffic);
//   * the lane keeps its K results and writes them with ONE store per array (K flag bytes, K x int32 from, K x int32 to: consecutive rows).
// Everything a row needs is in LDS while it is scanned (its bytes + one shared end-of-row cell: NUL, KILL x 15): HBM traffic = the rows once
// + the results.
// Bytes >= 0x80 (any program) and rows that end in the overlap state of a bordered prefix literal (candidate-list driver programs): the
// tile's rows are marked FX_NEEDS_GENERAL for ONE gated follow-up -- the one-launch kernel over marked tiles (fx_search_one<.., MARKED>:
// byte-level tables or the in-LDS decode, exception queues, the general row procedure for programs whose tables cannot decode) -- exactly
// as the half-row pipeline of 256-byte rows does (fxamd.hip, last_path 16 -> 18).
// (Sparse matches, measured and NOT built in: tiles with few rows that need the finish handing them to a per-wave queue ACROSS tiles, 64
//  gathered rows finished per pass from global memory as the one-launch kernel's match compaction does -- `\d{3}-\d{4}`-like nibble programs over
//  16-byte rows, 3 % of the rows matching: 0.563 -> 0.856 ms; BASELINE config 2: 19.7 us either way (gpurun call r05_c18).  The flush waits for
//  rows that have left every cache and scatters its results; one in-LDS finish pass per 8 KB tile costs less.)
// (A variant that walked such rows with the general row procedure inside this launch -- one launch for BASELINE config 2 -- was measured and
//  removed: `foo(bar|baz)` over 1 M x 64 B, one row in ten matching, 19.6-20.0 us against the one-launch kernel's 18.1 us (gpurun call
//  r05_c3): with sparse hits the one-launch kernel's match compaction finishes 64 gathered rows per pass ACROSS tiles, this kernel one
//  pass per 8 KB tile.)
#pragma once
#include "fx_tile.hpp"

#ifndef FX_SPAN_GB
#define FX_SPAN_GB 1   // compacted finish passes: 8-symbol groups of the forward window whose lookups are issued together (registers)
#endif
#ifndef FX_SPAN_WAVES
#define FX_SPAN_WAVES 4   // waves per SIMD the aligned instantiations are compiled for (experiment hook)
#endif
#ifndef FX_SPAN_RAG_WAVES
#define FX_SPAN_RAG_WAVES 4   // waves per SIMD the ragged instantiations are compiled for (round 5: three -- their wave-uniform guards, hoisted out of the tile loop, cost ~50 SGPRs and 8-40 VGPRs; round 6 recomputes them at their use sites, fx_tail_here: 100-122 VGPRs)
#endif
#ifndef FX_SPAN_MIN_ROUNDS
#define FX_SPAN_MIN_ROUNDS 3   // launch grid: at least this many rounds of the 1024 resident blocks (the half-row kernel's rule; 3 M x 64 B: 56.2 -> 53.5 us,
                               // 3 M x 128 B packed: 94.8 -> 89.9 us, 1 M x 64 B: 23.1 -> 22.5 us against one round, gpurun call r05_c23)
#endif

// RL = bytes of LDS a row gets: 16 * (its chunk count rounded up to a power of two).  Rows of exactly RL bytes are the aligned
// instantiations; RAG: rows of ANY other length 2 <= Lr < RL (character(20), (80), (100): what Fortran programs declare) stay
// LEFT-ALIGNED in their RL bytes with what follows the text in the wrapped string behind it -- the trailing NUL at byte Lr, then KILL
// symbols (the pad-free scheme of fx_tile.hpp, "Ragged rows, round 4": the loader reads per-row pieces at the row stride, the chunk the
// row ends in is patched after every staging store, the backward pass starts at the row's last byte) -- so every left-to-right walk is
// the aligned kernels' code and the work follows the row length.
template <int RL>
struct FxSpan {
   static_assert(RL == 128 || RL == 64 || RL == 32 || RL == 16, "span kernel: rows of up to 128, 64, 32 or 16 bytes");
   static constexpr int K = 128 / RL;     // rows per lane (one 128-byte span of LDS)
   static constexpr int NCH = RL / 16;    // chunks per row
};

// tile t = bytes [8192 t, 8192 t + 8192) of the batch: piece q * 64 + lane = the 16 bytes at tile offset 16 (64 q + lane) -- 1 KiB fully
// coalesced per instruction.  The buffer resource's extent is the tile's bytes that exist (`total` = n * RL, a multiple of 16: pieces lie
// wholly inside or wholly outside -- nothing behind the batch's last byte is read, whatever the base address).
__device__ __forceinline__ void fx_span_load(uint4 (&v)[8], const uint8_t* __restrict__ rows, const int64_t total, const int64_t t, const uint32_t lane) {
   const int64_t off0 = t << 13;
   const int64_t left = total - off0;
   const uint32_t valid = left <= 0 ? 0u : (uint32_t)(left >= 8192 ? 8192 : left);
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)(left > 0 ? off0 : 0);
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
#pragma unroll
   for (int q = 0; q < 8; ++q) {
      const fx_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u + (uint32_t)q * 1024u, 0, FX_LOAD_AUX);
      v[q] = make_uint4(x.x, x.y, x.z, x.w);
   }
}

// RAG: tile t = rows [64 K t, 64 K t + 64 K).  Piece q * 64 + lane = lane-span 8 q + lane / 8, cell lane % 8 of it = chunk k = cell % NCH of
// the span's row j = cell / NCH: the 16 bytes at row byte 16 k of row (8 q + lane / 8) K + j -- unaligned loads at the row stride; cells whose
// chunk holds no text ask for an address behind the extent and fetch nothing.  The extent: fx_tile.hpp, load_tile_rag (the batch's last
// tile ends it at the batch's last byte; the one text dword that cuts off is rebuilt in LDS after the staging store: fx_last_dword).
template <int RL>
__device__ __forceinline__ void fx_span_load_rag(uint4 (&v)[8], const uint8_t* __restrict__ rows, const int64_t n, const int64_t t, const uint32_t lane, const FxTail& T_) {
   const FxTail T = fx_tail_here(T_);
   constexpr uint32_t K = (uint32_t)FxSpan<RL>::K, NCH = (uint32_t)FxSpan<RL>::NCH;
   const int64_t row0 = (t << 6) * (int64_t)K;
   const int64_t rows_left = n - row0;
   const uint32_t tile_rows = rows_left <= 0 ? 0u : (rows_left >= 64 * (int64_t)K ? 64u * K : (uint32_t)rows_left);
   const uint32_t tile_bytes = tile_rows * T.Lr;
   const int64_t room = rows_left > 64 * (int64_t)K ? (rows_left - 64 * (int64_t)K) * (int64_t)T.Lr : 0;   // bytes of the batch behind this tile
   const bool last_tile = room < 3;
   const uint32_t valid = tile_rows == 0u ? 0u : (last_tile ? tile_bytes + (uint32_t)room : tile_bytes + 3u);
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)(rows_left > 0 ? row0 : 0) * (uint64_t)T.Lr;
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
   const uint32_t cell = lane & 7u, j = cell / NCH, k = cell % NCH;
   const uint32_t voff = k < T.nch ? ((lane >> 3) * K + j) * T.Lr + 16u * k : 0x7FFFFFF0u;
   const uint32_t step = __builtin_amdgcn_readfirstlane(8u * K * T.Lr);
#pragma unroll
   for (int q = 0; q < 8; ++q) {
      const fx_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (uint32_t)q * step, FX_LOAD_AUX);
      v[q] = make_uint4(x.x, x.y, x.z, x.w);
   }
}

// 8 symbols of the row that starts at chunk c0 of lane R's cells, from row position p (a multiple of 8, any value): text, then the
// trailing NUL at position RL, then KILL symbols (the shared end-of-row cell)
template <int RL>
__device__ __forceinline__ void fx_span_group(uint32_t& lo, uint32_t& hi, const uint8_t* tb, const uint8_t* eor, const uint32_t R, const uint32_t c0, const uint32_t p) {
   const uint32_t pc = p < (uint32_t)RL + 8u ? p : (uint32_t)RL + 8u;
   const uint8_t* src = pc >= (uint32_t)RL ? eor + (pc & 8u) : tb + (tile_cell(R, c0 + (pc >> 4)) << 4) + (pc & 8u);
   const uint2 r = *reinterpret_cast<const uint2*>(src);
   lo = r.x;
   hi = r.y;
}

// Right-to-left pass over ONE row (chunks c0 .. c0 + NCH - 1 of the lane's own cells): the half-row kernel's lean loop (fx_tile.hpp,
// FX_HALF4: running maximum instead of the group's eight states, one chunk of LDS prefetch).  gsel = the leftmost 8-byte group that
// holds a hit (0xFFFFFFFF: none), esel = the state entering it, state = the state after the leading NUL.
template <int RL, int SCH, bool RAG, bool LATCH = false, class TabT>
__device__ __forceinline__ void fx_span_back(const uint4* tile, const uint32_t lane, const uint32_t c0, const TabT* __restrict__ tabR, const uint8_t* TRp,
                                             const FastParams& fp, const FxTail& T_, uint32_t& na, uint32_t& gsel, uint32_t& esel, uint32_t& state) {
   using F = typename FxF<SCH>::type;
   constexpr int NCH = FxSpan<RL>::NCH;
   const FxTail T = RAG ? fx_tail_here(T_) : T_;   // (what the guards derive from the row length is recomputed here, not kept live across the tile loop: fx_tile.hpp)
   state = fp.R_start;
   gsel = 0xFFFFFFFFu;
   esel = 0;
   F fa[8], fb[8];
   if constexpr (RAG) {
      // from the row's LAST byte: the chunk the row ends in over its text bytes only, then the whole chunks behind wave-uniform guards (cell
      // addresses stay immediates); chunks behind the text are skipped -- fx_scan_tile's TAIL loop (fx_one.hpp) with the lean chain
      if (T.nb != 0u) {
         const uint4 wp = tile[tile_cell(lane, c0 + T.kt)];
         na |= fx_tail_or(wp, T.nb);
         if (T.nb > 8u) {
            lookup8(fa, wp.z, wp.w, tabR);
            const uint32_t entry = state;
            const uint32_t mx = chain8_back_n<F, LATCH>(fa, state, TRp, T.nb - 8u);
            gsel = mx >= fp.hit_min ? 2u * T.kt + 1u : gsel;
            esel = mx >= fp.hit_min ? entry : esel;
            if (LATCH) state &= FX_LATCH_MASK;
         }
         lookup8(fb, wp.x, wp.y, tabR);
         const uint32_t entry = state;
         const uint32_t nv0 = T.nb < 8u ? T.nb : 8u;
         const uint32_t mx = nv0 == 8u ? chain8_back<F, true, LATCH>(fb, state, TRp) : chain8_back_n<F, LATCH>(fb, state, TRp, nv0);
         gsel = mx >= fp.hit_min ? 2u * T.kt : gsel;
         esel = mx >= fp.hit_min ? entry : esel;
         if (LATCH) state &= FX_LATCH_MASK;
      }
      if (T.kt != 0u) {
         uint4 wk = tile[tile_cell(lane, c0 + T.kt - 1u)];
         lookup8(fa, wk.z, wk.w, tabR);
#pragma unroll
         for (int k = NCH - 2; k >= 0; --k) {   // (kt <= NCH - 1: a ragged row is shorter than RL bytes)
            if ((uint32_t)k < T.kt) {
               na |= wk.x | wk.y | wk.z | wk.w;
               lookup8(fb, wk.x, wk.y, tabR);
               __builtin_amdgcn_sched_barrier(0);
               {
                  const uint32_t entry = state;
                  const uint32_t mx = chain8_back<F, true, LATCH>(fa, state, TRp);
                  gsel = mx >= fp.hit_min ? (uint32_t)(2 * k + 1) : gsel;
                  esel = mx >= fp.hit_min ? entry : esel;
                  if (LATCH) state &= FX_LATCH_MASK;
                  asm volatile("" : "+v"(esel));
               }
               __builtin_amdgcn_sched_barrier(0);
               if (k >= 1) {
                  wk = tile[tile_cell(lane, c0 + (uint32_t)k - 1u)];
                  lookup8(fa, wk.z, wk.w, tabR);
               }
               __builtin_amdgcn_sched_barrier(0);
               {
                  const uint32_t entry = state;
                  const uint32_t mx = chain8_back<F, true, LATCH>(fb, state, TRp);
                  gsel = mx >= fp.hit_min ? (uint32_t)(2 * k) : gsel;
                  esel = mx >= fp.hit_min ? entry : esel;
                  if (LATCH) state &= FX_LATCH_MASK;
                  asm volatile("" : "+v"(esel));
               }
               __builtin_amdgcn_sched_barrier(0);
            }
         }
      }
      const F fz = tabR[0];   // leading NUL: a hit there is the leftmost start
      state = fxstep(fz, state, TRp);
      return;
   }
   uint4 wk = tile[tile_cell(lane, c0 + (uint32_t)NCH - 1u)];
   lookup8(fa, wk.z, wk.w, tabR);
#pragma unroll
   for (int k = NCH - 1; k >= 0; --k) {
      na |= wk.x | wk.y | wk.z | wk.w;
      lookup8(fb, wk.x, wk.y, tabR);
      __builtin_amdgcn_sched_barrier(0);
      {
         const uint32_t entry = state;
         const uint32_t mx = chain8_back<F, true, LATCH>(fa, state, TRp);
         gsel = mx >= fp.hit_min ? (uint32_t)(2 * k + 1) : gsel;
         esel = mx >= fp.hit_min ? entry : esel;
         if (LATCH) state &= FX_LATCH_MASK;   // (the latched format of R, program.h FXP_F_R_LATCH: no running maximum in the chain above)
         asm volatile("" : "+v"(esel));   // select now: otherwise every group's entry state stays live until after the loop
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k >= 1) {
         wk = tile[tile_cell(lane, c0 + (uint32_t)k - 1u)];
         lookup8(fa, wk.z, wk.w, tabR);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
         const uint32_t entry = state;
         const uint32_t mx = chain8_back<F, true, LATCH>(fb, state, TRp);
         gsel = mx >= fp.hit_min ? (uint32_t)(2 * k) : gsel;
         esel = mx >= fp.hit_min ? entry : esel;
         if (LATCH) state &= FX_LATCH_MASK;
         asm volatile("" : "+v"(esel));
      }
      __builtin_amdgcn_sched_barrier(0);
   }
   const F fz = tabR[0];   // leading NUL: a hit there is the leftmost start
   state = fxstep(fz, state, TRp);
}

// FINISH of one row per lane; the result in ONE register: flag | from << 8 | to << 16 (from, to <= 128)
// the row: the row = chunks c0 .. of lane R's cells (R, c0 per lane: a compacted slot, or the lane's own row), g = its
// leftmost hit group, e = the state entering it, nul = the start is the leading NUL; on = this lane has a row.
template <int RL, int SCH, int GB, bool RAG, bool LATCH = false, class TabT>
__device__ __forceinline__ uint32_t fx_span_finish(const uint8_t* tb, const uint8_t* eor, const uint32_t R, const uint32_t c0, const uint32_t g, const uint32_t e,
                                                   const bool nul, const bool on, const TabT* __restrict__ tabR, const TabT* __restrict__ tabA,
                                                   const uint8_t* TRp, const uint8_t* TAp, const FastParams& fp, const uint32_t L) {
   using F = typename FxF<SCH>::type;
   // exact byte of the leftmost hit: re-walk the hit group
   uint32_t s;   // wrapped start index (1 = leading NUL, j + 2 for text byte j)
   {
      uint32_t lo, hi;
      fx_span_group<RL>(lo, hi, tb, eor, R, c0, g * 8u);
      F f[8];
      lookup8(f, lo, hi, tabR);
      uint32_t st = e, loc = 0;
#pragma unroll
      for (int i = 7; i >= 0; --i) {
         if constexpr (RAG) {   // the group the row ends in: only its text bytes were walked (the NUL / KILL symbols behind them are not the row's)
            const uint32_t nx = fxstep(f[i], st, TRp);
            const bool in_text = g * 8u + (uint32_t)i < L;
            st = in_text ? nx : st;
            loc = (in_text && (LATCH ? (nx & FX_LATCH_MASK) >= fp.hit_base : nx >= fp.hit_min)) ? (uint32_t)i : loc;
         } else {
            st = fxstep(f[i], st, TRp);
            loc = (LATCH ? (st & FX_LATCH_MASK) >= fp.hit_base : st >= fp.hit_min) ? (uint32_t)i : loc;   // (a re-walk: the latch may be set by an earlier step)
         }
      }
      s = nul ? 1u : g * 8u + 2u + loc;
   }
   // ---- left-to-right pass from the start: anchored DFA, longest accept (api_internal_m.F90:119-148) ----
   uint32_t cur = on ? fp.A_init : 0u;
   uint32_t mm = 0;                          // max_match (wrapped index of the byte after the match)
   uint32_t j = nul ? 0u : s - 2u;           // 0-based text index of the next byte to consume
   if (__builtin_amdgcn_ballot_w64(nul) != 0) {
      const F f = tabA[0];
      const uint32_t nx = fxstep(f, cur, TAp);
      mm = (nul && nx >= fp.acc_min) ? 2u : 0u;
      cur = nul ? nx : cur;
   }
   {
      // first 32 symbols straight-line: five aligned 8-byte reads, a byte shift to start exactly at j; per 8-byte group only "any accept"
      // + entry state are kept and the last accepting group is re-walked for the exact byte
      uint32_t o[8];
      {
         const uint32_t base = j & ~7u, sh = j & 7u;
         uint32_t d[10];
#pragma unroll
         for (int q = 0; q < 5; ++q) fx_span_group<RL>(d[2 * q], d[2 * q + 1], tb, eor, R, c0, base + 8u * (uint32_t)q);
         const uint32_t up = 0u - ((sh >> 2) & 1u);   // all ones when the stream starts in the odd dword
         uint32_t ee[9];
#pragma unroll
         for (int k = 0; k < 9; ++k) ee[k] = (up & d[k + 1]) | (~up & d[k]);
#pragma unroll
         for (int k = 0; k < 8; ++k) o[k] = __builtin_amdgcn_alignbyte(ee[k + 1], ee[k], sh & 3u);
      }
      uint32_t gl = 0xFFFFFFFFu, el = 0, blo = 0, bhi = 0;   // (GB: 8-symbol groups whose lookups are issued together)
#pragma unroll
      for (int gb = 0; gb < 4; gb += GB) {
         F f[8 * GB];
#pragma unroll
         for (int q = 0; q < GB; ++q) lookup8(&f[8 * q], o[2 * (gb + q)], o[2 * (gb + q) + 1], tabA);
#pragma unroll
         for (int q = 0; q < GB; ++q) {
            const uint32_t entry = cur;
            uint32_t st[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
               cur = fxstep(f[8 * q + i], cur, TAp);
               st[i] = cur;
            }
            const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
            const bool acc = mx >= fp.acc_min;
            gl = acc ? (uint32_t)(gb + q) : gl;
            el = acc ? entry : el;
            blo = acc ? o[2 * (gb + q)] : blo;
            bhi = acc ? o[2 * (gb + q) + 1] : bhi;
         }
      }
      {
         F fr8[8];
         lookup8(fr8, blo, bhi, tabA);
         uint32_t st = el, loc = 0;
#pragma unroll
         for (int i = 0; i < 8; ++i) {
            st = fxstep(fr8[i], st, TAp);
            loc = st >= fp.acc_min ? (uint32_t)i : loc;
         }
         mm = gl != 0xFFFFFFFFu ? j + 8u * gl + loc + 3u : mm;
      }
      j += 32u;
      // matches longer than the window: 8 symbols per round trip; a rolling window of two aligned 8-byte groups, the group after them
      // read one round ahead.  Wave-uniform: dead lanes (state 0 is absorbing and below acc_min) ride along.
      if (__builtin_amdgcn_ballot_w64(cur != 0u) != 0) {
         const uint32_t sh = j & 7u, up = 0u - ((sh >> 2) & 1u);
         uint32_t gb = j & ~7u;
         uint32_t t0[2], t1[2];
         fx_span_group<RL>(t0[0], t0[1], tb, eor, R, c0, gb);
         fx_span_group<RL>(t1[0], t1[1], tb, eor, R, c0, gb + 8u);
         do {
            uint32_t t2[2];
            fx_span_group<RL>(t2[0], t2[1], tb, eor, R, c0, gb + 16u);
            const uint32_t e0 = (up & t0[1]) | (~up & t0[0]), e1 = (up & t1[0]) | (~up & t0[1]), e2 = (up & t1[1]) | (~up & t1[0]);
            const uint32_t o0 = __builtin_amdgcn_alignbyte(e1, e0, sh & 3u), o1 = __builtin_amdgcn_alignbyte(e2, e1, sh & 3u);
            F f8[8];
            lookup8(f8, o0, o1, tabA);
            uint32_t loc = 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
               cur = fxstep(f8[i], cur, TAp);
               loc = cur >= fp.acc_min ? (uint32_t)i : loc;
            }
            mm = loc != 8u ? j + loc + 3u : mm;
            j += 8u;
            gb += 8u;
            t0[0] = t1[0]; t0[1] = t1[1];
            t1[0] = t2[0]; t1[1] = t2[1];
         } while (__builtin_amdgcn_ballot_w64(cur != 0u) != 0);
      }
   }
   // api_internal_m.F90:140-148: from = max(start - 1, 1), to = max_match - 2 clamped to the row; a match needs to > 0
   uint32_t out = 0;
   if (on && mm > 2u) {
      const uint32_t fr = s >= 2u ? s - 1u : 1u;
      const uint32_t tt = mm >= L + 2u ? L : mm - 2u;
      out = 1u | (fr << 8) | (tt << 16);
   }
   return out;
}

// n_deferred: this call's group of four counter words (words of consecutive calls alternate; [0] "tiles were deferred", [2] / [3] the
// sample FX_ADAPT_CALLS describes)
template <int RL, int SCH, bool PACKED, bool RAG, bool LATCH = false>
__global__ __launch_bounds__(256, RAG ? FX_SPAN_RAG_WAVES : FX_SPAN_WAVES) void fx_search_span(const uint8_t* __restrict__ rows, const int64_t n, const uint8_t* __restrict__ prog, const FastParams fp,
                                                          uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                          uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ clear_next, uint8_t* __restrict__ marks,
                                                          const uint32_t Lr_in) {
   // fp.out_mode == 1: PACKED results (what a multi-GPU host gathers): `flags` = 1 bit per row (row i = bit i & 63 of the 64-bit
   // word i >> 6), `from` / `to` = one byte per row (rows of up to 128 bytes); "this tile is left to the follow-up" = marks[64-row tile] = 1
   // (The chain tables were tried as well -- 128-byte rows of a 17-state pattern: 0.504 ms against 0.494 ms for the 64-byte halves of
   //  fx_search_fast<4, ..., LONG>, gpurun call r05_c14: whole lines instead of split ones, but the dependent LDS read per byte is what that
   //  scheme waits for -- and are not built.)
   static_assert(SCH == 0 || SCH == 2, "class-level v_perm or nibble tables");
   static_assert(!LATCH || SCH == 0, "the latched format of R is an 8-state v_perm table");
   using S = FxSpan<RL>;
   using F = typename FxF<SCH>::type;
   constexpr int K = S::K, NCH = S::NCH;
   constexpr bool COMPACT = K >= 2;
   if (blockIdx.x == 0 && threadIdx.x == 0) {   // (a first pass of the multi-pass kind: it zeroes the next call's counter group)
      clear_next[0] = 0u;
      clear_next[1] = 0u;
      clear_next[2] = 0u;
      clear_next[3] = 0u;
   }
   __shared__ F tabR_s[256];
   __shared__ F tabA_s[256];
   __shared__ __attribute__((aligned(16))) uint4 tiles[4 * 512 + 4];   // 4 waves x 64 spans x 8 cells, then the four shared end-of-row cells
   __shared__ uint32_t slot_q[COMPACT ? 4 * 128 : 1];   // compaction: per wave a ring of 128 slots (K > 2: whole passes are finished in between so that it never overflows)
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const uint32_t Lr = RAG ? Lr_in : (uint32_t)RL;   // row length in bytes (RAG: 2 <= Lr < RL)
   const FxTail tl = fx_tail_of(Lr);
   const int64_t total = n * (int64_t)RL;               // (aligned rows: the batch's bytes)
   const int64_t n_tiles = (n + 64 * K - 1) / (64 * K);
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   // first passes whose tiles mostly hold UTF-8 (FX_ADAPT_CALLS, fx_tile.hpp): the follow-up's persistent word says "skip the loads"
   const bool adapt = (fp.defer_tiles & 2u) != 0u;
   if (adapt) {
      const uint32_t* hintw = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(n_deferred) & ~uintptr_t(31)) + 32u);
      if (__builtin_amdgcn_readfirstlane(hintw[0]) != 0u) {
         for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
            const int64_t r0 = ((t << 6) + lane) * K;
            if constexpr (PACKED) {
               if (lane < (uint32_t)K) marks[t * K + lane] = 1u;
               continue;
            }
#pragma unroll
            for (int i = 0; i < K; ++i)
               if (r0 + i < n) flags[r0 + i] = FX_NEEDS_GENERAL;
         }
         if (lane == 0) n_deferred[0] = 1u;
         return;
      }
   }
   // start-up: the table entries are READ first, then the first tile's loads go out, and only then are the entries written to LDS
   const uint2 t_r = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16R : (LATCH ? h->off_fastRL : h->off_fastR)))[threadIdx.x];
   const uint2 t_a = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16A : h->off_fastA))[threadIdx.x];
   __builtin_amdgcn_sched_barrier(0);
   uint4 stage[8];
   if constexpr (RAG) fx_span_load_rag<RL>(stage, rows, n, wave_global, lane, tl);
   else fx_span_load(stage, rows, total, wave_global, lane);
   reinterpret_cast<uint2*>(tabR_s)[threadIdx.x] = t_r;
   reinterpret_cast<uint2*>(tabA_s)[threadIdx.x] = t_a;
   uint4* const tile = tiles + wave * 512u;
   uint4* const eor_cell = tiles + 4 * 512 + wave;
   // (ragged rows: the NUL sits inside the row's cells, right behind the text, and the shared cell holds KILL symbols only -- a second NUL
   //  would be a second line end to patterns like `$$`)
   if (lane == 0) *eor_cell = make_uint4(RAG ? 0xFEFEFEFEu : 0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
   if constexpr (RAG) {   // chunks behind the text of each of the lane's rows: the trailing NUL / KILL symbols, written ONCE (the loader never touches them)
      for (uint32_t j = 0; j < (uint32_t)K; ++j)
         for (uint32_t k = tl.nch; k < (uint32_t)NCH; ++k)
            tile[tile_cell(lane, j * (uint32_t)NCH + k)] = (k == tl.kt) ? make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu)
                                                                         : make_uint4(0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
   }
   __syncthreads();
   const F* tabR = tabR_s;
   const F* tabA = tabA_s;
   const uint8_t* const TRp = nullptr;   // (the v_perm and nibble steps read no table through the state)
   const uint8_t* const TAp = nullptr;
   const uint8_t* const tb = reinterpret_cast<const uint8_t*>(tile);
   const uint8_t* const eor = reinterpret_cast<const uint8_t*>(eor_cell);
   constexpr uint32_t QCAP = 128u;
   uint32_t* const sq = slot_q + (COMPACT ? wave * QCAP : 0u);
   bool any_deferred = false;
   uint32_t n_def = 0, n_seen = 0;
   // (one wide store per array only where the arrays are aligned for it: result set i of a caller may start anywhere -- wave-uniform)
   const bool wide_ok = ((reinterpret_cast<uintptr_t>(flags) & (uintptr_t)(K - 1)) | (reinterpret_cast<uintptr_t>(from) & (uintptr_t)(K >= 4 ? 15 : 4 * K - 1)) |
                         (reinterpret_cast<uintptr_t>(to) & (uintptr_t)(K >= 4 ? 15 : 4 * K - 1))) == 0;
   for (int64_t t = wave_global; t < n_tiles;) {
      const int64_t t_next = t + wave_stride;
      n_seen += 1u;
      bool defer_tile = false;
      {
         // cheap sampled look at the staged bytes: a tile that shows a byte >= 0x80 here is deferred without being scanned
         const uint32_t smp = stage[0].x | stage[0].w | stage[4].y | stage[7].z;
         defer_tile = __builtin_amdgcn_ballot_w64((smp & 0x80808080u) != 0) != 0;
      }
      if constexpr (RAG) {
         if ((lane & 7u) % (uint32_t)NCH < tl.nch) store_tile<8>(stage, tile, lane);   // (the chunks behind the text keep their KILL symbols)
         {   // the batch's last tile: the one text dword its exact extent cut off (fx_last_dword, fx_tile.hpp), written into its row's cell
            FxLastDword d;
            if (fx_last_dword(d, rows, (t << 6) * (int64_t)K, n, 64u * (uint32_t)K, tl.Lr)) {   // wave-uniform
               if (lane == 0)
                  reinterpret_cast<uint32_t*>(tile)[(tile_cell(d.row / (uint32_t)K, (d.row % (uint32_t)K) * (uint32_t)NCH + (d.m >> 2)) << 2) + (d.m & 3u)] = d.word;
            }
         }
         fx_span_load_rag<RL>(stage, rows, n, t_next, lane, tl);   // the ONE place the staging registers are reloaded
         if (tl.nb != 0u) {   // what follows the text in the chunk each row ends in: the trailing NUL, then KILL symbols
#pragma unroll
            for (int j = 0; j < K; ++j) {
               const uint32_t cc = tile_cell(lane, (uint32_t)(j * NCH) + tl.kt);
               uint4 c = tile[cc];
               c.x = fx_tail_word(c.x, 0u, tl.nb);
               c.y = fx_tail_word(c.y, 4u, tl.nb);
               c.z = fx_tail_word(c.z, 8u, tl.nb);
               c.w = fx_tail_word(c.w, 12u, tl.nb);
               tile[cc] = c;
            }
         }
      } else {
         store_tile<8>(stage, tile, lane);
         fx_span_load(stage, rows, total, t_next, lane);   // the ONE place the staging registers are reloaded
      }
      // a row's result: flag | from << 8 | to << 16 -- or, until its slot is finished, the slot number with bit 31 set.  K >= 4 (rows of up
      // to 32 bytes: from, to <= 32) keeps TWO rows per register, 16 bits each: flag | from << 1 | to << 7, slot number with bit 15 set --
      // eight result registers were what the K = 8 instantiations spilled at four waves per SIMD
      constexpr bool RES16 = K >= 4;
      uint32_t resw[RES16 ? K / 2 : K];
#pragma unroll
      for (int i = 0; i < (RES16 ? K / 2 : K); ++i) resw[i] = 0u;
      auto res_get = [&](const int i) -> uint32_t {   // (i is a compile-time constant at every call site: the loops are unrolled)
         if constexpr (RES16) {
            const uint32_t h = (resw[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
            return (h & 0x8000u) != 0u ? (0x80000000u | (h & 0x7FFFu)) : ((h & 1u) | (((h >> 1) & 63u) << 8) | (((h >> 7) & 63u) << 16));
         } else return resw[i];
      };
      auto res_raw16 = [&](const int i) -> uint32_t { return (resw[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu; };   // RES16: the packed half as it is
      (void)res_raw16;
      auto res_set = [&](const int i, const uint32_t v) {
         if constexpr (RES16) {
            const uint32_t h = (v & 0x80000000u) != 0u ? (0x8000u | (v & 0x7FFFu)) : ((v & 1u) | (((v >> 8) & 63u) << 1) | (((v >> 16) & 63u) << 7));
            resw[i >> 1] = (i & 1) ? ((resw[i >> 1] & 0x0000FFFFu) | (h << 16)) : ((resw[i >> 1] & 0xFFFF0000u) | h);
         } else resw[i] = v;
      };
      uint32_t na = 0;
      bool sink = false;   // this lane has a row that ended in the overlap state of a bordered prefix literal
      if (!defer_tile) {
         if constexpr (!COMPACT) {
            uint32_t gsel, esel, state;
            fx_span_back<RL, SCH, RAG, LATCH>(tile, lane, 0u, tabR, TRp, fp, tl, na, gsel, esel, state);
            const bool hit = gsel != 0xFFFFFFFFu, nul = state >= fp.hit_min;
            sink = fp.inv_on != 0u && state == fp.inv;
            const bool want = hit || nul;
            if (__builtin_amdgcn_ballot_w64(want) != 0)
               res_set(0, fx_span_finish<RL, SCH, 2, RAG, LATCH>(tb, eor, lane, 0u, hit ? gsel : 0u, esel, nul, want, tabR, tabA, TRp, TAp, fp, Lr));
         } else {
            // every row's backward pass; the rows that need the finish take a slot: lane | row << 6 | hit group << 9 | nul << 13 | entry state << 14.
            // Until its slot is finished a row's result register holds the slot number (bit 31 set).
            uint32_t cnt = 0;   // slots taken (wave-uniform)
            auto finish_slots = [&](const uint32_t base) {   // slots base .. base + 63: lane q finishes slot base + q, the result replaces the entry
               const bool on = base + lane < cnt;
               const uint32_t en = on ? sq[(base + lane) % QCAP] : 0u;
               const uint32_t e8 = en >> 14;
               const uint32_t r = fx_span_finish<RL, SCH, FX_SPAN_GB, RAG, LATCH>(tb, eor, en & 63u, ((en >> 6) & 7u) * (uint32_t)NCH, (en >> 9) & 15u, SCH == 0 ? e8 * 0x01010101u : e8,
                                                                           ((en >> 13) & 1u) != 0u, on, tabR, tabA, TRp, TAp, fp, Lr);
               if (on) sq[(base + lane) % QCAP] = r;
            };
            uint32_t done = 0;   // slots finished (wave-uniform; K > 2: the ring holds 128 and whole passes are finished when the next row's slots may not fit)
#pragma unroll
            for (int jr = K - 1; jr >= 0; --jr) {
               uint32_t gsel, esel, state;
               // (the row's cell addresses -- (lane ^ chunk) << 4 | base, one per chunk column -- are loop-invariant, and hoisted out of the tile loop they
               //  are K more live registers, the ones the K = 8 instantiations spilled at four waves per SIMD: computed here, from an opaque copy of the lane)
               uint32_t lane_here = lane;
               if constexpr (K >= 4) asm volatile("" : "+v"(lane_here));
               fx_span_back<RL, SCH, RAG, LATCH>(tile, lane_here, (uint32_t)(jr * NCH), tabR, TRp, fp, tl, na, gsel, esel, state);
               const bool hit = gsel != 0xFFFFFFFFu, nul = state >= fp.hit_min;
               sink = sink || (fp.inv_on != 0u && state == fp.inv);
               const bool want = hit || nul;
               const uint64_t wm = __builtin_amdgcn_ballot_w64(want);
               if (want) {
                  const uint32_t slot = cnt + (uint32_t)__builtin_popcountll(wm & ((1ull << lane) - 1ull));
                  res_set(jr, 0x80000000u | slot);
                  sq[slot % QCAP] = lane | ((uint32_t)jr << 6) | ((hit && !nul ? gsel : 0u) << 9) | ((nul ? 1u : 0u) << 13) | ((SCH == 0 ? (esel & 0xFFu) : esel) << 14);
               }
               cnt += (uint32_t)__builtin_popcountll(wm);
               if constexpr (K > 2) {   // (finish whole passes before a later row's slots could overwrite unfinished ones)
                  while (cnt - done > QCAP - 64u) {
                     finish_slots(done);
                     // owners of the finished slots take their results now: the ring may reuse the slots
#pragma unroll
                     for (int j2 = K - 1; j2 >= jr; --j2) {
                        const uint32_t rv = res_get(j2);
                        if ((rv & 0x80000000u) != 0u && (rv & 0x7FFFFFFFu) < done + 64u) res_set(j2, sq[(rv & 0x7FFFFFFFu) % QCAP]);
                     }
                     done += 64u;
                  }
               }
            }
            for (uint32_t base = done; base < cnt; base += 64u) finish_slots(base);
#pragma unroll
            for (int jr = 0; jr < K; ++jr) {
               const uint32_t rv = res_get(jr);
               if ((rv & 0x80000000u) != 0u) res_set(jr, sq[(rv & 0x7FFFFFFFu) % QCAP]);
            }
         }
         // bytes >= 0x80, overlap rows: the whole TILE goes to the follow-up (wave-uniform)
         defer_tile = __builtin_amdgcn_ballot_w64((na & 0x80808080u) != 0u || sink) != 0;
      }
      if (defer_tile) {   // (the stores below write FX_NEEDS_GENERAL flags for it)
         any_deferred = true;
         n_def += 1u;
      }
      // (the lane's first row, computed HERE through an opaque copy of t: as a value of the loop's top it is two more registers live across the scan)
      int64_t t_res = t;
      asm volatile("" : "+s"(t_res));
      const int64_t row_first = ((t_res << 6) + lane) * K;
      if constexpr (PACKED) {
         // ---- PACKED results: the tile's 64 K rows = K flag words (8 K bytes of bits), one span byte per row; a deferred tile's words and
         //      spans are the follow-up's ----
         {   // (scalar base + lane, as for the flag bytes below)
            uint8_t* mk = marks + t_res * K;
            asm volatile("" : "+s"(mk));
            if (lane < (uint32_t)K) mk[lane] = defer_tile ? 1u : 0u;
         }
         if (!defer_tile) {
            uint32_t bits = 0, f8lo = 0, f8hi = 0, t8lo = 0, t8hi = 0;   // the lane's K flag bits, its K from / to bytes (rows behind the batch's end: zero)
#pragma unroll
            for (int i = 0; i < K; ++i) {
               const uint32_t r = row_first + i < n ? res_get(i) : 0u;
               bits |= (r & 1u) << i;
               if (i < 4) {
                  f8lo |= ((r >> 8) & 0xFFu) << (8 * i);
                  t8lo |= ((r >> 16) & 0xFFu) << (8 * i);
               } else {
                  f8hi |= ((r >> 8) & 0xFFu) << (8 * (i - 4));
                  t8hi |= ((r >> 16) & 0xFFu) << (8 * (i - 4));
               }
            }
            uint8_t* wbytes = flags + (t_res << 3) * K;   // this tile's 8 K bytes of flag bits
            asm volatile("" : "+s"(wbytes));             // (a scalar base + the lane: `flags + lane` hoisted out of the tile loop is a 64-bit value per lane, spilled at K = 8)
            const int64_t word_bytes = ((n + 63) >> 6) << 3;   // whole 64-bit words exist for the batch's rows: nothing is written behind them
            const int64_t bytes_left = word_bytes - (t_res << 3) * K;   // (wave-uniform; compared with the lane as a 32-bit scalar)
            uint32_t bytes_left32 = bytes_left >= 64 ? 64u : (bytes_left > 0 ? (uint32_t)bytes_left : 0u);
            asm volatile("" : "+s"(bytes_left32));
            const bool byte_ok = lane < bytes_left32;
            if constexpr (K == 1) {
               const uint64_t m = __builtin_amdgcn_ballot_w64(bits != 0u);
               if (lane == 0) reinterpret_cast<uint64_t*>(flags)[t] = m;
            } else if constexpr (K == 8) {
               if (byte_ok) wbytes[lane] = (uint8_t)bits;   // the lane's eight rows are one byte of the bit array
            } else {
               // 8 / K lanes make one byte: every lane leaves its K bits in a byte of the wave's slot list (free now), lane q < 8 K reads
               // the 8 / K bytes of its lanes and folds them
               uint8_t* const sb = reinterpret_cast<uint8_t*>(sq);
               sb[lane] = (uint8_t)bits;
               if (lane < 8u * (uint32_t)K) {
                  uint32_t x;
                  if constexpr (K == 2) {
                     x = *reinterpret_cast<const uint32_t*>(sb + 4u * lane);
                     x = x | (x >> 6) | (x >> 12) | (x >> 18);
                  } else {
                     x = *reinterpret_cast<const uint16_t*>(sb + 2u * lane);
                     x = x | (x >> 4);
                  }
                  if (byte_ok) wbytes[lane] = (uint8_t)x;
               }
            }
            // (whole words are written: bits of rows behind the batch's end are zero, and the layout rounds the bit array up to 16 bytes)
            uint8_t* const pf = reinterpret_cast<uint8_t*>(from) + row_first;
            uint8_t* const pt = reinterpret_cast<uint8_t*>(to) + row_first;
            if (row_first + K <= n) {
               if constexpr (K == 1) {
                  pf[0] = (uint8_t)f8lo;
                  pt[0] = (uint8_t)t8lo;
               } else if constexpr (K == 2) {
                  *reinterpret_cast<uint16_t*>(pf) = (uint16_t)f8lo;
                  *reinterpret_cast<uint16_t*>(pt) = (uint16_t)t8lo;
               } else if constexpr (K == 4) {
                  *reinterpret_cast<uint32_t*>(pf) = f8lo;
                  *reinterpret_cast<uint32_t*>(pt) = t8lo;
               } else {
                  *reinterpret_cast<uint2*>(pf) = make_uint2(f8lo, f8hi);
                  *reinterpret_cast<uint2*>(pt) = make_uint2(t8lo, t8hi);
               }
            } else {
#pragma unroll
               for (int i = 0; i < K; ++i)
                  if (row_first + i < n) {
                     pf[i] = (uint8_t)((res_get(i) >> 8) & 0xFFu);
                     pt[i] = (uint8_t)((res_get(i) >> 16) & 0xFFu);
                  }
            }
         }
         t = t_next;
         continue;
      }
      // ---- results: the lane's K consecutive rows, one store per array when all of them exist ----
      auto fl = [&](int i) -> uint32_t { return defer_tile ? (uint32_t)FX_NEEDS_GENERAL : (res_get(i) & 0xFFu); };
      auto fr = [&](int i) -> int32_t { return (int32_t)((res_get(i) >> 8) & 0xFFu); };
      auto tt = [&](int i) -> int32_t { return (int32_t)((res_get(i) >> 16) & 0xFFu); };
      if (row_first + K <= n && wide_ok) {
         if constexpr (K == 1) {
            flags[row_first] = (uint8_t)fl(0);
            if (!defer_tile) {
               from[row_first] = fr(0);
               to[row_first] = tt(0);
            }
         } else if constexpr (K == 2) {
            *reinterpret_cast<uint16_t*>(flags + row_first) = (uint16_t)(fl(0) | (fl(1) << 8));
            if (!defer_tile) {
               *reinterpret_cast<int2*>(from + row_first) = make_int2(fr(0), fr(1));
               *reinterpret_cast<int2*>(to + row_first) = make_int2(tt(0), tt(1));
            }
         } else if constexpr (K == 4) {
            *reinterpret_cast<uint32_t*>(flags + row_first) = fl(0) | (fl(1) << 8) | (fl(2) << 16) | (fl(3) << 24);
            if (!defer_tile) {
               *reinterpret_cast<int4*>(from + row_first) = make_int4(fr(0), fr(1), fr(2), fr(3));
               *reinterpret_cast<int4*>(to + row_first) = make_int4(tt(0), tt(1), tt(2), tt(3));
            }
         } else {
            *reinterpret_cast<uint2*>(flags + row_first) = make_uint2(fl(0) | (fl(1) << 8) | (fl(2) << 16) | (fl(3) << 24), fl(4) | (fl(5) << 8) | (fl(6) << 16) | (fl(7) << 24));
            if (!defer_tile) {
               *reinterpret_cast<int4*>(from + row_first) = make_int4(fr(0), fr(1), fr(2), fr(3));
               *reinterpret_cast<int4*>(from + row_first + 4) = make_int4(fr(4), fr(5), fr(6), fr(7));
               *reinterpret_cast<int4*>(to + row_first) = make_int4(tt(0), tt(1), tt(2), tt(3));
               *reinterpret_cast<int4*>(to + row_first + 4) = make_int4(tt(4), tt(5), tt(6), tt(7));
            }
         }
      } else {
#pragma unroll
         for (int i = 0; i < K; ++i)
            if (row_first + i < n) {
               flags[row_first + i] = (uint8_t)fl(i);
               if (!defer_tile) {
                  from[row_first + i] = fr(i);
                  to[row_first + i] = tt(i);
               }
            }
      }
      t = t_next;
   }
   // one plain store per wave (the value only gates the follow-up); the sample of FX_ADAPT_CALLS: two atomics from every 256th wave
   if (any_deferred && lane == 0) n_deferred[0] = 1u;
   if (adapt && (wave_global & 255) == 0 && lane == 0) {
      atomicAdd(&n_deferred[3], n_seen);
      if (n_def != 0u) atomicAdd(&n_deferred[2], n_def);
   }
}

// ctr: this call's counter group
template <int RL, int SCH>
hipError_t launch_span(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, int32_t* from, int32_t* to, uint32_t* ctr,
                       hipStream_t st, uint8_t* marks, uint32_t Lr) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ctr) ^ 16u);   // the other parity's group of four words
   const int64_t total = n * (int64_t)Lr;
   constexpr int64_t TROWS = 64 * FxSpan<RL>::K;
   const int64_t n_tiles = (n + TROWS - 1) / TROWS;
   if (Lr < 2u || Lr > (uint32_t)RL) return hipErrorInvalidValue;
   int64_t blocks = (n_tiles + 3) / 4;
   // whole rounds of the four resident blocks per CU, one per 225 MB of rows (the half-row kernel's rule: fx_tile.hpp); FXAMD_HALF_ROUNDS: experiment hook
   int64_t rounds = fx_env().half_rounds;
   if (rounds <= 0) {
      rounds = total / ((int64_t)225 << 20);
      if (rounds < FX_SPAN_MIN_ROUNDS) rounds = FX_SPAN_MIN_ROUNDS;
      if (rounds > 64) rounds = 64;
   }
   if (blocks > (int64_t)256 * 4 * rounds) blocks = (int64_t)256 * 4 * rounds;
   const int env_blocks = fx_env().one_blocks;   // FXAMD_ONE_BLOCKS, test hook: a tiny grid, many tiles per wave (queue overflow mid-loop)
   if (env_blocks > 0 && blocks > env_blocks) blocks = env_blocks;
   if (blocks < 1) blocks = 1;
   if (!from || !to) return hipErrorInvalidValue;   // (searches with spans only: fxamd.hip, span_kind)
   const bool rag = Lr != (uint32_t)RL;
#define FX_SPAN_GO(P, R) hipLaunchKernelGGL((fx_search_span<RL, SCH, P, R>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, from, to, ctr, clear_next, marks, Lr)
#define FX_SPAN_GO_L(P, R) hipLaunchKernelGGL((fx_search_span<RL, SCH, P, R, true>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, from, to, ctr, clear_next, marks, Lr)
   if constexpr (SCH != 0) {   // nibble tables: plain results (fxamd.hip, span_first); round 6: ragged rows too (101-124 VGPRs at four waves per SIMD, no scratch)
      if (fp.out_mode != 0u || fp.latch != 0u) return hipErrorInvalidValue;
      if (rag) FX_SPAN_GO(false, true);
      else FX_SPAN_GO(false, false);
   } else if (fp.latch != 0u) {   // the latched format of R (FXP_F_R_LATCH programs: no running maximum in the backward pass)
      if (fp.out_mode != 0u) {
         if (rag) FX_SPAN_GO_L(true, true);
         else FX_SPAN_GO_L(true, false);
      } else {
         if (rag) FX_SPAN_GO_L(false, true);
         else FX_SPAN_GO_L(false, false);
      }
   } else if (fp.out_mode != 0u) {
      if (rag) FX_SPAN_GO(true, true);
      else FX_SPAN_GO(true, false);
   } else {
      if (rag) FX_SPAN_GO(false, true);
      else FX_SPAN_GO(false, false);
   }
#undef FX_SPAN_GO
#undef FX_SPAN_GO_L
   return hipGetLastError();
}
#define FX_SPAN_SIG (const uint8_t*, int64_t, const uint8_t*, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t*, hipStream_t, uint8_t*, uint32_t)
