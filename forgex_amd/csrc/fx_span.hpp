// fx_search_span: `.in.` / regex with spans over rows of 128 / 64 / 32 bytes -- a lane owns a 256-byte SPAN of K = 256 / RL whole rows (round 5).
//
// The headline kernel (fx_search_fast<8, ..., LONG>: 256-byte rows staged as two 128-byte halves, 8 KB of tile per wave, 127 VGPRs, four
// waves per SIMD) runs at 0.72 of the HBM peak; the one-launch kernel that took 128- and 64-byte rows (fx_search_one<8 / 4>) at 0.58 / 0.53:
// a 64-row tile of short rows pays the per-tile fixed work (staging, exact start, forward window, result stores) for half / a quarter of
// the bytes, and its 163 VGPRs stop at three waves per SIMD.  Rows are independent (reference src/forgex.F90:74: the operators are
// elemental), so the SAME memory path serves short rows when the lane's 256 contiguous bytes are read as K whole rows instead of one:
//   * a wave's tile = 64 spans = 16 KB of contiguous bytes = 64 K rows; staged as two 128-byte halves of every span (whole 128-byte
//     lines, coalesced 16-byte pieces, the half-row kernel's loader and swizzled store), right half first;
//   * a half holds RH = 128 / RL whole rows of the lane; each is scanned on its own -- reverse automaton from the row's last byte with a
//     fresh start state (api_internal_m.F90:108-155: the leftmost start with a non-empty match), the leading NUL, exact start by re-walking
//     ONE 8-byte group, anchored automaton forwards from the start (a 32-symbol window, then 8 symbols per trip) -- NO state crosses a
//     row boundary, and everything a row needs is in LDS while it is scanned (its bytes + one shared end-of-row cell: NUL, KILL x 15);
//   * the lane keeps its K results and writes them with ONE store per array (K flag bytes, K x int32 from, K x int32 to: consecutive rows).
// Bytes >= 0x80: programs whose class-level tables decode UTF-8 mark the tile's rows FX_NEEDS_GENERAL for ONE gated follow-up (the
// one-launch kernel over marked tiles: byte-level tables or the in-LDS decode, exception queues inside) exactly as the half-row pipeline
// of 256-byte rows does (fxamd.hip, last_path 16 -> 18); programs whose tables cannot decode (GEN: candidate-list driver programs such as
// BASELINE config 2's `foo(bar|baz)`) queue such ROWS -- and rows that end in the overlap state of a bordered prefix literal -- per wave
// and walk them with the general row procedure inside the same launch (fxrow::run_row, the body of fx_general), so that config is ONE
// launch with no host-side state.
#pragma once
#include "fx_tile.hpp"

template <int RL>
struct FxSpan {
   static_assert(RL == 128 || RL == 64 || RL == 32, "span kernel: rows of 128, 64 or 32 bytes");
   static constexpr int K = 256 / RL;     // rows per lane (one 256-byte span)
   static constexpr int RH = 128 / RL;    // rows per staged half
   static constexpr int NCH = RL / 16;    // chunks per row
};

// row accessor of the general procedure: the row in global memory (GEN queue)
struct FxSpanGlobalRow {
   const uint8_t* p;
   __device__ __forceinline__ uint32_t operator[](int j) const { return p[j]; }
};

// half `hf` (1 = bytes 128..255, 0 = bytes 0..127 of every span) of tile t: piece q * 64 + lane = span 8 q + lane / 8, chunk lane % 8 of
// that half -- eight lanes read 128 contiguous bytes (one line when the batch is line-aligned).  The buffer resource's extent is the
// tile's bytes that exist (`total` = n * RL, a multiple of 16: pieces lie wholly inside or wholly outside, nothing behind the batch is read).
__device__ __forceinline__ void fx_span_load(uint4 (&v)[8], const uint8_t* __restrict__ rows, const int64_t total, const int64_t t, const uint32_t hf,
                                             const uint32_t lane, const bool enable) {
   const int64_t off0 = t << 14;
   const int64_t left = total - off0;
   const uint32_t valid = (!enable || left <= 0) ? 0u : (uint32_t)(left >= 16384 ? 16384 : left);
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)(left > 0 ? off0 : 0);
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
   const uint32_t voff = (lane >> 3) * 256u + (lane & 7u) * 16u;
   const uint32_t s0 = __builtin_amdgcn_readfirstlane(hf * 128u);
#pragma unroll
   for (int q = 0; q < 8; ++q) {
      const fx_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s0 + (uint32_t)q * 2048u, FX_LOAD_AUX);
      v[q] = make_uint4(x.x, x.y, x.z, x.w);
   }
}

// 8 symbols of the row that starts at chunk C0 of the lane's cells, from row position p (a multiple of 8, any value): text, then the
// trailing NUL at position RL, then KILL symbols (the shared end-of-row cell)
template <int RL>
__device__ __forceinline__ void fx_span_group(uint32_t& lo, uint32_t& hi, const uint8_t* tb, const uint8_t* eor, const uint32_t lane, const uint32_t c0, const uint32_t p) {
   const uint32_t pc = p < (uint32_t)RL + 8u ? p : (uint32_t)RL + 8u;
   const uint8_t* src = pc >= (uint32_t)RL ? eor + (pc & 8u) : tb + (tile_cell(lane, c0 + (pc >> 4)) << 4) + (pc & 8u);
   const uint2 r = *reinterpret_cast<const uint2*>(src);
   lo = r.x;
   hi = r.y;
}

// what one row's scan yields, in ONE register (the lane holds its K rows' results until the tile's end): flag | from << 8 | to << 16
// (from, to <= 128), bit 31 = GEN: the tables cannot answer this row (it ended in the overlap state of a bordered prefix literal)
#define FX_SPAN_EXC 0x80000000u

// One row in LDS: chunks C0 .. C0 + NCH - 1 of lane r's cells.  The half-row kernel's lean loops (fx_tile.hpp, FX_HALF4): running maximum
// instead of the group's eight states, one chunk of LDS prefetch, the forward window's lookups 16 at a time.
template <int RL, bool SPANS, int SCH, class TabT>
__device__ __forceinline__ uint32_t fx_span_scan_row(const uint4* tile, const uint8_t* tb, const uint8_t* eor, const uint32_t lane, const uint32_t c0,
                                                      const TabT* __restrict__ tabR, const TabT* __restrict__ tabA, const uint8_t* TRp, const uint8_t* TAp,
                                                      const FastParams& fp, uint32_t& na) {
   using F = typename FxF<SCH>::type;
   constexpr int NCH = FxSpan<RL>::NCH;
   uint32_t state = fp.R_start;
   uint32_t gsel = 0xFFFFFFFFu, esel = 0;   // leftmost 8-byte group holding a hit, and the state entering it
   {
      F fa[8], fb[8];
      uint4 wk = tile[tile_cell(lane, c0 + (uint32_t)NCH - 1u)];
      lookup8(fa, wk.z, wk.w, tabR);
#pragma unroll
      for (int k = NCH - 1; k >= 0; --k) {
         na |= wk.x | wk.y | wk.z | wk.w;
         lookup8(fb, wk.x, wk.y, tabR);
         __builtin_amdgcn_sched_barrier(0);
         {
            const uint32_t entry = state;
            const uint32_t mx = chain8_back<F, true>(fa, state, TRp);
            gsel = mx >= fp.hit_min ? (uint32_t)(2 * k + 1) : gsel;
            esel = mx >= fp.hit_min ? entry : esel;
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
            const uint32_t mx = chain8_back<F, true>(fb, state, TRp);
            gsel = mx >= fp.hit_min ? (uint32_t)(2 * k) : gsel;
            esel = mx >= fp.hit_min ? entry : esel;
            asm volatile("" : "+v"(esel));
         }
         __builtin_amdgcn_sched_barrier(0);
      }
   }
   const bool hit = gsel != 0xFFFFFFFFu;
   // exact byte of the leftmost hit: re-walk the selected group (skipped when no lane has one)
   uint32_t s = 0;   // wrapped start index (1 = leading NUL, j + 2 for text byte j), 0 = none
   if (__builtin_amdgcn_ballot_w64(hit) != 0) {
      const uint32_t g = hit ? gsel : 0u;
      const uint2 rw = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, c0 + (g >> 1)) << 4) + ((g & 1u) << 3));
      F f[8];
      lookup8(f, rw.x, rw.y, tabR);
      uint32_t st = esel, loc = 8;
#pragma unroll
      for (int i = 7; i >= 0; --i) {
         st = fxstep(f[i], st, TRp);
         loc = st >= fp.hit_min ? (uint32_t)i : loc;
      }
      s = hit ? g * 8u + 2u + loc : 0u;
   }
   {
      const F fz = tabR[0];   // leading NUL: a hit there is the leftmost start
      state = fxstep(fz, state, TRp);
   }
   s = state >= fp.hit_min ? 1u : s;
   const bool except = fp.inv_on != 0u && state == fp.inv;   // (bordered prefix literal: the general procedure's candidate list decides)
   // ---- left-to-right pass from the leftmost start: anchored DFA, longest accept (api_internal_m.F90:119-148) ----
   // flags only: a start inside the text always gives to >= from >= 1, so only starts at the leading NUL need the walk
   uint32_t cur = (s != 0u && (SPANS || s == 1u) && !except) ? fp.A_init : 0u;
   uint32_t mm = 0;                          // max_match (wrapped index of the byte after the match)
   uint32_t j = s >= 2u ? s - 2u : 0u;       // 0-based text index of the next byte to consume
   if (__builtin_amdgcn_ballot_w64(s == 1u) != 0) {
      const F f = tabA[0];
      const uint32_t nx = fxstep(f, cur, TAp);
      mm = (s == 1u && nx >= fp.acc_min) ? 2u : 0u;
      cur = s == 1u ? nx : cur;
   }
   if (__builtin_amdgcn_ballot_w64(cur != 0u) != 0) {
      // first 32 symbols straight-line: five aligned 8-byte reads, a byte shift to start exactly at j; per 8-byte group only "any accept"
      // + entry state are kept and the last accepting group is re-walked for the exact byte
      uint32_t o[8];
      {
         const uint32_t base = j & ~7u, sh = j & 7u;
         uint32_t d[10];
#pragma unroll
         for (int g = 0; g < 5; ++g) fx_span_group<RL>(d[2 * g], d[2 * g + 1], tb, eor, lane, c0, base + 8u * (uint32_t)g);
         const uint32_t up = 0u - ((sh >> 2) & 1u);   // all ones when the stream starts in the odd dword
         uint32_t e[9];
#pragma unroll
         for (int k = 0; k < 9; ++k) e[k] = (up & d[k + 1]) | (~up & d[k]);
#pragma unroll
         for (int k = 0; k < 8; ++k) o[k] = __builtin_amdgcn_alignbyte(e[k + 1], e[k], sh & 3u);
      }
      constexpr int GB = 2;
      uint32_t gl = 0xFFFFFFFFu, el = 0, blo = 0, bhi = 0;
#pragma unroll
      for (int gb = 0; gb < 4; gb += GB) {
         F f[8 * GB];
#pragma unroll
         for (int g = 0; g < GB; ++g) lookup8(&f[8 * g], o[2 * (gb + g)], o[2 * (gb + g) + 1], tabA);
#pragma unroll
         for (int g = 0; g < GB; ++g) {
            const uint32_t entry = cur;
            uint32_t st[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
               cur = fxstep(f[8 * g + q], cur, TAp);
               st[q] = cur;
            }
            const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
            const bool acc = mx >= fp.acc_min;
            gl = acc ? (uint32_t)(gb + g) : gl;
            el = acc ? entry : el;
            blo = acc ? o[2 * (gb + g)] : blo;
            bhi = acc ? o[2 * (gb + g) + 1] : bhi;
         }
      }
      {
         F fr8[8];
         lookup8(fr8, blo, bhi, tabA);
         uint32_t st = el, loc = 0;
#pragma unroll
         for (int q = 0; q < 8; ++q) {
            st = fxstep(fr8[q], st, TAp);
            loc = st >= fp.acc_min ? (uint32_t)q : loc;
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
         fx_span_group<RL>(t0[0], t0[1], tb, eor, lane, c0, gb);
         fx_span_group<RL>(t1[0], t1[1], tb, eor, lane, c0, gb + 8u);
         do {
            uint32_t t2[2];
            fx_span_group<RL>(t2[0], t2[1], tb, eor, lane, c0, gb + 16u);
            const uint32_t e0 = (up & t0[1]) | (~up & t0[0]), e1 = (up & t1[0]) | (~up & t0[1]), e2 = (up & t1[1]) | (~up & t1[0]);
            const uint32_t o0 = __builtin_amdgcn_alignbyte(e1, e0, sh & 3u), o1 = __builtin_amdgcn_alignbyte(e2, e1, sh & 3u);
            F f8[8];
            lookup8(f8, o0, o1, tabA);
            uint32_t loc = 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
               cur = fxstep(f8[q], cur, TAp);
               loc = cur >= fp.acc_min ? (uint32_t)q : loc;
            }
            mm = loc != 8u ? j + loc + 3u : mm;
            j += 8u;
            gb += 8u;
            t0[0] = t1[0]; t0[1] = t1[1];
            t1[0] = t2[0]; t1[1] = t2[1];
         } while (__builtin_amdgcn_ballot_w64(cur != 0u) != 0);
      }
   }
   uint32_t out = except ? FX_SPAN_EXC : 0u;
   if (SPANS) {
      if (s != 0u && mm != 0u) {   // api_internal_m.F90:140-148
         const uint32_t fr = s >= 2u ? s - 1u : 1u;
         const uint32_t tt = mm >= (uint32_t)RL + 2u ? (uint32_t)RL : mm - 2u;
         if (mm > 2u) out |= 1u | (fr << 8) | (tt << 16);
      }
   } else {
      out |= (s >= 2u || (s == 1u && mm > 2u)) ? 1u : 0u;
   }
   return out;
}

// n_deferred: this call's group of four counter words (words of consecutive calls alternate; [0] "tiles were deferred", [2] / [3] the
// sample FX_ADAPT_CALLS describes); unused by the GEN instantiations, which leave nothing behind.
template <int RL, bool SPANS, int SCH, bool GEN>
__global__ __launch_bounds__(256, 4) void fx_search_span(const uint8_t* __restrict__ rows, const int64_t n, const uint8_t* __restrict__ prog, const FastParams fp,
                                                          uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                          uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ clear_next) {
   static_assert(SCH == 0 || SCH == 2, "class-level v_perm or nibble tables");
   using S = FxSpan<RL>;
   using F = typename FxF<SCH>::type;
   constexpr int K = S::K, RH = S::RH, NCH = S::NCH;
   if (!GEN && blockIdx.x == 0 && threadIdx.x == 0) {   // (a first pass of the multi-pass kind: it zeroes the next call's counter group)
      clear_next[0] = 0u;
      clear_next[1] = 0u;
      clear_next[2] = 0u;
      clear_next[3] = 0u;
   }
   __shared__ F tabR_s[256];
   __shared__ F tabA_s[256];
   __shared__ __attribute__((aligned(16))) uint4 tiles[4 * 512 + 4];   // 4 waves x 64 spans x 8 cells, then the four shared end-of-row cells
   __shared__ uint32_t exc_q[GEN ? 4 * 64 : 1];                        // GEN: per-wave queues of rows for the general procedure
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int64_t total = n * (int64_t)RL;
   const int64_t n_tiles = (total + 16383) >> 14;
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   // first passes whose tiles mostly hold UTF-8 (FX_ADAPT_CALLS, fx_tile.hpp): the follow-up's persistent word says "skip the loads"
   const bool adapt = !GEN && (fp.defer_tiles & 2u) != 0u;
   if (adapt) {
      const uint32_t* hintw = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(n_deferred) & ~uintptr_t(31)) + 32u);
      if (__builtin_amdgcn_readfirstlane(hintw[0]) != 0u) {
         for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
            const int64_t r0 = ((t << 6) + lane) * K;
#pragma unroll
            for (int i = 0; i < K; ++i)
               if (r0 + i < n) flags[r0 + i] = FX_NEEDS_GENERAL;
         }
         if (lane == 0) n_deferred[0] = 1u;
         return;
      }
   }
   // start-up: the table entries are READ first, then the first half's loads go out, and only then are the entries written to LDS
   const uint2 t_r = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16R : h->off_fastR))[threadIdx.x];
   const uint2 t_a = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16A : h->off_fastA))[threadIdx.x];
   __builtin_amdgcn_sched_barrier(0);
   uint4 stage[8];
   fx_span_load(stage, rows, total, wave_global, 1u, lane, true);
   reinterpret_cast<uint2*>(tabR_s)[threadIdx.x] = t_r;
   reinterpret_cast<uint2*>(tabA_s)[threadIdx.x] = t_a;
   uint4* const tile = tiles + wave * 512u;
   uint4* const eor_cell = tiles + 4 * 512 + wave;
   if (lane == 0) *eor_cell = make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
   __syncthreads();
   const F* tabR = tabR_s;
   const F* tabA = tabA_s;
   const uint8_t* const tb = reinterpret_cast<const uint8_t*>(tile);
   const uint8_t* const eor = reinterpret_cast<const uint8_t*>(eor_cell);
   const bool utf8 = !GEN && fp.defer_tiles != 0u;   // tiles holding a byte >= 0x80 are deferred whole to the follow-up
   bool any_deferred = false;
   uint32_t n_def = 0, n_seen = 0;
   uint32_t* const myq = exc_q + (GEN ? wave * 64u : 0u);
   uint32_t q_n = 0;   // rows in this wave's queue (wave-uniform)
   // GEN: rows the tile pass could not queue yet (wave-uniform masks, one per row slot of the lanes; the rows are pend_row0 + slot)
   uint64_t pend_m[K];
#pragma unroll
   for (int i = 0; i < K; ++i) pend_m[i] = 0;
   uint32_t pend_row0 = 0;
   for (int64_t t = wave_global;;) {
      if constexpr (GEN) {
         // One iteration is a tile or -- at ONE place in the code -- a drain of the queue: every lane below q_n walks one queued row with the
         // general row procedure, from global memory (fxrow::run_row: candidate-list driver, UTF-8 decode in both directions).
         bool more = false;
#pragma unroll
         for (int i = 0; i < K; ++i) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(pend_m[i]);
            if (cnt != 0u && q_n + cnt <= 64u) {
               if ((pend_m[i] >> lane) & 1ull) myq[q_n + (uint32_t)__builtin_popcountll(pend_m[i] & ((1ull << lane) - 1ull))] = pend_row0 + (uint32_t)i;
               q_n += cnt;
               pend_m[i] = 0;
            }
            more = more || pend_m[i] != 0;
         }
         const bool at_end = t >= n_tiles;
         if (at_end && !more && q_n == 0u) break;
         if (more || at_end) {
            if (lane < q_n) {
               const int64_t row = (int64_t)myq[lane];
               fxrow::ProgView pv(prog);
               fxrow::DfaSim sim(pv);
               fxrow::Result rr;
               FxSpanGlobalRow gr{rows + row * (int64_t)RL};
               fxrow::run_row(pv, sim, gr, RL, rr);
               flags[row] = (uint8_t)rr.flag;
               if (SPANS) {
                  from[row] = rr.from;
                  to[row] = rr.to;
               }
            }
            q_n = 0;
            continue;
         }
      } else if (t >= n_tiles) break;
      const int64_t t_next = t + wave_stride;
      const int64_t row_first = ((t << 6) + lane) * K;
      n_seen += 1u;
      uint32_t res[K];   // flag | from << 8 | to << 16 | FX_SPAN_EXC
#pragma unroll
      for (int i = 0; i < K; ++i) res[i] = 0u;
      uint32_t na = 0;
      bool defer_early = false;
      // the two halves, right one first: a ROLLED loop, so that the staging registers are reloaded at ONE place in the code (a second load
      // site meets the first in a register merge at the back edge: copies behind a vmcnt(0), fx_tile.hpp)
#pragma unroll 1
      for (uint32_t hf = 1u;; --hf) {
         if (hf == 1u && utf8) {
            // cheap sampled look at the staged bytes: a tile that shows a byte >= 0x80 here is deferred without being scanned
            const uint32_t smp = stage[0].x | stage[0].w | stage[4].y | stage[7].z;
            defer_early = __builtin_amdgcn_ballot_w64((smp & 0x80808080u) != 0) != 0;
         }
         store_tile<8>(stage, tile, lane);
         // the left half of this tile, or the right half of the next
         const bool last = hf == 0u || defer_early;
         fx_span_load(stage, rows, total, last ? t_next : t, last ? 1u : 0u, lane, true);
         if (defer_early) break;
         uint32_t hr[RH];
#pragma unroll
         for (int jr = RH - 1; jr >= 0; --jr) hr[jr] = fx_span_scan_row<RL, SPANS, SCH>(tile, tb, eor, lane, (uint32_t)(jr * NCH), tabR, tabA, nullptr, nullptr, fp, na);
         if (hf == 1u) {   // (wave-uniform)
#pragma unroll
            for (int i = 0; i < RH; ++i) res[RH + i] = hr[i];
         } else {
#pragma unroll
            for (int i = 0; i < RH; ++i) res[i] = hr[i];
            break;
         }
      }
      // bytes >= 0x80: with decode tables the whole TILE goes to the follow-up (wave-uniform); without them (GEN) just those ROWS are
      // queued for the general procedure -- as are rows that ended in the overlap state
      bool defer_tile = defer_early;
      if (!defer_early && !GEN && utf8) defer_tile = __builtin_amdgcn_ballot_w64((na & 0x80808080u) != 0u) != 0;
      if (defer_tile) {
#pragma unroll
         for (int i = 0; i < K; ++i) res[i] = FX_NEEDS_GENERAL;
         any_deferred = true;
         n_def += 1u;
      }
      bool lane_exc = false;
      if constexpr (GEN) {
         // (the OR is the lane's: one row with a byte >= 0x80 sends the lane's K rows to the queue -- the general procedure answers any row)
         const bool hi = (na & 0x80808080u) != 0u;
#pragma unroll
         for (int i = 0; i < K; ++i) {
            if (hi) res[i] |= FX_SPAN_EXC;
            if (row_first + i >= n) res[i] &= ~FX_SPAN_EXC;
            lane_exc = lane_exc || (res[i] & FX_SPAN_EXC) != 0u;
         }
      }
      // ---- results: the lane's K consecutive rows, one store per array when all of them exist (and none is left to the queue) ----
      auto fl = [&](int i) -> uint32_t { return res[i] & 0xFFu; };
      auto fr = [&](int i) -> int32_t { return (int32_t)((res[i] >> 8) & 0xFFu); };
      auto tt = [&](int i) -> int32_t { return (int32_t)((res[i] >> 16) & 0xFFu); };
      if (row_first + K <= n && !lane_exc) {
         if constexpr (K == 2) {
            *reinterpret_cast<uint16_t*>(flags + row_first) = (uint16_t)(fl(0) | (fl(1) << 8));
            if (SPANS && !defer_tile) {
               *reinterpret_cast<int2*>(from + row_first) = make_int2(fr(0), fr(1));
               *reinterpret_cast<int2*>(to + row_first) = make_int2(tt(0), tt(1));
            }
         } else if constexpr (K == 4) {
            *reinterpret_cast<uint32_t*>(flags + row_first) = fl(0) | (fl(1) << 8) | (fl(2) << 16) | (fl(3) << 24);
            if (SPANS && !defer_tile) {
               *reinterpret_cast<int4*>(from + row_first) = make_int4(fr(0), fr(1), fr(2), fr(3));
               *reinterpret_cast<int4*>(to + row_first) = make_int4(tt(0), tt(1), tt(2), tt(3));
            }
         } else {
            *reinterpret_cast<uint2*>(flags + row_first) = make_uint2(fl(0) | (fl(1) << 8) | (fl(2) << 16) | (fl(3) << 24), fl(4) | (fl(5) << 8) | (fl(6) << 16) | (fl(7) << 24));
            if (SPANS && !defer_tile) {
               *reinterpret_cast<int4*>(from + row_first) = make_int4(fr(0), fr(1), fr(2), fr(3));
               *reinterpret_cast<int4*>(from + row_first + 4) = make_int4(fr(4), fr(5), fr(6), fr(7));
               *reinterpret_cast<int4*>(to + row_first) = make_int4(tt(0), tt(1), tt(2), tt(3));
               *reinterpret_cast<int4*>(to + row_first + 4) = make_int4(tt(4), tt(5), tt(6), tt(7));
            }
         }
      } else {
#pragma unroll
         for (int i = 0; i < K; ++i)
            if (row_first + i < n && !(GEN && (res[i] & FX_SPAN_EXC) != 0u)) {
               flags[row_first + i] = (uint8_t)fl(i);
               if (SPANS && !defer_tile) {
                  from[row_first + i] = fr(i);
                  to[row_first + i] = tt(i);
               }
            }
      }
      if constexpr (GEN) {   // the excepted rows, slot by slot of the lanes' K: queued at the top of the next iteration
#pragma unroll
         for (int i = 0; i < K; ++i) pend_m[i] = __builtin_amdgcn_ballot_w64((res[i] & FX_SPAN_EXC) != 0u);
         pend_row0 = (uint32_t)row_first;
      }
      t = t_next;
   }
   if (!GEN) {
      // one plain store per wave (the value only gates the follow-up); the sample of FX_ADAPT_CALLS: two atomics from every 256th wave
      if (any_deferred && lane == 0) n_deferred[0] = 1u;
      if (adapt && (wave_global & 255) == 0 && lane == 0) {
         atomicAdd(&n_deferred[3], n_seen);
         if (n_def != 0u) atomicAdd(&n_deferred[2], n_def);
      }
   }
}

// ctr: this call's counter group (GEN: unused)
template <int RL, int SCH, bool GEN>
hipError_t launch_span(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, int32_t* from, int32_t* to, uint32_t* ctr,
                       hipStream_t st) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ctr) ^ 16u);   // the other parity's group of four words
   const int64_t total = n * (int64_t)RL;
   const int64_t n_tiles = (total + 16383) >> 14;
   int64_t blocks = (n_tiles + 3) / 4;
   // whole rounds of the four resident blocks per CU, one per 225 MB of rows, at least three (the half-row kernel's rule: fx_tile.hpp)
   int64_t rounds = fx_env().half_rounds;
   if (rounds <= 0) {
      rounds = total / ((int64_t)225 << 20);
      if (rounds < 3) rounds = 3;
      if (rounds > 64) rounds = 64;
   }
   if (blocks > (int64_t)256 * 4 * rounds) blocks = (int64_t)256 * 4 * rounds;
   const int env_blocks = fx_env().one_blocks;   // FXAMD_ONE_BLOCKS, test hook: a tiny grid, many tiles per wave (queue overflow mid-loop)
   if (env_blocks > 0 && blocks > env_blocks) blocks = env_blocks;
   if (blocks < 1) blocks = 1;
   const bool spans = from && to;
   if (spans) hipLaunchKernelGGL((fx_search_span<RL, true, SCH, GEN>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, from, to, ctr, clear_next);
   else hipLaunchKernelGGL((fx_search_span<RL, false, SCH, GEN>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, from, to, ctr, clear_next);
   return hipGetLastError();
}
#define FX_SPAN_SIG (const uint8_t*, int64_t, const uint8_t*, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t*, hipStream_t)
