// fx_search_one: a `.in.` / regex search -- or, with MATCH, a `.match.` -- over rows of up to 256 bytes in ONE launch.
//
// fx_search_fast needs up to three launches per call when rows may hold bytes >= 0x80 (first pass that DEFERS such tiles, a second
// pass over them, a third over the exception rows), reads deferred tiles twice and pays for the empty gate launches on pure-ASCII
// batches.  Here every tile is finished by the wave that staged it, with the scheme its bytes ask for:
//   * pure-ASCII tile            -> class-level tables (v_perm / nibble / chain), exactly the first pass of fx_search_fast
//   * tile with a byte >= 0x80   -> byte-level tables (FXP_F_BYTE_DFA; chain or nibble format) on the RAW bytes of the same LDS tile,
//                                   or, without them (ragged rows, programs without byte-level automata), the in-LDS UTF-8
//                                   decode (fxrow::translate_cell16) followed by the class-level scan
//   * exception rows of the byte-level tables (structurally invalid UTF-8: the backward pass ends in the INVALID state) are kept
//     in a per-wave queue in LDS (64 row indices); when a tile would overflow it, and once more when the wave has finished its
//     tiles, the queued rows are GATHERED into a tile (lane r loads row queue[r] into its own cells), decoded in LDS and
//     scanned with the class-level tables -- the decode pass of fx_search_fast's MODE 4, inside the same launch.
// Programs whose class-level tables cannot decode UTF-8 (candidate-list driver programs) queue the rows their tables cannot answer
// the same way and walk them with the general row procedure (GEN).  Rows longer than 256 bytes stay on fx_search_fast / fx_match_fast
// and their multi-pass pipeline; 256-byte rows on the 8-state tables take fx_search_fast's half-row first pass and this kernel (MARKED)
// as its one gated follow-up.
// Round 3: `.match.` (fx_match_tile), match compaction for short rows (DEFERQ: the exact start + forward pass of sparse tiles' hit rows
// are finished 64 at a time from global memory), the aligned forward walk for long matches, byte-level tables whose forward
// automaton is in the v_perm format (BSCH 3, FXP_F_BYTE_A8); then: lookups issued TWO chains ahead where a group's work is one short chain
// (the aligned forward walk, `.match.` on the 8-state tables: three lookup buffers in rolled trips of three chunks), the chunk-parallel scan
// of the few exception rows a wave ends with (fx_few.hpp), and a start-up that overlaps the tables (L2) with the first tile (HBM).
#pragma once
#include "fx_tile.hpp"
#include "fx_few.hpp"
#ifndef FX_FEW_ROWS
#define FX_FEW_ROWS 1   // gathered tiles of a few exception rows: chunk-parallel scan (fx_few.hpp) instead of one lane per row
#endif

// row accessors of the general procedure: the row in global memory / in lane r's cells of the LDS tile
struct FxGlobalRow {
   const uint8_t* p;
   __device__ __forceinline__ uint32_t operator[](int j) const { return p[j]; }
};
struct FxTileRow {
   const uint8_t* tb;
   uint32_t lane;
   __device__ __forceinline__ uint32_t operator[](int j) const {
      const uint32_t k = (uint32_t)j >> 4;
      return tb[(tile_cell(lane, k) << 4) + ((uint32_t)j & 15u)];
   }
};

// SCH_: scheme of the backward (and, unless SA_ says otherwise, the forward) tables; SA_: scheme of the FORWARD tables (byte-level tables
// with FXP_F_BYTE_A8: nibble tables backwards, 8-state v_perm tables forwards)
template <int SCH_, bool BYTES_, bool DECODED_, int SA_ = SCH_>
struct FxScanCfg {
   static constexpr int sch = SCH_, sch_a = SA_;
   static constexpr bool bytes = BYTES_, decoded = DECODED_;
};

// ---- one scan of a tile in LDS: backward pass (leftmost start), forward pass (longest end), results -----------------------------
// S_: table scheme; BYTES: byte-level tables on raw bytes; DECODED: the tile was rewritten into symbol ids (every byte value means
// something: no byte >= 0x80 test).  A class-level scan of RAW bytes that meets a byte >= 0x80 either gives the whole tile back
// (REDO_TILE: returns true, nothing written -- the caller redoes it with byte-level tables or after a decode) or marks just those
// rows (ROW_EXC: `except`, as for rows ending in the overlap state of a bordered prefix literal); `except` is also set for rows whose
// byte-level backward pass ends in the INVALID state.  Excepted rows are not emitted.  PREPAD: ragged rows were padded and their
// bytes OR-ed by the caller (several scans of one tile: fx_search_multi).
struct FxScanCtx {
   uint4* tile;
   const uint8_t* tb;
   uint32_t lane, L, Lr;
   bool whole, raw;
   uint32_t pre_na;
   const FxTail* tl = nullptr;   // TAIL scans (ragged rows of fx_search_one, fx_tile.hpp "Ragged rows, round 4"): L = Lr, the cells are walked as 16*CH bytes
   const uint8_t* pfx = nullptr;   // FXP_F_PREFIX_CHECK programs (round 6): the prefix literal (global memory, wave-uniform reads) ...
   uint32_t pfx_len = 0;           // ... and its length; 0 = no per-row check (fxrow::prefix_start_ok, row_engine.hpp)
   const uint8_t* sfx = nullptr;   // FXP_F_SUFFIX_CHECK: the suffix literal ...
   uint32_t sfx_len = 0;           // ... and its length; 0 = no check of the match's end (fxrow::suffix_end_ok)
};
// Match compaction (DEFERQ): the exact start and the forward pass are per-ROW work that only rows with a hit need, but a
// wave pays for them per TILE -- at full price when a few lanes in 64 have a hit (config 2: one row in ten matches).  Unless the tile
// is dense in hits (more than FX_DEFER_DENSE of 64 rows), those rows are queued per wave (row, hit group, state entering it) and
// finished 64 at a time from global memory (fx_finish_from_global); their flag is known at once (a start inside the text always
// yields a span), only from / to follow at the flush.
#ifndef FX_FWD_ALIGNED
#define FX_FWD_ALIGNED 1      // long matches on many lanes: the forward pass continues in aligned 8-byte groups (fx_scan_tile)
#endif
#ifndef FX_FWD_ALIGNED_MIN
#define FX_FWD_ALIGNED_MIN 12  // lanes still walking after the first window for that to pay
#endif
#ifndef FX_FWD_PIPE3
#define FX_FWD_PIPE3 1        // the aligned forward walk with three lookup buffers (lookups two chains ahead) instead of two
#endif
#ifndef FX_MATCH_PIPE3
#define FX_MATCH_PIPE3 1      // `.match.` on the 8-state tables, rows of 96 bytes and longer: three lookup buffers (lookups two chains ahead)
#endif
#ifndef FX_MATCH_P3_MINCH
#define FX_MATCH_P3_MINCH 6   // rows of 96 bytes and longer (at least two trips of three chunks)
#endif
#ifndef FX_FWD_DIRECT
#define FX_FWD_DIRECT 1       // clustered starts on many lanes: straight into the aligned walk, no 32-symbol window
#endif
#ifndef FX_ONE_MINW
#define FX_ONE_MINW 1   // minimum waves per SIMD fx_search_one is compiled for (experiment hook: tools/ru_one.sh).  The built-in rule: THREE
                        // for rows of 96 / 128 bytes -- three blocks fit a CU by LDS, the variants use 139-169 VGPRs and the step from 168 to 169
                        // costs a wave per SIMD (config 5: 0.36 -> 0.42 ms when one more live value crossed it); every variant fits without scratch
#endif
#ifndef FX_SPEC_FWD
#define FX_SPEC_FWD 1         // speculative forward pass from the row's first character (fx_spec_forward; programs with FXP_F_SPEC_FWD)
#endif
#ifndef FX_SPEC_FAIL_MAX
#define FX_SPEC_FAIL_MAX 24   // rows of a tile (of 64) that may fail it and be queued; beyond that the tile is scanned in place and the pass pauses
#endif
#ifndef FX_SPEC_RETRY
#define FX_SPEC_RETRY 16      // ... for this many tiles of the wave
#endif
#ifndef FX_SPEC_FEW_GROUPS
#define FX_SPEC_FEW_GROUPS 3   // gathered tiles of up to this many groups of 64 / CH rows take the in-LDS decode + the chunk-parallel scan of fx_few.hpp (round 5: one group,
                               // exception rows only); rows the speculative walk could not answer go there directly, without the byte-level scan in between
#endif
#ifndef FX_DEFER_DENSE
#define FX_DEFER_DENSE 12   // tiles with more hit rows than this finish them in place (config 3 / 5: half of the rows match -- queueing those costs
                            // scattered from / to stores and a second read of the rows' bytes: measured slower, profiles/r03_defer_ab.txt)
#endif
#ifndef FX_DEFER_STASH
#define FX_DEFER_STASH 0   // (1: match compaction copies a queued row's 24 bytes from its hit group on into the queue while the row is in LDS, so that the
                           //  flush reads no global memory -- measured and NOT kept: config 2 18.2-18.6 -> 18.6-18.9 us per step, gpurun call r05_c27: three
                           //  more LDS reads + predicated stores per tile cost more than the flush's round trip at the end of a wave)
#endif
struct FxFwdQueue {
   uint32_t* q;        // LDS: 64 row numbers, then 64 x (hit group | entry state << 16)
   uint32_t n;         // entries (wave-uniform)
   uint32_t family;    // table family of the queued entries: 0 class-level, 1 byte-level (wave-uniform)
   uint2* w;           // LDS (FX_DEFER_STASH): 3 x 64 8-byte groups -- group k of slot s at w[64 k + s]: the flush's re-walk + first window read no global memory
   const uint8_t* rows = nullptr;   // FX_DEFER_PREFETCH: the batch (rows of L bytes) ...
   uint32_t pre[6] = {0, 0, 0, 0, 0, 0};   // ... and, per LANE, the three 8-byte groups of the row in the slot this lane will finish (registers)
};
struct FxNoFlush {
   __device__ __forceinline__ void operator()() const {}
};
template <int CH, bool SPANS, bool RAGGED, int S_, bool BYTES, bool DECODED, bool REDO_TILE, bool ROW_EXC, bool PREPAD, bool DEFERQ = false, int S_A = S_,
          bool TAIL = false, class TabT, class TabTA, class Emit, class Flush = FxNoFlush>
__device__ __forceinline__ bool fx_scan_tile(const FxScanCtx& c, const TabT* __restrict__ tabR, const TabTA* __restrict__ tabA, const uint8_t* TRp,
                                             const uint8_t* TAp, const FastParams& P, const int64_t row, const bool row_ok, const bool ordered,
                                             bool& except, Emit& emit, FxFwdQueue* fq = nullptr, Flush flush = Flush()) {
   {
      constexpr bool CHAIN = S_ == 1, WIDE = S_ == 2;
      (void)CHAIN;
      (void)WIDE;
      using F = typename FxF<S_>::type;      // backward tables (R)
      using FA = typename FxF<S_A>::type;    // forward tables (A)
      uint4* const tile = c.tile;
      const uint8_t* const tb = c.tb;
      const uint32_t lane = c.lane, L = c.L, Lr = c.Lr;
      const bool whole = c.whole, raw = c.raw;
      (void)Lr;
      (void)whole;
      static_assert(!TAIL || (!RAGGED && !PREPAD && !DEFERQ), "TAIL: the round-4 ragged scheme (no pad symbol, no match compaction)");
      const uint32_t Lc = TAIL ? 16u * (uint32_t)CH : L;   // what the left-to-right fetches address: the cells (TAIL: the NUL and KILL symbols sit behind the text)
      uint32_t state = P.R_start;
      uint32_t gsel = 0xFFFFFFFFu, esel = 0;   // leftmost 8-byte group holding a hit, and the state entering it
      uint32_t na = 0;
      if (PREPAD) na = c.pre_na;
      else if (RAGGED && (!whole || DECODED)) na |= pad_rows<CH>(tile, lane, Lr);
      if constexpr (TAIL) {
         // right-to-left from the row's LAST byte: the chunk the row ends in over its valid bytes, then the whole chunks; same two
         // lookup buffers as below
         const FxTail T = fx_tail_here(*c.tl);
         F fa[8], fb[8];
         if (T.nb != 0u) {
            const uint4 wp = tile[tile_cell(lane, T.kt)];
            if (!DECODED) na |= fx_tail_or(wp, T.nb);
            if (T.nb > 8u) {
               lookup8(fa, wp.z, wp.w, tabR);
               const uint32_t entry = state;
               const uint32_t mx = chain8_back_n(fa, state, TRp, T.nb - 8u);
               gsel = mx >= P.hit_min ? 2u * T.kt + 1u : gsel;
               esel = mx >= P.hit_min ? entry : esel;
            }
            lookup8(fb, wp.x, wp.y, tabR);
            const uint32_t entry = state;
            const uint32_t nv0 = T.nb < 8u ? T.nb : 8u;
            const uint32_t mx = nv0 == 8u ? chain8_back(fb, state, TRp) : chain8_back_n(fb, state, TRp, nv0);
            gsel = mx >= P.hit_min ? 2u * T.kt : gsel;
            esel = mx >= P.hit_min ? entry : esel;
         }
         if (T.kt != 0u) {
            // whole chunks kt - 1 .. 0: the aligned kernels' unrolled loop (cell addresses are immediates) behind wave-uniform guards;
            // only the first chunk's words are read through a run-time address
            uint4 wk = tile[tile_cell(lane, T.kt - 1u)], wn = tile[tile_cell(lane, T.kt >= 2u ? T.kt - 2u : 0u)];
            lookup8(fa, wk.z, wk.w, tabR);
#pragma unroll
            for (int k = CH - 2; k >= 0; --k) {   // (kt <= CH - 1: a ragged row is shorter than 16 * CH bytes)
               if ((uint32_t)k < T.kt) {
                  if (!DECODED) na |= wk.x | wk.y | wk.z | wk.w;
                  lookup8(fb, wk.x, wk.y, tabR);
                  __builtin_amdgcn_sched_barrier(0);
                  {
                     const uint32_t entry = state;
                     const uint32_t mx = chain8_back(fa, state, TRp);
                     gsel = mx >= P.hit_min ? (uint32_t)(2 * k + 1) : gsel;
                     esel = mx >= P.hit_min ? entry : esel;
                     asm volatile("" : "+v"(esel));
                  }
                  __builtin_amdgcn_sched_barrier(0);
                  if (k >= 1) {
                     wk = wn;
                     lookup8(fa, wk.z, wk.w, tabR);
                     if (k >= 2) wn = tile[tile_cell(lane, k - 2)];
                  }
                  __builtin_amdgcn_sched_barrier(0);
                  {
                     const uint32_t entry = state;
                     const uint32_t mx = chain8_back(fb, state, TRp);
                     gsel = mx >= P.hit_min ? (uint32_t)(2 * k) : gsel;
                     esel = mx >= P.hit_min ? entry : esel;
                     asm volatile("" : "+v"(esel));
                  }
                  __builtin_amdgcn_sched_barrier(0);
               }
            }
         }
      } else
      // (Three lookup buffers -- a group's lookups two chains ahead, as in the forward walk and in fx_match_tile below -- do NOT pay here: with the
      //  max tree and the selects a group's chain is ~60 cycles and the two-buffer distance already ~125, an LDS round trip; measured in rolled
      //  trips: config 5 +1.3 %, 256-byte rows on this kernel +2.7 %; fully unrolled: +38 % -- profiles/r03_pipe3_ab.txt.)
      {
         F fa[8], fb[8];
         uint4 wk = tile[tile_cell(lane, CH - 1)], wn = make_uint4(0, 0, 0, 0);
         if (CH >= 2) wn = tile[tile_cell(lane, CH - 2)];
         lookup8(fa, wk.z, wk.w, tabR);
#pragma unroll
         for (int k = CH - 1; k >= 0; --k) {
            if (!PREPAD && (!RAGGED || (whole && !DECODED && (uint32_t)k < (Lr >> 4)))) na |= wk.x | wk.y | wk.z | wk.w;
            lookup8(fb, wk.x, wk.y, tabR);
            __builtin_amdgcn_sched_barrier(0);
            {
               const uint32_t entry = state;
               const uint32_t mx = chain8_back(fa, state, TRp);
               gsel = mx >= P.hit_min ? (uint32_t)(2 * k + 1) : gsel;
               esel = mx >= P.hit_min ? entry : esel;
               asm volatile("" : "+v"(esel));   // select now: otherwise all 2*CH entry states stay live until after the loop
            }
            __builtin_amdgcn_sched_barrier(0);
            if (k >= 1) {
               wk = wn;
               lookup8(fa, wk.z, wk.w, tabR);
               if (k >= 2) wn = tile[tile_cell(lane, k - 2)];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
               const uint32_t entry = state;
               const uint32_t mx = chain8_back(fb, state, TRp);
               gsel = mx >= P.hit_min ? (uint32_t)(2 * k) : gsel;
               esel = mx >= P.hit_min ? entry : esel;
               asm volatile("" : "+v"(esel));
            }
            __builtin_amdgcn_sched_barrier(0);
         }
      }
      // a class-level scan of raw bytes that met a byte >= 0x80: the tile is redone with the byte-level tables / after a decode;
      // a program that has neither (GEN without byte-level tables) hands just those ROWS to the general procedure
      bool row_hi = false;
      if (!BYTES && !DECODED && !raw) {
         row_hi = (na & 0x80808080u) != 0;
         if (REDO_TILE && __builtin_amdgcn_ballot_w64(row_hi) != 0) return true;
      }
      // leading NUL: a hit there is the leftmost start
      {
         const F fz = tabR[0];
         state = fxstep(fz, state, TRp);
      }
      const bool s_nul = state >= P.hit_min;
      const bool hit = gsel != 0xFFFFFFFFu;
      // byte-level tables: the backward pass ended in the INVALID state -> structurally invalid UTF-8: the row is queued for the decode pass
      // (GEN, class-level tables: a row that ended in the overlap state of a bordered prefix literal, or that holds a byte >= 0x80)
      except = (BYTES && state == P.inv) || (ROW_EXC && !BYTES && !DECODED && ((P.inv_on != 0 && state == P.inv) || row_hi));
      // ---- match compaction: rows with a start inside the text go to the wave's queue unless the tile is dense in them ----
      bool queued = false;
      if constexpr (DEFERQ && SPANS && !DECODED) {
         const bool want = hit && !s_nul && !except && row_ok && P.lit_len == 0 && L >= 16u && c.pfx_len == 0u;   // (prefix-check programs: the exact start decides whether the row is the tables' at all)
         const uint64_t qm = __builtin_amdgcn_ballot_w64(want);
         const uint32_t cnt = (uint32_t)__builtin_popcountll(qm);
         if (cnt != 0u && cnt <= (uint32_t)FX_DEFER_DENSE) {
            if (fq->n + cnt > 64u || (fq->n != 0u && fq->family != (BYTES ? 1u : 0u))) flush();
            queued = want;
            const uint32_t slot = fq->n + (uint32_t)__builtin_popcountll(qm & ((1ull << lane) - 1ull));
            if (queued) {
               fq->q[slot] = (uint32_t)row;
               fq->q[64u + slot] = gsel | ((S_ == 0 ? (esel & 0xFFu) : esel) << 16);
            }
            if constexpr (FX_DEFER_STASH != 0) {
               // Round 5: the three 8-byte groups from the hit group on (the exact start's re-walk + the flush's 16-symbol window; text, then the
               // trailing NUL, then KILL symbols) go into the queue NOW, out of the tile -- the flush used to read them from global memory
               // again: 1.13 x config 2's algorithmic HBM traffic, and a round trip to L2 / HBM at the end of every wave with nothing to overlap it
               const uint32_t p0 = queued ? gsel * 8u : 0u;
#pragma unroll
               for (int q = 0; q < 3; ++q) {
                  uint32_t lo, hi;
                  group_words<false, false>(lo, hi, tb, lane, p0 + 8u * (uint32_t)q, L);
                  if (queued) fq->w[64u * (uint32_t)q + slot] = make_uint2(lo, hi);
               }
            }
            if constexpr (FX_DEFER_PREFETCH != 0 && FX_DEFER_STASH == 0 && CH <= 4) {
               // the lanes that will FINISH the new slots load those rows' three groups now (fx_finish_from_global's own loads, issued one to three tiles early)
               const uint32_t n_old = fq->n;
               if (lane >= n_old && lane < n_old + cnt) {
                  const uint32_t qrow = fq->q[lane], ge = fq->q[64u + lane];
                  const uint8_t* rp = fq->rows + (int64_t)qrow * (int64_t)L;
                  const uint32_t b0 = (ge & 0xFFFFu) * 8u;
#pragma unroll
                  for (int q = 0; q < 3; ++q) group_words<false, true>(fq->pre[2 * q], fq->pre[2 * q + 1], rp, lane, b0 + 8u * (uint32_t)q, L, nullptr, true);
               }
            }
            fq->n += cnt;
            fq->family = BYTES ? 1u : 0u;
         }
      }
      uint32_t s = hit ? 2u : 0u;          // wrapped start index (1 = leading NUL, j+2 for text byte j), 0 = none; 2 stands for "inside the text"
      // exact byte of the leftmost hit: re-walk the selected group -- only spans need it (a verdict is "some start"), and only rows
      // that were not queued
      const bool pfx_on = ROW_EXC && !BYTES && !DECODED && c.pfx_len != 0u;   // (wave-uniform)
      if ((SPANS || pfx_on) && __builtin_amdgcn_ballot_w64(hit && !queued && !s_nul && !except) != 0) {
         const uint32_t g = hit ? gsel : 0u;
         const uint2 rw = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, g >> 1) << 4) + ((g & 1u) << 3));
         F f[8];
         lookup8(f, rw.x, rw.y, tabR);
         uint32_t st = esel, loc = 8;
#pragma unroll
         for (int i = 7; i >= 0; --i) {
            if constexpr (TAIL) {   // the group the row ends in: only its text bytes were walked
               const uint32_t nx = fxstep(f[i], st, TRp);
               const bool in_text = g * 8u + (uint32_t)i < L;
               st = in_text ? nx : st;
               loc = (in_text && nx >= P.hit_min) ? (uint32_t)i : loc;
            } else {
               st = fxstep(f[i], st, TRp);
               loc = st >= P.hit_min ? (uint32_t)i : loc;
            }
         }
         s = hit ? g * 8u + 2u + loc : 0u;
      }
      s = s_nul ? 1u : s;
      bool pfx_nowhere = false;   // (per lane)
      (void)pfx_nowhere;
      if constexpr (ROW_EXC && !BYTES && !DECODED) {
         // FXP_F_PREFIX_CHECK (compile.cpp): the tables searched by brute force; the reference searches its candidate list.  The two agree on this row when the
         // start found is a candidate -- the prefix literal stands there, no earlier occurrence overlaps it; any other row with a hit is the general row
         // procedure's (queued like a row in the overlap state).  Rare programs (0.6 % of generated patterns), byte reads from the tile: not a hot path.
         if (pfx_on) {
            bool ok = true;
            if (s != 0u && !except && row_ok) {
               const FxTileRow tr{tb, lane};
               const uint8_t* const pp = c.pfx;
               auto pre = [&](int k) -> uint32_t { return pp[k]; };
               ok = fxrow::prefix_start_ok(pre, (int)c.pfx_len, tr, (int)L, (int)s);
               if (!ok && !fxrow::prefix_occurs(pre, (int)c.pfx_len, tr, (int)L)) {   // the prefix occurs nowhere: the reference searches by brute force itself, suffix not consulted
                  ok = true;
                  pfx_nowhere = true;
               }
            }
            except = except || !ok;
         }
      }
      // ---- left-to-right pass from the leftmost start: anchored DFA, longest accept (api_internal_m.F90:119-148) ----
      const bool sfx_on = pfx_on && c.sfx_len != 0u;   // (wave-uniform: the match's end is needed whatever the caller asked for)
      uint32_t cur = (s != 0 && !queued && !except && (SPANS || s == 1 || sfx_on) && P.lit_len == 0) ? P.A_init : 0u;
      uint32_t mm = (P.lit_len != 0 && s != 0) ? s + P.lit_len : 0u;   // max_match (wrapped index of the byte after the match)
      uint32_t j = s >= 2 ? s - 2 : 0;      // 0-based text index of the next byte to consume
      if (s == 1) {
         const FA f = tabA[0];
         cur = fxstep(f, cur, TAp);
         mm = cur >= P.acc_min ? 2u : 0u;
      }
      // Long matches on many lanes (config 4: nine rows in ten match from their first character to their last): the row from each lane's
      // start in ALIGNED 8-byte groups with the backward pass's lookup pipeline -- 3.9 instead of 5.3-7.6 instructions per byte.
      constexpr bool ALN = !RAGGED && CH >= 4 && FX_FWD_ALIGNED != 0 && BYTES;   // (byte-level scans: UTF-8 text, where matches run long; the class-level
                                                                                 //  scans of ASCII tiles keep the shorter code: config 5 lost 1 % to its mere presence)
      auto aligned_walk = [&]() {
         // (1) every lane to its next 8-byte boundary: up to 7 symbols, per lane
         const uint32_t nrem = (0u - j) & 7u;
         if (__builtin_amdgcn_ballot_w64(cur != 0 && nrem != 0u) != 0) {
            uint32_t o1[2];
            fetch_groups<RAGGED, 1>(o1, tb, lane, j, Lc);
            FA f8[8];
            lookup8(f8, o1[0], o1[1], tabA);
#pragma unroll
            for (int q = 0; q < 7; ++q) {
               const uint32_t nx = fxstep(f8[q], cur, TAp);
               const bool on = (uint32_t)q < nrem;
               cur = on ? nx : cur;
               mm = (on && nx >= P.acc_min) ? j + (uint32_t)q + 3u : mm;
            }
            j += nrem;
         }
         // (2) aligned groups from this lane's group g0 on; the wave walks chunks c0 .. CH (chunk CH = the end-of-row column:
         //     the trailing NUL, then KILL symbols), a lane joins at its own group; per group only "any accept" + entry state
         const uint32_t g0 = j >> 3;
         uint32_t c0 = 0;
         while (c0 < (uint32_t)CH && __builtin_amdgcn_ballot_w64(cur != 0 && (g0 >> 1) <= c0) == 0) ++c0;
         uint32_t gl2 = 0xFFFFFFFFu, el2 = 0;
#if FX_FWD_PIPE3
         // THREE lookup buffers, each group's lookups issued two chains ahead of its use: with one v_perm_b32 per step a chain is 32 cycles, and
         // lookups issued one chain ahead (the backward pass's scheme, whose nibble chains are three times as long) come back late -- the
         // walk with FEWER instructions ran SLOWER than the 32-symbol window it replaced (profiles/r03_few_ab.txt).  Six groups = three
         // chunks per trip; chunks behind the end-of-row column read it again (KILL symbols: every lane is dead by then).
         // (The same for the NIBBLE backward pass of 192-byte rows -- chains of 24 instructions -- changed nothing: 95.4 -> 95.2 us,
         //  profiles/r03_pipe3_ab.txt.)
         auto cellc = [&](const uint32_t c) { return tile[tile_cell(lane, c <= (uint32_t)CH ? c : (uint32_t)CH)]; };
         // (JOINED: every walking lane has reached its own group -- no per-lane "not yet" selects.  With clustered starts that is so after
         //  the first trip.)
         auto step8 = [&](auto joined, const FA (&f)[8], const uint32_t g) {
            constexpr bool JOINED = decltype(joined)::value;
            const uint32_t entry = cur;
            uint32_t st[8], t = cur;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
               t = fxstep(f[q], t, TAp);
               st[q] = t;
            }
            const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
            const bool act = JOINED || g >= g0;
            cur = act ? t : cur;
            const bool hit = act && mx >= P.acc_min;
            gl2 = hit ? g : gl2;
            el2 = hit ? entry : el2;
         };
         FA fa[8], fb[8], fc[8];
         uint4 w0 = cellc(c0), w1 = cellc(c0 + 1u), w2 = cellc(c0 + 2u);
         lookup8(fa, w0.x, w0.y, tabA);
         lookup8(fb, w0.z, w0.w, tabA);
         uint32_t c = c0;
         auto trip = [&](auto joined) {   // six groups = chunks c, c + 1, c + 2
            lookup8(fc, w1.x, w1.y, tabA);
            __builtin_amdgcn_sched_barrier(0);
            step8(joined, fa, 2u * c);
            __builtin_amdgcn_sched_barrier(0);
            lookup8(fa, w1.z, w1.w, tabA);
            w0 = cellc(c + 3u);
            __builtin_amdgcn_sched_barrier(0);
            step8(joined, fb, 2u * c + 1u);
            __builtin_amdgcn_sched_barrier(0);
            lookup8(fb, w2.x, w2.y, tabA);
            w1 = cellc(c + 4u);
            __builtin_amdgcn_sched_barrier(0);
            step8(joined, fc, 2u * c + 2u);
            __builtin_amdgcn_sched_barrier(0);
            lookup8(fc, w2.z, w2.w, tabA);
            w2 = cellc(c + 5u);
            __builtin_amdgcn_sched_barrier(0);
            step8(joined, fa, 2u * c + 3u);
            __builtin_amdgcn_sched_barrier(0);
            lookup8(fa, w0.x, w0.y, tabA);
            __builtin_amdgcn_sched_barrier(0);
            step8(joined, fb, 2u * c + 4u);
            __builtin_amdgcn_sched_barrier(0);
            lookup8(fb, w0.z, w0.w, tabA);
            __builtin_amdgcn_sched_barrier(0);
            step8(joined, fc, 2u * c + 5u);
            __builtin_amdgcn_sched_barrier(0);
         };
         bool alive = true;
#pragma unroll 1
         for (; c <= (uint32_t)CH; c += 3u) {   // until every walking lane has joined
            trip(std::false_type{});
            alive = __builtin_amdgcn_ballot_w64(cur != 0) != 0;
            if (!alive) break;
            if (__builtin_amdgcn_ballot_w64(cur != 0 && g0 > 2u * (c + 3u)) == 0) {
               c += 3u;
               break;
            }
         }
         if (alive) {
#pragma unroll 1
            for (; c <= (uint32_t)CH; c += 3u) {
               trip(std::true_type{});
               if (__builtin_amdgcn_ballot_w64(cur != 0) == 0) break;
            }
         }
#else
         FA fa[8], fb[8];
         uint4 wk = tile[tile_cell(lane, c0)], wn = tile[tile_cell(lane, c0 < (uint32_t)CH ? c0 + 1u : c0)];
         lookup8(fa, wk.x, wk.y, tabA);
#pragma unroll 1
         for (uint32_t c = c0; c <= (uint32_t)CH; ++c) {
            lookup8(fb, wk.z, wk.w, tabA);
            __builtin_amdgcn_sched_barrier(0);
            {
               const uint32_t entry = cur;
               uint32_t st[8], t = cur;
#pragma unroll
               for (int q = 0; q < 8; ++q) {
                  t = fxstep(fa[q], t, TAp);
                  st[q] = t;
               }
               const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
               const bool act = 2u * c >= g0;
               cur = act ? t : cur;
               const bool hit = act && mx >= P.acc_min;
               gl2 = hit ? 2u * c : gl2;
               el2 = hit ? entry : el2;
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // (unconditional -- after the last chunk the end-of-row column once more, unused: behind a branch the lookups landed in
                //  fresh registers and were copied back at the loop's end, twelve 64-bit moves per chunk)
               wk = wn;
               lookup8(fa, wk.x, wk.y, tabA);
               wn = tile[tile_cell(lane, c + 2u <= (uint32_t)CH ? c + 2u : (uint32_t)CH)];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
               const uint32_t entry = cur;
               uint32_t st[8], t = cur;
#pragma unroll
               for (int q = 0; q < 8; ++q) {
                  t = fxstep(fb[q], t, TAp);
                  st[q] = t;
               }
               const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
               const bool act = 2u * c + 1u >= g0;
               cur = act ? t : cur;
               const bool hit = act && mx >= P.acc_min;
               gl2 = hit ? 2u * c + 1u : gl2;
               el2 = hit ? entry : el2;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (__builtin_amdgcn_ballot_w64(cur != 0) == 0) break;
         }
#endif
         // the exact symbol of the last accept: re-walk that group (every lane one group)
         if (__builtin_amdgcn_ballot_w64(gl2 != 0xFFFFFFFFu) != 0) {
            const uint32_t g = gl2 != 0xFFFFFFFFu ? gl2 : 0u;
            const uint2 rw = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, g >> 1) << 4) + ((g & 1u) << 3));
            FA fr8[8];
            lookup8(fr8, rw.x, rw.y, tabA);
            uint32_t st = el2, loc = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
               st = fxstep(fr8[q], st, TAp);
               loc = st >= P.acc_min ? (uint32_t)q : loc;
            }
            mm = gl2 != 0xFFFFFFFFu ? 8u * g + loc + 3u : mm;
         }
      };
      (void)aligned_walk;
      if (__builtin_amdgcn_ballot_w64(cur != 0) != 0) {
         bool aligned_done = false;
         // Many lanes walk and their starts lie within three chunks of each other (config 4: at the row's first bytes): straight into the
         // aligned walk -- no window (370 instructions up to the loop against 126 for the same 32 bytes in it; scattered starts keep the
         // window: the aligned loop runs from the first start to the last end, and lanes wait in it for their group)
         if constexpr (ALN && FX_FWD_DIRECT != 0) {
            if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cur != 0)) >= (uint32_t)FX_FWD_ALIGNED_MIN) {
               uint32_t cmin = 0;
               while (cmin < (uint32_t)CH && __builtin_amdgcn_ballot_w64(cur != 0 && (j >> 4) <= cmin) == 0) ++cmin;
               if (__builtin_amdgcn_ballot_w64(cur != 0 && (j >> 4) > cmin + 2u) == 0) {
                  aligned_walk();
                  aligned_done = true;
               }
            }
         }
         if (!aligned_done) {
            // first window: 32 symbols from j (16 for rows of up to 64 bytes, where the window is a large share of the tile's work),
            // all lookups issued before the chain
            constexpr int NG = CH <= 4 ? 2 : 4;
            uint32_t o[2 * NG];
            fetch_groups<RAGGED, NG>(o, tb, lane, j, Lc);
            constexpr int GB = NG;   // 8-symbol groups whose lookups are issued together
            uint32_t gl = 0xFFFFFFFFu, el = 0, blo = 0, bhi = 0;
#pragma unroll
            for (int gb = 0; gb < NG; gb += GB) {
               FA f[8 * GB];
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
                  const bool hit = mx >= P.acc_min;
                  gl = hit ? (uint32_t)(gb + g) : gl;
                  el = hit ? entry : el;
                  blo = hit ? o[2 * (gb + g)] : blo;
                  bhi = hit ? o[2 * (gb + g) + 1] : bhi;
               }
            }
            {
               FA fr8[8];
               lookup8(fr8, blo, bhi, tabA);
               uint32_t st = el, loc = 0;
#pragma unroll
               for (int q = 0; q < 8; ++q) {
                  st = fxstep(fr8[q], st, TAp);
                  loc = st >= P.acc_min ? (uint32_t)q : loc;
               }
               mm = gl != 0xFFFFFFFFu ? j + 8u * gl + loc + 3u : mm;
            }
            j += 8u * NG;
            // Matches longer than the window.  MANY lanes still walking: the aligned walk.  Few lanes: 8 symbols per round trip from wherever
            // each lane stands (below).
            if constexpr (ALN) {
               if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cur != 0)) >= (uint32_t)FX_FWD_ALIGNED_MIN) {
                  aligned_walk();
                  aligned_done = true;
               }
            }
         }
         // 8 symbols per round trip, the next group read one round ahead (see fx_search_fast)
         if (!aligned_done && __builtin_amdgcn_ballot_w64(cur != 0) != 0) {
            const uint32_t sh = j & 7u, up = 0u - ((sh >> 2) & 1u);
            uint32_t gb = j & ~7u;
            uint32_t t0[2], t1[2];
            group_words<RAGGED, false>(t0[0], t0[1], tb, lane, gb, Lc);
            group_words<RAGGED, false>(t1[0], t1[1], tb, lane, gb + 8u, Lc);
            // per round only "any accept" (one max tree) + the entry state and the round's symbols are kept; the last accepting round
            // is re-walked afterwards for the exact symbol (as in the window above) -- 4 instructions per round instead of 16
            uint32_t jl = 0xFFFFFFFFu, el2 = 0, lo2 = 0, hi2 = 0;
            do {
               uint32_t t2[2];
               group_words<RAGGED, false>(t2[0], t2[1], tb, lane, gb + 16u, Lc);
               const uint32_t e0 = (up & t0[1]) | (~up & t0[0]), e1 = (up & t1[0]) | (~up & t0[1]), e2 = (up & t1[1]) | (~up & t1[0]);
               const uint32_t o0 = __builtin_amdgcn_alignbyte(e1, e0, sh & 3u), o1 = __builtin_amdgcn_alignbyte(e2, e1, sh & 3u);
               FA f8[8];
               lookup8(f8, o0, o1, tabA);
               const uint32_t entry = cur;
               uint32_t st[8];
#pragma unroll
               for (int q = 0; q < 8; ++q) {
                  cur = fxstep(f8[q], cur, TAp);
                  st[q] = cur;
               }
               const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
               const bool hit = mx >= P.acc_min;
               jl = hit ? j : jl;
               el2 = hit ? entry : el2;
               lo2 = hit ? o0 : lo2;
               hi2 = hit ? o1 : hi2;
               j += 8u;
               gb += 8u;
               t0[0] = t1[0]; t0[1] = t1[1];
               t1[0] = t2[0]; t1[1] = t2[1];
            } while (__builtin_amdgcn_ballot_w64(cur != 0) != 0);
            if (__builtin_amdgcn_ballot_w64(jl != 0xFFFFFFFFu) != 0) {
               FA fr8[8];
               lookup8(fr8, lo2, hi2, tabA);
               uint32_t st = el2, loc = 0;
#pragma unroll
               for (int q = 0; q < 8; ++q) {
                  st = fxstep(fr8[q], st, TAp);
                  loc = st >= P.acc_min ? (uint32_t)q : loc;
               }
               mm = jl != 0xFFFFFFFFu ? jl + loc + 3u : mm;
            }
         }
      }
      if constexpr (ROW_EXC && !BYTES && !DECODED) {
         if (sfx_on) {   // FXP_F_SUFFIX_CHECK: the match found ends with the suffix literal, behind its start -- else the row is the general procedure's
            bool ok = true;
            if (s != 0u && !except && row_ok && !pfx_nowhere) {
               const FxTileRow tr{tb, lane};
               const uint8_t* const sp = c.sfx;
               ok = fxrow::suffix_end_ok([&](int k) -> uint32_t { return sp[k]; }, (int)c.sfx_len, tr, (int)L, (int)s, (int)mm);
            }
            except = except || !ok;
         }
      }
      uint32_t flag = 0;
      int32_t fr = 0, tt = 0;
      if (SPANS) {
         if (s != 0 && mm != 0) {   // api_internal_m.F90:140-148
            fr = (int32_t)(s - 1);
            if (fr == 0) fr = 1;
            tt = mm >= L + 2u ? (int32_t)L : (int32_t)mm - 2;
            if (fr > 0 && tt > 0) flag = 1;
            else { fr = 0; tt = 0; }
         }
      } else {
         flag = (s >= 2 || (s == 1 && mm > 2)) ? 1u : 0u;
      }
      if (queued) flag = 1;   // (from / to follow when the queue is flushed)
      emit(row, row_ok && !except, ordered, flag, fr, tt, !queued);
      return false;
   }
}

// ---- speculative forward pass (round 4; FXP_F_SPEC_FWD) ----------------------------------------------------------------
// The reference tries the candidates in order: the leading NUL, the row's first character, the second, ... (api_internal_m.F90:84-88,
// 108-155) and takes the first one with a NON-EMPTY match.  For programs whose anchored automaton dies on the leading NUL, a walk of A from
// the row's first byte that reaches an accept therefore IS the answer: from = 1, to = the longest accept of that walk -- no backward
// pass.  Every lane walks its row from byte 0 in aligned 8-byte groups with the 8-state v_perm byte-level tables (one v_perm_b32 per byte,
// three lookup buffers as in the aligned forward walk of fx_scan_tile) until every lane is dead (flags only: dead or accepted).  Returns
// max_match (SPANS: the wrapped index of the byte after the longest match; flags only: nonzero) or 0 when the first character starts no
// match: such rows take the backward + forward scan (queued per wave, or in place when the tile is dense in them).
template <int CH, bool SPANS>
__device__ __forceinline__ uint32_t fx_spec_forward(const uint4* tile, const uint8_t* tb, const uint32_t lane, const uint2* __restrict__ tabA,
                                                    const FastParams& P, const bool on) {
   uint32_t cur = on ? P.A_init : 0u;
   uint32_t gl2 = 0xFFFFFFFFu, el2 = 0;
   auto cellc = [&](const uint32_t c) { return tile[tile_cell(lane, c <= (uint32_t)CH ? c : (uint32_t)CH)]; };
   auto step8 = [&](const uint2 (&f)[8], const uint32_t g) {
      const uint32_t entry = cur;
      uint32_t st[8], t = cur;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
         t = fxstep(f[q], t, nullptr);
         st[q] = t;
      }
      const uint32_t mx = max(max(max(max(st[0], st[1]), st[2]), max(max(st[3], st[4]), st[5])), max(st[6], st[7]));
      cur = t;
      const bool hit = mx >= P.acc_min;
      gl2 = hit ? g : gl2;
      el2 = hit ? entry : el2;
   };
   uint2 fa[8], fb[8], fc[8];
   uint4 w0 = cellc(0u), w1 = cellc(1u), w2 = cellc(2u);
   lookup8(fa, w0.x, w0.y, tabA);
   lookup8(fb, w0.z, w0.w, tabA);
   // Whole trips of three text chunks (six groups), then the CH % 3 chunks left, then ONE group of the end-of-row column (chunk CH: the NUL, then
   // KILL symbols -- no lane outlives it).  (Round 6: the loop used to run ceil((CH + 1) / 3) whole trips with the chunk index clamped to the
   // end-of-row column -- config 4's 192-byte rows, whose matches mostly reach the row's end: 30 groups of lookups for 24 groups of text, a
   // sixth of the kernel's LDS instructions, which are what bounds it: profiles/r06_cfg4_phases.md.)
   constexpr uint32_t TRIPS = (uint32_t)CH / 3u, REST = (uint32_t)CH % 3u;
   bool alive = true;
   // in flight on entry of a trip: fa, fb = the two groups of chunk c (w0), w1 / w2 = chunks c + 1 / c + 2.  (END: the trip before the end-of-row
   // column -- CH % 3 == 0, the last trip --: the column's second half is never walked, its lookups are not issued.  A trip of its own at
   // compile time: a run-time guard around those eight lookups kept both versions of the buffer live -- 232 -> 266 registers, one wave per SIMD.)
   auto trip = [&](auto end_, const uint32_t c) {
      constexpr bool END = decltype(end_)::value;
      lookup8(fc, w1.x, w1.y, tabA);
      __builtin_amdgcn_sched_barrier(0);
      step8(fa, 2u * c);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fa, w1.z, w1.w, tabA);
      w0 = cellc(c + 3u);
      __builtin_amdgcn_sched_barrier(0);
      step8(fb, 2u * c + 1u);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fb, w2.x, w2.y, tabA);
      if (!END) w1 = cellc(c + 4u);
      __builtin_amdgcn_sched_barrier(0);
      step8(fc, 2u * c + 2u);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fc, w2.z, w2.w, tabA);
      if (!END) w2 = cellc(c + 5u);
      __builtin_amdgcn_sched_barrier(0);
      step8(fa, 2u * c + 3u);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fa, w0.x, w0.y, tabA);
      __builtin_amdgcn_sched_barrier(0);
      step8(fb, 2u * c + 4u);
      __builtin_amdgcn_sched_barrier(0);
      if (!END) lookup8(fb, w0.z, w0.w, tabA);
      __builtin_amdgcn_sched_barrier(0);
      step8(fc, 2u * c + 5u);
      __builtin_amdgcn_sched_barrier(0);
      alive = __builtin_amdgcn_ballot_w64(cur != 0 && (SPANS || gl2 == 0xFFFFFFFFu)) != 0;
   };
   constexpr uint32_t PLAIN = (REST == 0u && TRIPS >= 1u) ? TRIPS - 1u : TRIPS;   // trips of the rolled loop
#pragma unroll 1
   for (uint32_t c = 0; c < 3u * PLAIN; c += 3u) {
      trip(std::false_type{}, c);
      if (!alive) break;
   }
   if constexpr (PLAIN != TRIPS) {
      if (alive) trip(std::true_type{}, 3u * PLAIN);
   }
   if (alive) {   // c = 3 TRIPS: fa (and, for a text chunk, fb) hold chunk c's groups, w1 / w2 = chunks c + 1 / c + 2 (clamped to the end-of-row column)
      constexpr uint32_t c = 3u * TRIPS;
      if constexpr (REST == 0) {
         step8(fa, 2u * c);
      } else if constexpr (REST == 1) {
         lookup8(fc, w1.x, w1.y, tabA);
         step8(fa, 2u * c);
         step8(fb, 2u * c + 1u);
         if (__builtin_amdgcn_ballot_w64(cur != 0 && (SPANS || gl2 == 0xFFFFFFFFu)) != 0) step8(fc, 2u * c + 2u);
      } else {
         lookup8(fc, w1.x, w1.y, tabA);
         step8(fa, 2u * c);
         lookup8(fa, w1.z, w1.w, tabA);
         step8(fb, 2u * c + 1u);
         lookup8(fb, w2.x, w2.y, tabA);
         step8(fc, 2u * c + 2u);
         step8(fa, 2u * c + 3u);
         if (__builtin_amdgcn_ballot_w64(cur != 0 && (SPANS || gl2 == 0xFFFFFFFFu)) != 0) step8(fb, 2u * c + 4u);
      }
   }
   if (!SPANS) return gl2 != 0xFFFFFFFFu ? 3u : 0u;
   uint32_t mm = 0;
   if (__builtin_amdgcn_ballot_w64(gl2 != 0xFFFFFFFFu) != 0) {   // the exact symbol of the last accept: re-walk that group
      const uint32_t g = gl2 != 0xFFFFFFFFu ? gl2 : 0u;
      const uint2 rw = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, g >> 1) << 4) + ((g & 1u) << 3));
      uint2 fr8[8];
      lookup8(fr8, rw.x, rw.y, tabA);
      uint32_t st = el2, loc = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
         st = fxstep(fr8[q], st, nullptr);
         loc = st >= P.acc_min ? (uint32_t)q : loc;
      }
      mm = gl2 != 0xFFFFFFFFu ? 8u * g + loc + 3u : 0u;
   }
   return mm;
}

// ---- `.match.` on a tile in LDS: one forward pass of the anchored automaton over every byte of the row from M_start (the state after
// the optional leading NUL, api_internal_m.F90:280-289), verdict = the final state's FINAL entry (accept at ci = n + 2 or after the
// trailing NUL, :296-302), behind the reference's literal / prefix / suffix gate (`gate`: 2 = TRUE, 0 = FALSE, 1 = the automaton
// decides; fxrow::match_gate, evaluated by the caller on the RAW bytes).  Table families, REDO_TILE / ROW_EXC / `except` as in
// fx_scan_tile; byte-level tables: a row whose walk ends inside a character or in the INVALID state (FINAL = 2) is an exception.
template <int CH, bool RAGGED, int S_, bool BYTES, bool DECODED, bool REDO_TILE, bool ROW_EXC, bool TAIL = false, class TabT, class Emit>
__device__ __forceinline__ bool fx_match_tile(const FxScanCtx& c, const TabT* __restrict__ tabA, const uint8_t* TAp, const FastParams& P, const FxpHeader* h,
                                              const uint32_t gate, const int64_t row, const bool row_ok, const bool ordered, bool& except, Emit& emit) {
   constexpr bool CHAIN = S_ == 1, WIDE = S_ == 2;
   using F = typename FxF<S_>::type;
   uint4* const tile = c.tile;
   const uint32_t lane = c.lane, Lr = c.Lr;
   const bool whole = c.whole;
   (void)Lr;
   (void)whole;
   uint32_t st = P.A_init;   // = M_start
   uint32_t na = 0;
   if (RAGGED && (!whole || DECODED)) na |= pad_rows<CH>(tile, lane, Lr);   // pads (symbol 255) are the identity for A
   static_assert(!TAIL || !RAGGED, "TAIL: the round-4 ragged scheme (no pad symbol)");
   if constexpr (TAIL) {
      // ragged rows (fx_tile.hpp, "Ragged rows, round 4"): whole chunks in a rolled loop, then the chunk the row ends in over its text bytes
      const FxTail T = fx_tail_here(*c.tl);
      // (rows of 96 bytes and more on the 8-state tables: whole trips of three chunks with three lookup buffers first, as the aligned rows)
      uint32_t k0 = 0;
      if constexpr (FX_MATCH_PIPE3 != 0 && S_ == 0 && CH >= FX_MATCH_P3_MINCH) {
         if (T.kt >= (uint32_t)FX_MATCH_P3_MINCH) {
            uint32_t nax = 0;
            k0 = 3u * (T.kt / 3u);
            fx_match_trips_pipe3<CH>(tile, lane, tabA, st, nax, T.kt / 3u);
            if (!DECODED) na |= nax;
         }
      }
      F fa[8], fb[8];
      uint4 wk = tile[tile_cell(lane, k0)];   // (k0 <= kt <= CH - 1)
      lookup8(fa, wk.x, wk.y, tabA);
#pragma unroll 1
      for (uint32_t k = k0; k < T.kt; ++k) {
         if (!DECODED) na |= wk.x | wk.y | wk.z | wk.w;
         lookup8(fb, wk.z, wk.w, tabA);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fa, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         wk = tile[tile_cell(lane, k + 1u)];   // (k + 1 <= kt <= CH - 1: a ragged row is shorter than 16 * CH bytes)
         lookup8(fa, wk.x, wk.y, tabA);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fb, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
      }
      if (T.nb != 0u) {   // `wk` is chunk kt, `fa` the lookups of its first group
         if (!DECODED) na |= fx_tail_or(wk, T.nb);
         if (T.nb > 8u) lookup8(fb, wk.z, wk.w, tabA);
         if (T.nb >= 8u) chain8_fwd(fa, st, TAp);
         else chain8_fwd_n(fa, st, TAp, T.nb);
         if (T.nb > 8u) chain8_fwd_n(fb, st, TAp, T.nb - 8u);
      }
   } else if constexpr (FX_MATCH_PIPE3 != 0 && S_ == 0 && CH >= FX_MATCH_P3_MINCH && !RAGGED) {
      // 8-state tables: THREE lookup buffers, a group's lookups issued two chains ahead of its use (a chain of eight v_perm_b32 is 32 cycles:
      // one chain ahead, the lookups come back late -- see the aligned forward walk of fx_scan_tile).  Three chunks per trip.  Measured
      // (profiles/r03_pipe3_ab.txt): 10 M x 256 B 0.476-0.488 -> 0.415-0.430 ms (two waves per SIMD), 12.5 M x 128 B 0.290 -> 0.263 ms (three).
      F fa[8], fb[8], fc[8];
      auto cellc = [&](const uint32_t c) { return tile[tile_cell(lane, c < (uint32_t)CH ? c : (uint32_t)CH - 1u)]; };
      uint4 w0 = cellc(0), w1 = cellc(1), w2 = cellc(2);
      lookup8(fa, w0.x, w0.y, tabA);
      lookup8(fb, w0.z, w0.w, tabA);
      constexpr uint32_t TRIPS = (uint32_t)CH / 3u, REST = (uint32_t)CH % 3u;   // whole trips of three chunks, then REST chunks (their lookups are in flight)
#pragma unroll 1
      for (uint32_t c = 0; c < 3u * TRIPS; c += 3u) {
         na |= w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w | w2.x | w2.y | w2.z | w2.w;
         lookup8(fc, w1.x, w1.y, tabA);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fa, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         lookup8(fa, w1.z, w1.w, tabA);
         w0 = cellc(c + 3u);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fb, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         lookup8(fb, w2.x, w2.y, tabA);
         w1 = cellc(c + 4u);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fc, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         lookup8(fc, w2.z, w2.w, tabA);
         w2 = cellc(c + 5u);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fa, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         lookup8(fa, w0.x, w0.y, tabA);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fb, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         lookup8(fb, w0.z, w0.w, tabA);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fc, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (REST >= 1) {   // chunk 3 TRIPS: its lookups are in fa / fb, its words in w0
         na |= w0.x | w0.y | w0.z | w0.w;
         if constexpr (REST == 2) {
            na |= w1.x | w1.y | w1.z | w1.w;
            lookup8(fc, w1.x, w1.y, tabA);
         }
         chain8_fwd(fa, st, TAp);
         if constexpr (REST == 2) lookup8(fa, w1.z, w1.w, tabA);
         chain8_fwd(fb, st, TAp);
         if constexpr (REST == 2) {
            chain8_fwd(fc, st, TAp);
            chain8_fwd(fa, st, TAp);
         }
      }
   } else
   {
      F fa[8], fb[8];
      uint4 wk = tile[tile_cell(lane, 0)], wn = make_uint4(0, 0, 0, 0);
      if (CH >= 2) wn = tile[tile_cell(lane, 1)];
      lookup8(fa, wk.x, wk.y, tabA);
#pragma unroll 1   // rolled on purpose: fully unrolled, the state-independent lookups of ALL chunks get hoisted (registers)
      for (int k = 0; k < CH; ++k) {
         if (!RAGGED || (whole && !DECODED && (uint32_t)k < (Lr >> 4))) na |= wk.x | wk.y | wk.z | wk.w;
         lookup8(fb, wk.z, wk.w, tabA);
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fa, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
         if (k + 1 < CH) {
            wk = wn;
            lookup8(fa, wk.x, wk.y, tabA);
            if (k + 2 < CH) wn = tile[tile_cell(lane, k + 2)];
         }
         __builtin_amdgcn_sched_barrier(0);
         chain8_fwd(fb, st, TAp);
         __builtin_amdgcn_sched_barrier(0);
      }
   }
   bool row_hi = false;
   if (!BYTES && !DECODED) {
      row_hi = (na & 0x80808080u) != 0;
      if (REDO_TILE && __builtin_amdgcn_ballot_w64(row_hi) != 0) return true;
   }
   uint32_t fin;
   if (CHAIN) fin = *reinterpret_cast<const uint16_t*>(TAp + st + 2u * ((BYTES ? h->byte_n_classes : h->n_classes) + 2u));   // FINAL column
   else if (WIDE) {
      const uint32_t* fm = BYTES ? h->bw16_finalM : h->w16_finalM;   // byte j = verdict of state j
      fin = (fm[(st >> 2) & 3u] >> ((st & 3u) * 8u)) & 3u;
   } else fin = __builtin_amdgcn_perm(h->fast_finalM[1], h->fast_finalM[0], st) & 1u;
   const uint32_t flag = gate == 2u ? 1u : (gate == 0u ? 0u : (st != 0 && fin == 1u ? 1u : 0u));
   except = (BYTES && gate == 1u && st != 0 && fin == 2u) || (ROW_EXC && !BYTES && !DECODED && row_hi);
   emit(row, row_ok && !except, ordered, flag, 0, 0, true);
   return false;
}

// SCH: scheme of the class-level tables (0 v_perm, 1 chain, 2 wide); BSCH: scheme of the byte-level tables (0 = none in this
// launch, 1 chain, 2 wide).  SCH == 0 && BSCH != 0: per-tile selection.  SCH != 0 && BSCH != 0: the byte-level tables take every
// tile (the class-level ones are no faster on ASCII), the class-level tables only serve the exception rows.
// GEN: the program's class-level tables cannot decode UTF-8 (candidate-list driver: a prefix literal proven equal to brute force on
// pure-ASCII rows only, FXP_F_OVERLAP_SINK programs): rows the tables cannot answer -- exception rows of the byte-level tables,
// rows with a byte >= 0x80 when there are no byte-level tables, overlap rows of a bordered prefix -- are queued the same way and
// the gathered rows go through the GENERAL row procedure (fxrow::run_row, the body of fx_general) instead of the decode + scan.
// MARKED: the follow-up of a multi-pass first pass (the half-row kernel of 256-byte rows): only tiles whose rows that pass marked
// FX_NEEDS_GENERAL are staged and finished here -- with the byte-level tables or the in-LDS decode, and the exception queues --
// and the launch leaves at once when `gate` says nothing was deferred.  ONE gated launch instead of two.
// ---- optional phase stamps of fx_search_one (debug builds only: `make stamp-one`, tools/stamp_one.py) --------------------------------------
// -DFX_STAMP_ONE: every wave accumulates s_memtime deltas per phase in scalar registers and lane 0 adds them to fx_one_stamp_acc[] at its end.
// The array has internal linkage (one copy per translation unit): the object built with the macro exports fxamd_debug_stamps_one (fx_tile_inst.hip).
#ifdef FX_STAMP_ONE
#define FX_STAMP_MAX_WAVES 16384
#define FX_STAMP_SLOTS 20
static __device__ unsigned long long fx_one_stamp_buf[FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS];   // per wave: slots 0..11 and 14..17 phase ticks, 12 tiles, 13 gathered passes, 18 lifetime, 19 launches seen
// (the per-phase sums live in LDS, 16 slots per wave, added to by lane 0 with ds_add_u64, and every wave ADDS its slots to its own row of the global buffer
//  with plain stores at its end -- no atomics.  A first version kept twelve 64-bit accumulators in scalar registers: the kernel then ran at one wave per
//  SIMD and took 242 us instead of 51; a second one added every wave's sums to ONE set of global words: 35 k same-address atomics at ~12 ns each made the
//  launch 346 us and delayed the last waves' own memory operations)
#define ONE_STAMP_DECL                                                                   \
   __shared__ unsigned long long _os_lds[4 * FX_STAMP_SLOTS];                            \
   if (threadIdx.x < 4 * FX_STAMP_SLOTS) _os_lds[threadIdx.x] = 0ull;                    \
   const unsigned long long _os_t0 = __builtin_amdgcn_s_memtime();                       \
   unsigned long long _os_t = _os_t0
#define ONE_STAMP(i)                                                                                                                         \
   do {                                                                                                                                      \
      const unsigned long long _n = __builtin_amdgcn_s_memtime();                                                                            \
      if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(&_os_lds[(threadIdx.x >> 6) * FX_STAMP_SLOTS + (i)], _n - _os_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      _os_t = _n;                                                                                                                            \
   } while (0)
#define ONE_STAMP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define ONE_STAMP_COUNT(slot)                                                                                                                \
   do {                                                                                                                                      \
      if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(&_os_lds[(threadIdx.x >> 6) * FX_STAMP_SLOTS + (slot)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
   } while (0)
#define ONE_STAMP_FLUSH                                                                                   \
   do {                                                                                                   \
      const unsigned long long _life = __builtin_amdgcn_s_memtime() - _os_t0;                             \
      if (lane < (uint32_t)FX_STAMP_SLOTS && wave_global < FX_STAMP_MAX_WAVES) {                          \
         unsigned long long* _row = fx_one_stamp_buf + wave_global * FX_STAMP_SLOTS;                      \
         _row[lane] += lane == 18u ? _life : (lane == 19u ? 1ull : _os_lds[wave * FX_STAMP_SLOTS + lane]); \
      }                                                                                                   \
   } while (0)
#else
#define ONE_STAMP_DECL
#define ONE_STAMP(i)
#define ONE_STAMP_WAIT_VM()
#define ONE_STAMP_COUNT(slot)
#define ONE_STAMP_FLUSH
#endif
#ifndef FX_ONE_ROWS_FIRST
#define FX_ONE_ROWS_FIRST 0   // (1: the first tile's loads before the table reads -- measured SLOWER, see the start-up comment in the kernel)
#endif
template <int CH, bool SPANS, int SCH, int BSCH, bool RAGGED, bool GEN, bool MARKED = false, bool MATCH = false>
__global__ __launch_bounds__(256, (FX_ONE_MINW > 1 ? FX_ONE_MINW : ((CH == 6 || CH == 8) ? 3 : ((CH == 16 && SCH == 0 && BSCH != 0 && !RAGGED && !GEN && !MARKED && !MATCH) ? 2 : 1)))) void fx_search_one(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog, FastParams fp,
                                                       FastParams fpb, uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                       uint32_t class_map_in_lds, uint32_t Lr, uint32_t out_mode, const uint32_t* __restrict__ gate = nullptr,
                                                       const uint8_t* __restrict__ marks = nullptr) {
   if (MARKED && blockIdx.x == 0 && threadIdx.x == 0) {
      // (FX_ADAPT_CALLS, fx_tile.hpp) the persistent word behind the two counter groups: count down while it is set, else look at what the
      // first pass sampled -- more than half of its tiles deferred sets it.  No other block of this kernel reads it; the next call's first
      // pass does, after this kernel.
      uint32_t* hintw = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(gate) & ~uintptr_t(31)) + 32u);
      const uint32_t hv = hintw[0];
      if (hv != 0u) hintw[0] = hv - 1u;
      else if (gate[3] >= 8u && gate[2] * 2u > gate[3]) hintw[0] = FX_ADAPT_CALLS;
   }
   if (MARKED && gate[0] == 0u) return;   // nothing was deferred
   ONE_STAMP_DECL;
   // out_mode 0: flags u8[n], from / to int32[n].  out_mode 1 / 2 / 4: PACKED results (what a multi-GPU host gathers):
   // `flags` = 1 bit per row (row i = bit i & 63 of the 64-bit word i >> 6: the ballot of the tile's wave, one store per tile),
   // `from` / `to` = arrays of that many bytes per row (uint8 / uint16 / int32).
   // RAGGED: rows of any length 2 <= Lr < 16 * CH, left-aligned in their cells with the trailing NUL and KILL symbols behind the text
   // (fx_tile.hpp, "Ragged rows, round 4"): no pad symbol, so every table family -- the byte-level ones too -- runs on them
   static_assert(!MATCH || (!SPANS && !MARKED), "`.match.` has no span and no multi-pass first pass");
   constexpr bool HAS_B = BSCH != 0, ALLB = HAS_B && SCH != 0, POOL = HAS_B || GEN;
   using BCfg = FxScanCfg<(BSCH == 3 ? 2 : (BSCH != 0 ? BSCH : 1)), true, false, (BSCH == 3 ? 0 : (BSCH != 0 ? BSCH : 1))>;   // the byte-level tables' scan
   static_assert(BSCH != 3 || !MATCH, "FXP_F_BYTE_A8 is a search program's table");
   const uint32_t L = RAGGED ? Lr : 16u * CH;
   __shared__ uint2 permR[SCH == 0 ? 256 : 1];
   __shared__ uint2 permA[SCH == 0 ? 256 : 1];
   __shared__ fx_nib wideR[SCH == 2 ? 256 : 1];
   __shared__ fx_nib wideA[SCH == 2 ? 256 : 1];
   __shared__ fx_nib bwideR[(BSCH == 2 || BSCH == 3) ? 256 : 1];
   __shared__ fx_nib bwideA[BSCH == 2 ? 256 : 1];
   __shared__ uint2 bpermA[BSCH == 3 ? 256 : 1];   // BSCH 3: byte-level tables, nibble format backwards, 8-state v_perm format forwards (FXP_F_BYTE_A8)
   __shared__ uint32_t pool_q[POOL ? 4 * 64 : 1];   // per-wave queues of exception rows
   // speculative forward pass (fx_spec_forward): byte-level tables with the 8-state forward automaton, programs that say it is sound
   // (fpb.spec: FXP_F_SPEC_FWD); rows whose first character starts no match are queued per wave for the backward + forward scan
   constexpr bool SPEC = FX_SPEC_FWD != 0 && BSCH == 3 && !MATCH && !GEN;
   __shared__ uint32_t spec_q[SPEC ? 4 * 64 : 1];
   // match compaction (see fx_scan_tile): rows of up to 64 bytes, where the exact start + first forward window are a third of a tile's
   // instructions (config 2: 21.0 -> 18.8 us); on longer rows its bookkeeping cost more than it saved on the BASELINE shapes
   // (config 5: +2 %, config 4: +1 %, profiles/r03_defer_ab.txt)
   constexpr bool DEFERQ = FX_DEFER_FWD != 0 && SPANS && !MARKED && !MATCH && CH <= 4 && !RAGGED;
   __shared__ uint32_t fwd_q[DEFERQ ? 4 * 128 : 1];   // per-wave queues of rows whose exact start + forward pass are finished 64 at a time
   __shared__ uint2 fwd_w[(DEFERQ && FX_DEFER_STASH != 0) ? 4 * 192 : 1];   // ... and their first three 8-byte groups (FX_DEFER_STASH)
   extern __shared__ __attribute__((aligned(16))) uint4 tiles[];   // 4 waves x 64*(CH+1) cells [+ class chain tables] [+ byte chain tables] [+ class map]
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int64_t n_tiles = (n + 63) >> 6;
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   // Start-up: this thread's entries of the 256-entry tables are READ first, then the first tile's global loads go out, and only then are the
   // table entries written to LDS -- the block's two start-up latencies (tables from L2, rows from HBM) overlap instead of adding up, and
   // the wait for the table entries (the older loads: vmcnt counts in order) does not wait for the rows.  (The marked-tile follow-up reads the
   // flags before it loads a tile.)
   // (Round 5, FX_ONE_ROWS_FIRST = 1, measured and NOT kept: the first tile's loads BEFORE the table reads -- they need no header field -- so that
   //  the start-up is max(rows, header + tables) instead of header + max(tables, rows): config 2 18.2 -> 18.8-19.4 us per step, config 4 55.1 ->
   //  56.0 us (gpurun call r05_c26) -- the table entries then return behind 4 KB of rows per wave, and the LDS stores + barrier wait for all of it.)
   uint4 stage[CH];
   const FxTail tl = fx_tail_of(RAGGED ? Lr : 16u * CH);
   if constexpr (!MARKED && FX_ONE_ROWS_FIRST != 0) {
      if constexpr (RAGGED) load_tile_rag<CH>(stage, rows, wave_global << 6, n, lane, tl);
      else load_tile<CH>(stage, rows, wave_global << 6, n, lane, true);
      __builtin_amdgcn_sched_barrier(0);
   }
   uint2 t_r = make_uint2(0, 0), t_a = make_uint2(0, 0), t_br = make_uint2(0, 0), t_ba = make_uint2(0, 0);
   if (SCH == 2) {
      t_r = reinterpret_cast<const uint2*>(prog + h->off_w16R)[threadIdx.x];
      t_a = reinterpret_cast<const uint2*>(prog + h->off_w16A)[threadIdx.x];
   } else if (SCH == 0) {
      t_r = reinterpret_cast<const uint2*>(prog + h->off_fastR)[threadIdx.x];
      t_a = reinterpret_cast<const uint2*>(prog + h->off_fastA)[threadIdx.x];
   }
   if (BSCH == 2 || BSCH == 3) {
      t_br = reinterpret_cast<const uint2*>(prog + h->off_bw16R)[threadIdx.x];
      t_ba = reinterpret_cast<const uint2*>(prog + (BSCH == 2 ? h->off_bw16A : h->off_b8A))[threadIdx.x];
   }
   // the BMP class map of the in-LDS decode (2 KB page index + 128 B per page): 16-byte pieces, the first 256 of them read HERE with the
   // tables (round 5: a 2-byte copy loop behind the tables' LDS stores was five dependent round trips to L2 at the start of every block --
   // config 4 is a 57 us kernel)
   const uint32_t cm_n4 = class_map_in_lds ? 128u + h->n_pages * 8u : 0u;
   uint4 t_cm = make_uint4(0, 0, 0, 0);
   if (threadIdx.x < cm_n4)
      t_cm = threadIdx.x < 128u ? reinterpret_cast<const uint4*>(prog + h->off_cls_page)[threadIdx.x]
                                : reinterpret_cast<const uint4*>(prog + h->off_cls_pages)[threadIdx.x - 128u];
   __builtin_amdgcn_sched_barrier(0);
   if constexpr (!MARKED && FX_ONE_ROWS_FIRST == 0) {
      if constexpr (RAGGED) load_tile_rag<CH>(stage, rows, wave_global << 6, n, lane, tl);
      else load_tile<CH>(stage, rows, wave_global << 6, n, lane, true);
   }
   // ---- tables -> LDS ----
   uint8_t* dyn = reinterpret_cast<uint8_t*>(tiles + 4 * 64 * (CH + 1));
   uint16_t* cmap = reinterpret_cast<uint16_t*>(dyn);   // class-level chain: symbol -> 2*column map (512 B), then T_R, then T_A
   const uint32_t c_tr = SCH == 1 ? h->chain_TR_bytes : 0u, c_ta = SCH == 1 ? h->chain_TA_bytes : 0u;
   const uint32_t c_bytes = SCH == 1 ? ((512u + c_tr + c_ta + 15u) & ~15u) : 0u;
   uint16_t* bmap = reinterpret_cast<uint16_t*>(dyn + c_bytes);   // byte-level chain tables, same layout
   const uint32_t b_tr = BSCH == 1 ? h->byte_TR_bytes : 0u, b_ta = BSCH == 1 ? h->byte_TA_bytes : 0u;
   const uint32_t b_bytes = BSCH == 1 ? ((512u + b_tr + b_ta + 15u) & ~15u) : 0u;
   if (SCH == 1) {
      const uint16_t* g = reinterpret_cast<const uint16_t*>(prog + h->off_chain_cls);
      const uint16_t* gr = reinterpret_cast<const uint16_t*>(prog + h->off_chain_TR);
      const uint16_t* ga = reinterpret_cast<const uint16_t*>(prog + h->off_chain_TA);
      const uint32_t nr = c_tr / 2, na = c_ta / 2;
      fx_stage_chain(cmap, g, gr, ga, nr, na);
   } else if (SCH == 2) {
      reinterpret_cast<uint2*>(wideR)[threadIdx.x] = t_r;
      reinterpret_cast<uint2*>(wideA)[threadIdx.x] = t_a;
   } else {
      permR[threadIdx.x] = t_r;
      permA[threadIdx.x] = t_a;
   }
   if (BSCH == 1) {
      const uint16_t* g = reinterpret_cast<const uint16_t*>(prog + h->off_byte_cls);
      const uint16_t* gr = reinterpret_cast<const uint16_t*>(prog + h->off_byte_TR);
      const uint16_t* ga = reinterpret_cast<const uint16_t*>(prog + h->off_byte_TA);
      const uint32_t nr = b_tr / 2, na = b_ta / 2;
      fx_stage_chain(bmap, g, gr, ga, nr, na);
   } else if (BSCH == 2 || BSCH == 3) {
      reinterpret_cast<uint2*>(bwideR)[threadIdx.x] = t_br;
      if (BSCH == 2) reinterpret_cast<uint2*>(bwideA)[threadIdx.x] = t_ba;
      else bpermA[threadIdx.x] = t_ba;
   }
   // BMP class map (page index + pages) of the in-LDS UTF-8 decode, behind the tables when it fits
   const uint16_t* page_p = reinterpret_cast<const uint16_t*>(prog + h->off_cls_page);
   const uint16_t* pages_p = reinterpret_cast<const uint16_t*>(prog + h->off_cls_pages);
   if (class_map_in_lds) {
      uint16_t* l16 = reinterpret_cast<uint16_t*>(dyn + c_bytes + b_bytes);
      uint4* l4 = reinterpret_cast<uint4*>(l16);   // (blob offsets and the LDS offset are multiples of 16: compile.cpp Blob::put, c_bytes / b_bytes)
      if (threadIdx.x < cm_n4) l4[threadIdx.x] = t_cm;
      for (uint32_t i = threadIdx.x + 256u; i < cm_n4; i += 256u) l4[i] = reinterpret_cast<const uint4*>(pages_p)[i - 128u];   // (more than 16 pages)
      page_p = l16;
      pages_p = l16 + 1024;
   }
   __syncthreads();
   const fxrow::ClassTables ct{page_p, pages_p, reinterpret_cast<const uint16_t*>(prog + h->off_bound_cls),
                               reinterpret_cast<const int32_t*>(prog + h->off_bounds), h->n_bounds};
   const uint32_t sym_ffff = 128u + h->cls_ffff;
   const bool raw = (h->flags & FXP_F_RAW_BYTES) != 0;   // literal search: bytes are symbols, nothing is decoded
   uint4* tile = tiles + wave * (64 * (CH + 1));
   // one extra chunk column per row holds what follows the text: the trailing NUL (symbol 0), then KILL symbols (see fx_search_fast)
   // (ragged rows: the NUL sits inside the row's cells, right behind the text, and this column holds KILL symbols only -- a second NUL would
   //  be a second line end to patterns like `$$`)
   tile[tile_cell(lane, CH)] = make_uint4(RAGGED ? 0xFEFEFEFEu : 0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
   const bool whole = false;
   if (RAGGED) fx_tail_init<CH>(tile, lane, tl);   // chunks behind the text: the trailing NUL / KILL symbols, written once
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   uint32_t* myq = pool_q + (POOL ? wave * 64u : 0u);
   uint32_t* mysq = spec_q + (SPEC ? wave * 64u : 0u);

   // results of one row.  `ordered`: the wave holds 64 consecutive rows (a tile of the batch): the packed flag word is its ballot;
   // a gathered row (exception queue) sets its bit in the word its tile's wave stored earlier.
   // spans_live == false: a row of the match-compaction queue -- its flag is final, its from / to are stored when the queue is flushed
   auto emit = [&](const int64_t row, const bool live, const bool ordered, const uint32_t flag, const int32_t fr, const int32_t tt, const bool spans_live = true) {
      if (out_mode == 0u) {
         if (live) {
            flags[row] = (uint8_t)flag;
            if (SPANS && spans_live) {
               from[row] = fr;
               to[row] = tt;
            }
         }
         return;
      }
      if (ordered) {
         // (rows the queue will redo, and rows behind the batch's end, contribute a zero bit)
         const uint64_t m = __builtin_amdgcn_ballot_w64(live && flag != 0u);
         if (lane == 0) reinterpret_cast<uint64_t*>(flags)[__builtin_amdgcn_readfirstlane((uint32_t)(row >> 6))] = m;
      } else if (live && flag != 0u) {
         atomicOr(reinterpret_cast<uint32_t*>(flags) + (row >> 5), 1u << ((uint32_t)row & 31u));
      }
      if (SPANS && live && spans_live) {
         if (out_mode == 1u) {
            reinterpret_cast<uint8_t*>(from)[row] = (uint8_t)fr;
            reinterpret_cast<uint8_t*>(to)[row] = (uint8_t)tt;
         } else if (out_mode == 2u) {
            reinterpret_cast<uint16_t*>(from)[row] = (uint16_t)fr;
            reinterpret_cast<uint16_t*>(to)[row] = (uint16_t)tt;
         } else {
            from[row] = fr;
            to[row] = tt;
         }
      }
   };

   // ---- one scan of the tile in LDS (fx_scan_tile) with the tables of one family ----------------------------------------------
   const bool pfx_chk = GEN && !MATCH && (h->flags & FXP_F_PREFIX_CHECK) != 0u;
   const bool sfx_chk = pfx_chk && (h->flags & FXP_F_SUFFIX_CHECK) != 0u;
   const FxScanCtx sctx{tile, tb, lane, L, Lr, whole, raw, 0u, &tl, pfx_chk ? prog + h->off_prefix : nullptr, pfx_chk ? h->len_prefix : 0u,
                        sfx_chk ? prog + h->off_suffix : nullptr, sfx_chk ? h->len_suffix : 0u};
   // tables of one family (class-level / byte-level) in the scheme `S_`, handed to `fn(tabR, tabA, TRp, TAp, P)`
   auto with_tables = [&](auto cfg, auto&& fn) {
      using C = decltype(cfg);
      constexpr int S_ = C::sch;
      constexpr bool CHAIN = S_ == 1, WIDE = S_ == 2, BYTES = C::bytes;
      using TabT = typename std::conditional<CHAIN, uint16_t, typename std::conditional<WIDE, fx_nib, uint2>::type>::type;
      const FastParams& P = BYTES ? fpb : fp;
      const uint16_t* cm = BYTES ? bmap : cmap;
      const TabT* tabR = CHAIN ? reinterpret_cast<const TabT*>(cm) : (WIDE ? reinterpret_cast<const TabT*>(BYTES ? bwideR : wideR) : reinterpret_cast<const TabT*>(permR));
      const uint8_t* TRp = reinterpret_cast<const uint8_t*>(cm) + 512;
      const uint8_t* TAp = TRp + (CHAIN ? (BYTES ? b_tr : c_tr) : 0u);
      if constexpr (C::sch_a != S_) {   // byte-level tables with FXP_F_BYTE_A8: the forward automaton in the 8-state v_perm format
         static_assert(C::sch_a == 0 && BYTES && S_ == 2, "forward scheme differs: nibble tables backwards, v_perm forwards");
         return fn(tabR, reinterpret_cast<const uint2*>(bpermA), TRp, TAp, P);
      } else {
         const TabT* tabA = CHAIN ? reinterpret_cast<const TabT*>(cm) : (WIDE ? reinterpret_cast<const TabT*>(BYTES ? bwideA : wideA) : reinterpret_cast<const TabT*>(permA));
         return fn(tabR, tabA, TRp, TAp, P);
      }
   };
   // ---- match compaction: this wave's queue and its flush (every lane finishes one queued row from global memory) ----------------
   FxFwdQueue fwdq{fwd_q + (DEFERQ ? wave * 128u : 0u), 0u, 0u, fwd_w + ((DEFERQ && FX_DEFER_STASH != 0) ? wave * 192u : 0u), rows};
   auto flush_fwd = [&]() {
      if constexpr (DEFERQ) {
         if (fwdq.n == 0u) return;
         const bool on = lane < fwdq.n;
         const uint32_t qrow = on ? fwdq.q[lane] : 0u, ge = on ? fwdq.q[64u + lane] : 0u;
         const bool fam_bytes = fwdq.family != 0u;
         fwdq.n = 0u;
         const uint8_t* rp = rows + (int64_t)qrow * (int64_t)L;
         uint32_t s = 0, mm = 0;
         auto finish = [&](auto cfg) {
            constexpr int S_ = decltype(cfg)::sch;
            with_tables(cfg, [&](auto tabR, auto tabA, const uint8_t* TRp, const uint8_t* TAp, const FastParams& P) {
               const uint32_t e = S_ == 0 ? (ge >> 16) * 0x01010101u : (ge >> 16);
               fx_finish_from_global<S_, (CH <= 4 ? 2 : 4), 2, decltype(cfg)::sch_a, (FX_DEFER_STASH != 0)>(tabR, tabA, TRp, TAp, P, rp, L, lane, on, ge & 0xFFFFu, e, s, mm, fwdq.w,
                                                                                                            (FX_DEFER_PREFETCH != 0 && FX_DEFER_STASH == 0 && CH <= 4) ? fwdq.pre : nullptr);
               return 0;
            });
         };
         if constexpr (HAS_B) {
            if (fam_bytes) finish(BCfg{});
            else if constexpr (!ALLB) finish(FxScanCfg<SCH, false, false>{});
         } else finish(FxScanCfg<SCH, false, false>{});
         if (on) {   // api_internal_m.F90:140-148 with a start inside the text (from = s - 1 >= 1)
            const int32_t fr = (int32_t)(s - 1u), tt = mm >= L + 2u ? (int32_t)L : (int32_t)mm - 2;
            if (out_mode == 0u || out_mode == 4u) {
               from[qrow] = fr;
               to[qrow] = tt;
            } else if (out_mode == 1u) {
               reinterpret_cast<uint8_t*>(from)[qrow] = (uint8_t)fr;
               reinterpret_cast<uint8_t*>(to)[qrow] = (uint8_t)tt;
            } else {
               reinterpret_cast<uint16_t*>(from)[qrow] = (uint16_t)fr;
               reinterpret_cast<uint16_t*>(to)[qrow] = (uint16_t)tt;
            }
         }
      }
   };
   uint32_t mgate = 1u;   // `.match.`: the literal / prefix / suffix gate of the row in this lane's cells, evaluated on the raw bytes
   // ---- one scan of the tile in LDS (fx_scan_tile / fx_match_tile) with the tables of one family ---------------------------------
   auto scan = [&](auto cfg, const int64_t row, const bool row_ok, const bool ordered, bool& except) -> bool {
      using C = decltype(cfg);
      constexpr int S_ = C::sch;
      constexpr bool BYTES = C::bytes, DECODED = C::decoded;
      return with_tables(cfg, [&](auto tabR, auto tabA, const uint8_t* TRp, const uint8_t* TAp, const FastParams& P) -> bool {
         if constexpr (MATCH) {
            (void)tabR;
            (void)TRp;
            return fx_match_tile<CH, false, S_, BYTES, DECODED, (HAS_B || !GEN), GEN, RAGGED>(sctx, tabA, TAp, P, h, mgate, row, row_ok, ordered, except, emit);
         } else {
            return fx_scan_tile<CH, SPANS, false, S_, BYTES, DECODED, (HAS_B || !GEN), GEN, false, DEFERQ, C::sch_a, RAGGED>(sctx, tabR, tabA, TRp, TAp, P, row, row_ok,
                                                                                                                  ordered, except, emit, &fwdq, flush_fwd);
         }
      });
   };

   // ---- the wave's loop: tiles of the batch, and -- when its queue would overflow, and at the end -- gathered tiles of exception rows
   uint32_t pool_n = 0;        // rows in this wave's queue (wave-uniform)
   uint64_t pend_mask = 0;     // lanes whose row of the last byte-level scan is an exception not yet queued (wave-uniform)
   uint32_t pend_row = 0;      // that row (per lane)
   // marked-tile mode: does tile t hold a row the first pass left behind (FX_NEEDS_GENERAL)?  wave-uniform
   // (a first pass that writes PACKED results leaves a byte per 64-row tile in `marks` instead: the flag array holds bit words)
   auto tile_marked = [&](const int64_t t) -> bool {
      if (marks != nullptr) return t < n_tiles && __builtin_amdgcn_readfirstlane((uint32_t)marks[t < n_tiles ? t : 0]) != 0u;
      const int64_t rr = (t << 6) + lane;
      return __builtin_amdgcn_ballot_w64(rr < n && flags[rr] == FX_NEEDS_GENERAL) != 0;
   };
   bool live = MARKED ? tile_marked(wave_global) : true;   // the tile in `stage` is to be scanned
   if constexpr (MARKED) {
      if constexpr (RAGGED) load_tile_rag<CH>(stage, rows, wave_global << 6, n, lane, tl, live);
      else load_tile<CH>(stage, rows, wave_global << 6, n, lane, live);
   }
   // speculative pass: on while it pays (wave-uniform).  A tile where more than FX_SPEC_FAIL_MAX rows fail is scanned in place and turns
   // it off; it is tried again FX_SPEC_RETRY tiles later (batches are rarely uniform: sorted inputs, sections of a file).
   bool spec_on = SPEC && (fpb.spec & 1u) != 0u;
   uint32_t spec_wait = 0;     // tiles until the next try
   uint32_t spec_n = 0;        // rows in this wave's queue of rows the speculative pass could not answer (wave-uniform)
   uint64_t spec_mask = 0;     // lanes whose row of the last speculative pass failed and is not yet queued (wave-uniform)
   uint32_t spec_base = 0;     // ... the first row of that tile (wave-uniform: the rows are spec_base + lane)
   ONE_STAMP(0);   // start-up: header + tables -> LDS, barrier, tile columns
   for (int64_t t = wave_global;;) {
      bool is_tile = false;
      bool spec_gather = false;   // the gathered tile holds rows of the speculative pass's queue: byte-level scan, not the decode
      uint32_t take = 0;
      if (POOL && pend_mask != 0 && pool_n + (uint32_t)__builtin_popcountll(pend_mask) <= 64u) {
         if ((pend_mask >> lane) & 1ull) myq[pool_n + (uint32_t)__builtin_popcountll(pend_mask & ((1ull << lane) - 1ull))] = pend_row;
         pool_n += (uint32_t)__builtin_popcountll(pend_mask);
         pend_mask = 0;
      }
      if (SPEC && spec_mask != 0 && spec_n + (uint32_t)__builtin_popcountll(spec_mask) <= 64u) {
         if ((spec_mask >> lane) & 1ull) mysq[spec_n + (uint32_t)__builtin_popcountll(spec_mask & ((1ull << lane) - 1ull))] = spec_base + lane;
         spec_n += (uint32_t)__builtin_popcountll(spec_mask);
         spec_mask = 0;
      }
      if (POOL && pend_mask != 0) {
         take = pool_n;   // the queue has to be drained before the pending rows fit
      } else if (SPEC && spec_mask != 0) {
         take = spec_n;
         spec_gather = true;
      } else if (t < n_tiles) {
         is_tile = true;
      } else {
         // end of the wave's tiles: what is left in its queues (no merging across the block's waves: waiting for the slowest wave
         // behind a barrier cost more than the three partial passes it saved -- config 4: 126.8 -> 118.7 us); the speculative pass's
         // queue first: its scan may add exception rows
         if (SPEC && spec_n != 0u) {
            take = spec_n;
            spec_gather = true;
         } else {
            if (!POOL || pool_n == 0u) break;
            take = pool_n;
         }
      }
      int64_t row;
      bool row_ok;
      bool hint = false;   // a sampled look at the staged bytes found a byte >= 0x80 (wave-uniform)
      if (is_tile) {
         row = (t << 6) + lane;
         row_ok = row < n;
         const bool process = live;
         if (MARKED && !GEN) hint = true;   // (that is why the first pass left the tile; GEN: it may also be a row in the overlap state -- the sampled look decides as in a whole-batch launch)
         else if (!ALLB && !raw && (HAS_B || !GEN)) {
            // (ragged rows: the staging registers behind chunk nch - 1 were not loaded)
            const uint32_t smp = RAGGED ? (stage[0].x | stage[0].y | stage[0].w) : (stage[0].x | stage[0].w | stage[CH / 2].y | stage[CH - 1].z);
            hint = __builtin_amdgcn_ballot_w64((smp & 0x80808080u) != 0) != 0;
         }
         ONE_STAMP_WAIT_VM();
         ONE_STAMP(1);   // wait for the tile's loads (issued one tile of work ago)
         if (process) {
            if constexpr (RAGGED) {
               store_tile_rag<CH>(stage, tile, lane, tl);
               fx_rag_fix_last(tile, rows, t << 6, n, lane, tl);   // (the batch's last tile: the text dword its exact extent cut off)
               fx_tail_patch(tile, lane, tl);
            } else store_tile<CH>(stage, tile, lane);
         }
         ONE_STAMP(14);   // staging registers -> LDS (the stamp waits for the LDS stores' completion: queued behind the other waves' lookups)
         ONE_STAMP_COUNT(12);
         // the ONE place the staging registers are reloaded; a tile behind the last one, or one this pass skips, is "loaded" with
         // zero valid bytes
         t += wave_stride;
         if (MARKED) live = tile_marked(t);
         if constexpr (RAGGED) load_tile_rag<CH>(stage, rows, t << 6, n, lane, tl, live);
         else load_tile<CH>(stage, rows, t << 6, n, lane, live);
         ONE_STAMP(2);   // issue of the next tile's loads
         if (!process) continue;
      } else {
         // gathered tile: lane r loads row queue[r] straight into its own cells (one row per lane: nothing to transpose)
         // packed results: the flag words of this wave's own tiles must be in L2 before a gathered row ORs its bit into one (a wave's
         // queue only ever holds rows of its own tiles): waiting for the stores' acknowledgements is enough -- a device-scope fence
         // would write the whole L2 back (measured: +50 us on a 1M-row batch when every wave did that once)
         if (out_mode != 0u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
         row_ok = lane < take;
         const uint32_t ridx = (POOL && row_ok) ? (spec_gather ? mysq[lane] : myq[lane]) : 0u;
         row = (int64_t)ridx;
         const uint4* src = reinterpret_cast<const uint4*>(rows + row * (int64_t)(16 * CH));
         // (ragged rows: whole chunks as unaligned 16-byte loads, the chunk the row ends in from the row's last 16 bytes; GEN: the general
         //  procedure reads its rows from global memory)
         if constexpr (RAGGED && !GEN) {
            gather_row_rag<CH>(tile, lane, rows + row * (int64_t)L, row_ok, tl);
            fx_tail_patch(tile, lane, tl);
         }
         // (up to twelve loads in flight -- the whole row when it has that many chunks, two rounds of eight at 256 bytes: each round
         //  is a trip to L2 / HBM that the end of the kernel waits for; the staging registers hold the next tile's loads and stay
         //  untouched)
         // (ragged rows -- GEN only -- start at any byte: the general procedure reads them from global memory instead)
         constexpr int GD = CH <= 12 ? CH : 8;
#pragma unroll 1
         for (int k0 = 0; k0 < (RAGGED ? 0 : CH); k0 += GD) {
            uint4 g4[GD];
#pragma unroll
            for (int i = 0; i < GD; ++i) g4[i] = (row_ok && k0 + i < CH) ? src[k0 + i < CH ? k0 + i : 0] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < GD; ++i)
               if (k0 + i < CH) tile[tile_cell(lane, k0 + i)] = g4[i];
         }
         if (spec_gather) spec_n = 0;
         else pool_n = 0;
         ONE_STAMP(5);   // gather of queued rows: global memory -> LDS (issued; the wait lands in the scan that reads the cells)
         ONE_STAMP_COUNT(13);
      }
      if constexpr (MATCH) mgate = match_gate(h, prog, tb, lane, L);   // (on the raw bytes: before any decode rewrites the cells)
      bool except = false;
      bool redo = false;
      // the OR of this lane's row (the tile in LDS): "any byte >= 0x80" tests of the FXP_F_NEEDS_NONASCII shortcuts
      auto row_or = [&]() {
         uint32_t na = 0;
         if constexpr (RAGGED) {   // text bytes only: the NUL / KILL symbols behind the text are not the row's
            const FxTail T = fx_tail_here(tl);
            for (uint32_t k = 0; k < T.kt; ++k) {
               const uint4 c = tile[tile_cell(lane, k)];
               na |= c.x | c.y | c.z | c.w;
            }
            if (T.nb != 0u) na |= fx_tail_or(tile[tile_cell(lane, T.kt)], T.nb);
         } else {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
               const uint4 c = tile[tile_cell(lane, k)];
               na |= c.x | c.y | c.z | c.w;
            }
         }
         return na;
      };
      if constexpr (HAS_B && !MATCH && !GEN) {
         // FXP_F_NEEDS_NONASCII: no match is made of ASCII symbols only -- a pure-ASCII tile holds none (60 instructions instead of a scan)
         if (is_tile && !ALLB && !hint && (fpb.spec & 2u) != 0u) {
            if (__builtin_amdgcn_ballot_w64((row_or() & 0x80808080u) != 0u) == 0) {
               emit(row, row_ok, true, 0u, 0, 0, true);
               continue;
            }
            hint = true;
         }
      }
      if (is_tile && !ALLB && !hint) {
         redo = scan(FxScanCfg<SCH, false, false>{}, row, row_ok, true, except);
         if (GEN && !redo) {
            pend_mask = __builtin_amdgcn_ballot_w64(except && row_ok);
            pend_row = (uint32_t)row;
         }
         ONE_STAMP(3);   // class-level scan of a pure-ASCII tile (or up to the byte >= 0x80 that gives it back)
      }
      const bool nonascii = !is_tile || hint || redo;   // (gathered rows always take the decode)
      // A FEW rows the speculative walk could not answer -- the usual end of a wave: config 4 queues five per wave, its rows with a corrupted byte -- skip
      // the byte-level scan (a 64-lane backward + forward pass for a handful of rows, which then finds the structure error and queues them AGAIN for a
      // second gather): the in-LDS decode + the chunk-parallel scan of fx_few.hpp answer every row, broken or not, in a fraction of the instructions
      // (round 6; FX_SPEC_FEW_GROUPS groups of 64 / CH rows)
      // (not the whole-batch kernels of 256-byte rows: with the chunk-parallel scan's 32 table registers they needed 260-262 registers -- ONE wave
      //  per SIMD where the comment below promised two; without it 228.  Its marked-tile sibling, the half-row pipeline's follow-up, has both at 239.)
      constexpr bool FEW_OK = SCH == 0 && HAS_B && !MATCH && CH >= 12 && FX_FEW_ROWS != 0 && !RAGGED && !GEN && !(CH == 16 && !MARKED);
      const bool spec_few = FEW_OK && SPEC && spec_gather && take <= (uint32_t)FX_SPEC_FEW_GROUPS * fx_few_rows_max<CH>();
      if constexpr (HAS_B) {
         if ((is_tile && (ALLB || nonascii)) || (spec_gather && !spec_few)) {
            bool done = false;
            if constexpr (SPEC) {
               if (is_tile && !spec_on && (fpb.spec & 1u) != 0u && ++spec_wait >= (uint32_t)FX_SPEC_RETRY) spec_on = true;
               if (is_tile && spec_on) {
                  const uint32_t mm = fx_spec_forward<CH, SPANS>(tile, tb, lane, reinterpret_cast<const uint2*>(bpermA), fpb, row_ok);
                  const bool ok = mm != 0u;
                  uint64_t failm = __builtin_amdgcn_ballot_w64(row_ok && !ok);
                  bool no_match = false;   // FXP_F_NEEDS_NONASCII: a failed row without a byte >= 0x80 holds no match at all
                  if (failm != 0 && (fpb.spec & 2u) != 0u) {
                     no_match = !ok && (row_or() & 0x80808080u) == 0u;
                     failm = __builtin_amdgcn_ballot_w64(row_ok && !ok && !no_match);
                  }
                  if ((uint32_t)__builtin_popcountll(failm) <= (uint32_t)FX_SPEC_FAIL_MAX) {
                     // api_internal_m.F90:140-148 with start = 2 (the row's first character): from = 1, to = max_match - 2, clamped to the row
                     const int32_t tt = mm >= L + 2u ? (int32_t)L : (int32_t)mm - 2;
                     emit(row, row_ok && (ok || no_match), true, ok ? 1u : 0u, ok ? 1 : 0, ok ? tt : 0, true);
                     spec_mask = failm;
                     spec_base = __builtin_amdgcn_readfirstlane((uint32_t)row - lane);
                     done = true;
                  } else {
                     spec_on = false;   // too many rows start their match elsewhere (or hold none): the full scan, in place, for all of them
                     spec_wait = 0;
                  }
                  ONE_STAMP(4);   // speculative forward walk + its results
               }
            }
            if (!done) {
               (void)scan(BCfg{}, row, row_ok, is_tile, except);
               pend_mask = __builtin_amdgcn_ballot_w64(except && row_ok);
               pend_row = (uint32_t)row;
               if (is_tile) ONE_STAMP(6);   // byte-level backward + forward scan of a tile in place
               else ONE_STAMP(7);           // ... of a gathered tile (rows the speculative walk could not answer)
            }
         }
      }
      if constexpr (GEN) {
         if (!is_tile && row_ok) {
            // the general row procedure (every mode, candidate-list driver, UTF-8 decode in both directions): one lane = one queued row
            fxrow::ProgView pv(prog);
            fxrow::DfaSim sim(pv);
            fxrow::Result res;
            if (RAGGED) {
               FxGlobalRow gr{rows + row * (int64_t)L};
               fxrow::run_row(pv, sim, gr, (int)L, res);
            } else {
               FxTileRow tr{tb, lane};
               fxrow::run_row(pv, sim, tr, (int)L, res);
            }
            emit(row, true, false, res.flag, res.from, res.to);
         }
         if (!is_tile) ONE_STAMP(10);   // general row procedure over gathered rows
      }
      if (!GEN && (!spec_gather || spec_few) && ((HAS_B && !is_tile) || (!HAS_B && is_tile && nonascii))) {
         // On-device UTF-8 decode, in place in LDS, into fast-path symbol ids (fxrow::translate_cell16); the 4 bytes before / after
         // a cell are taken from the ORIGINAL neighbours.
         if (HAS_B && !is_tile) {
            // gathered tile: usually a handful of rows (what a wave's queue holds at the end), and a single lane's serial decode of a row is
            // what the end of the kernel waits for -- so FOUR lanes share a row (a quarter of its cells each), sixteen rows per round.
            // All reads of a round are issued before its writes (one wave: its LDS operations complete in order).
            constexpr int CQ = (CH + 3) / 4;
            const uint32_t q = lane & 3u;
            // (a few rows only -- the usual end of a wave: ONE cell per lane, 64 / CH whole rows per round, when that takes fewer rounds
            //  than a lane of the four has cells)
            constexpr uint32_t RPR = 64u / (uint32_t)CH;   // rows per round
            const bool one_cell = (take + RPR - 1u) / RPR < (uint32_t)CQ;
            for (uint32_t r0 = 0; r0 < (one_cell ? take : 0u); r0 += RPR) {
               const uint32_t r = r0 + lane / (uint32_t)CH, k = lane % (uint32_t)CH;
               const bool on = lane < RPR * (uint32_t)CH && r < take;
               const uint32_t prev = (on && k >= 1u) ? tile[tile_cell(r, k - 1u)].w : 0u;
               const uint4 cur = on ? tile[tile_cell(r, k)] : make_uint4(0, 0, 0, 0);
               const uint32_t nxt = (on && k + 1u < (uint32_t)CH) ? tile[tile_cell(r, k + 1u)].x : 0u;
               const fxrow::Cell16 o = fxrow::translate_cell16(prev, cur.x, cur.y, cur.z, cur.w, nxt, ct, sym_ffff);
               if (on) tile[tile_cell(r, k)] = make_uint4(o.x, o.y, o.z, o.w);
            }
            for (uint32_t r0 = 0; r0 < (one_cell ? 0u : take); r0 += 16u) {
               const uint32_t r = r0 + (lane >> 2);
               uint4 outc[CQ];
               uint32_t k = q * CQ;
               uint32_t prev = (k >= 1u && k - 1u < (uint32_t)CH) ? tile[tile_cell(r, k - 1u)].w : 0u;
               uint4 cur = k < (uint32_t)CH ? tile[tile_cell(r, k)] : make_uint4(0, 0, 0, 0);
#pragma unroll
               for (int i = 0; i < CQ; ++i, ++k) {
                  const uint4 nxt = k + 1u < (uint32_t)CH ? tile[tile_cell(r, k + 1u)] : make_uint4(0, 0, 0, 0);
                  const fxrow::Cell16 o = fxrow::translate_cell16(prev, cur.x, cur.y, cur.z, cur.w, nxt.x, ct, sym_ffff);
                  outc[i] = make_uint4(o.x, o.y, o.z, o.w);
                  prev = cur.w;
                  cur = nxt;
               }
               k = q * CQ;
#pragma unroll
               for (int i = 0; i < CQ; ++i, ++k)
                  if (k < (uint32_t)CH) tile[tile_cell(r, k)] = outc[i];
            }
         } else {
            uint32_t prev = 0;   // lane r rewrites its own row cell by cell
            uint4 cur = tile[tile_cell(lane, 0)];
            const int kd = RAGGED ? (int)tl.nch : CH;   // (ragged rows: the chunks that hold text)
            for (int k = 0; k < kd; ++k) {
               const uint4 nxt = k + 1 < CH ? tile[tile_cell(lane, k + 1)] : make_uint4(0, 0, 0, 0);
               const fxrow::Cell16 o = fxrow::translate_cell16(prev, cur.x, cur.y, cur.z, cur.w, nxt.x, ct, sym_ffff);
               tile[tile_cell(lane, k)] = make_uint4(o.x, o.y, o.z, o.w);
               prev = cur.w;
               cur = nxt;
            }
         }
         if (RAGGED) {   // the decode rewrote the NUL / KILL symbols behind the text as well (0xFE is a broken byte to it): put them back
            fx_tail_init<CH>(tile, lane, tl);
            fx_tail_patch(tile, lane, tl);
         }
         ONE_STAMP(8);   // in-LDS UTF-8 decode
         // a gathered tile of a FEW rows -- the usual end of a wave -- on the 8-state tables: the lanes share the rows' cells (fx_few.hpp).
         // (Rows of 192 / 256 bytes only: those kernels run two waves per SIMD whatever their registers; with the 32 registers of a cell's
         //  table rows the kernels of shorter rows would drop from three waves per SIMD to two -- CH 8: 163 -> 191 VGPRs.)
         if constexpr (FEW_OK) {
            if (!is_tile && take <= (uint32_t)FX_SPEC_FEW_GROUPS * fx_few_rows_max<CH>()) {
               for (uint32_t r0 = 0; r0 < take; r0 += fx_few_rows_max<CH>())
                  fx_scan_few_rows<CH, SPANS>(tile, permR, permA, fp, lane, take, spec_gather ? mysq : myq, emit, r0);
               ONE_STAMP(9);   // scan of the decoded rows (few rows: chunk-parallel)
               continue;
            }
         }
         (void)scan(FxScanCfg<SCH, false, true>{}, row, row_ok, is_tile, except);
         ONE_STAMP(9);
      }
   }
   flush_fwd();
   ONE_STAMP(11);   // match-compaction flush (and the loop's last bookkeeping)
   ONE_STAMP_FLUSH;
}

// FastParams of the class-level tables (fp) and of the byte-level tables (fpb) are prepared by the host (fxamd.hip)
// Blocks of `fn` (256 threads, `lds` bytes of dynamic LDS) that one CU holds at a time, as the runtime computes it from the code object
// (registers AND LDS -- config 2's kernel fits five blocks by LDS but four by VGPRs: a grid of 5 x 256 would run a second, quarter-full
// round).  One query per (kernel, LDS size) and device; `fallback` when the query fails.
inline int resident_blocks_per_cu(const void* fn, size_t lds, int fallback) {
   struct Key {
      const void* fn;
      size_t lds;
      int dev;
      int blocks;
   };
   static std::mutex mu;
   static std::vector<Key> seen;
   int dev = 0;
   if (hipGetDevice(&dev) != hipSuccess) return fallback;
   std::lock_guard<std::mutex> g(mu);
   for (const Key& k : seen)
      if (k.fn == fn && k.lds == lds && k.dev == dev) return k.blocks;
   int nb = 0;
   if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = fallback;
   }
   seen.push_back(Key{fn, lds, dev, nb});
   return nb;
}

template <int CH, int SCH, int BSCH, bool GEN>
hipError_t launch_one(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, FastParams fpb, uint8_t* flags, int32_t* from, int32_t* to,
                      uint32_t class_map_bytes, uint32_t table_bytes, uint32_t Lr, hipStream_t st, uint32_t out_mode, bool is_match) {
   const int64_t n_tiles = (n + 63) >> 6;
   int64_t blocks = (n_tiles + 3) / 4;
   const size_t tiles_b = (size_t)4 * 64 * (CH + 1) * 16;
   // the BMP class map rides behind the tables when two blocks per CU still fit (else the decode reads it from global memory)
   const size_t static_b = (SCH == 0 ? 4096 : (SCH == 2 ? 4096 : 0)) + ((BSCH == 2 || BSCH == 3) ? 4096 : 0) + 1024 + 64 + ((FX_DEFER_FWD != 0 && CH <= 4) ? 2048 + (FX_DEFER_STASH != 0 ? 6144 : 0) : 0) +
                           ((FX_SPEC_FWD != 0 && BSCH == 3) ? 1024 : 0);
   const uint32_t map_lds = (!GEN && tiles_b + table_bytes + class_map_bytes + static_b <= 80 * 1024 && class_map_bytes <= 24u * 1024u) ? class_map_bytes : 0u;
   const size_t lds = tiles_b + table_bytes + map_lds;
   const bool ragged = Lr != 16u * CH;
   const bool spans = from && to;   // (packed results: `from` / `to` are the narrow arrays)
   // Grid: with exception queues every wave ends with one pass over its queued rows, so the grid is sized to what is
   // RESIDENT (one tail per CU slot, not one per 1/8 of it); without them the usual cap with grid-stride beyond it.
   // Grid: whole rounds of what is RESIDENT (asked of the runtime per kernel and LDS size).  How many rounds: a block costs its table
   // staging and a first tile without overlap, and with exception queues every wave ends with one pass over its queued rows -- against
   // that, a finer grid lets the hardware's block scheduler balance the tail.  Measured (profiles/r03_grid_ab.txt, interleaved
   // repetitions in one allocation): config 5's shard (1.6 GB) 372-374 us with one round, 352-366 us with four, 356-367 us with seven or eight;
   // configs 2 and 4 (67 / 201 MB: 19 / 101 us in all) are best with ONE round (config 4 with six: 144 us).  So: one round per
   // 300 MB of rows.  FXAMD_ONE_GRID = blocks per CU in the grid, FXAMD_ONE_ROUND_MB = MB per round: experiment hooks.
   auto cap_grid = [&](const void* fn) {
      const int env_mult = fx_env().one_grid;
      const size_t per_block = lds + static_b;
      int64_t by_lds = per_block > 0 ? (int64_t)((160 * 1024) / per_block) : 8;   // blocks per CU by LDS alone (used when the runtime cannot say)
      if (by_lds < 1) by_lds = 1;
      if (by_lds > 8) by_lds = 8;
      int64_t resident = resident_blocks_per_cu(fn, lds, (int)by_lds);
      if (resident > 8) resident = 8;
      const int env_mb = fx_env().one_round_mb;
      int64_t rounds = (n * (int64_t)Lr) / ((int64_t)(env_mb > 0 ? env_mb : 300) << 20);
      if (rounds < 1) rounds = 1;
      if (rounds > 64) rounds = 64;
      const int64_t cap = 256 * (env_mult > 0 ? env_mult : resident * rounds);
      if (blocks > cap) blocks = cap;
      const int env_blocks = fx_env().one_blocks;   // FXAMD_ONE_BLOCKS, test hook: a tiny grid, many tiles per wave
      if (env_blocks > 0 && blocks > env_blocks) blocks = env_blocks;
   };
   auto go = [&](auto kern) -> hipError_t {
      const void* fn = reinterpret_cast<const void*>(kern);
      if (lds > 64 * 1024) {
         hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
         if (e != hipSuccess) return e;
      }
      cap_grid(fn);
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, fpb, flags, from, to, map_lds, Lr, out_mode, (const uint32_t*)nullptr, (const uint8_t*)nullptr);
      return hipGetLastError();
   };
   constexpr bool RAG_OK = (CH & (CH - 1)) == 0;   // ragged rows run on the power-of-two instantiations (fxamd.hip: one_chunks)
   if (ragged && !RAG_OK) return hipErrorInvalidValue;
   if (is_match) {   // `.match.`: one verdict per row, no span
      if constexpr (BSCH == 3 || GEN) return hipErrorInvalidValue;   // (never dispatched: FXP_F_BYTE_A8 is a search program's table; `.match.` programs that cannot decode take the multi-pass pipeline)
      else {
         if constexpr (RAG_OK)
            if (ragged) return go(&fx_search_one<CH, false, SCH, BSCH, true, GEN, false, true>);
         return go(&fx_search_one<CH, false, SCH, BSCH, false, GEN, false, true>);
      }
   }
   if constexpr (RAG_OK)
      if (ragged) return spans ? go(&fx_search_one<CH, true, SCH, BSCH, true, GEN>) : go(&fx_search_one<CH, false, SCH, BSCH, true, GEN>);
   return spans ? go(&fx_search_one<CH, true, SCH, BSCH, false, GEN>) : go(&fx_search_one<CH, false, SCH, BSCH, false, GEN>);
}

// the gated follow-up of the half-row first pass (256-byte rows) and of the span kernel (fx_span.hpp: rows of 128 / 64 / 32 / 16 bytes, spans
// only): 8-state class-level tables, marked tiles only.  GEN: programs whose class-level tables cannot decode UTF-8 (the general row
// procedure for the rows the tables cannot answer, inside the launch)
template <int CH, int BSCH, bool GEN>
hipError_t launch_one_marked(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, FastParams fpb, uint8_t* flags, int32_t* from, int32_t* to,
                             uint32_t class_map_bytes, uint32_t table_bytes, hipStream_t st, const uint32_t* gate, uint32_t out_mode, const uint8_t* marks, uint32_t Lr) {
   static_assert(CH == 16 || CH == 8 || CH == 4 || CH == 2 || CH == 1, "follow-up of the half-row first pass (256-byte rows) and of the span kernel");
   static_assert(CH != 16 || !GEN, "the half-row pipeline is for programs whose tables decode");
   const size_t tiles_b = (size_t)4 * 64 * (CH + 1) * 16;
   const size_t static_b = 4096 + ((BSCH == 2 || BSCH == 3) ? 4096 : 0) + 1024 + 64;
   const uint32_t map_lds = (!GEN && tiles_b + table_bytes + class_map_bytes + static_b <= 80 * 1024 && class_map_bytes <= 24u * 1024u) ? class_map_bytes : 0u;
   const size_t lds = tiles_b + table_bytes + map_lds;
   const int64_t n_tiles = (n + 63) >> 6;
   int64_t blocks = (n_tiles + 3) / 4;
   if (blocks > 256 * 2) blocks = 256 * 2;   // what is resident: an empty follow-up is one round of blocks that leave at once
   const bool spans = from && to;
   auto go = [&](auto kern) -> hipError_t {
      if (lds > 64 * 1024) {
         hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
         if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, fpb, flags, from, to, map_lds, Lr, out_mode, gate, marks);
      return hipGetLastError();
   };
   if constexpr (BSCH == 3 && GEN) return hipErrorInvalidValue;   // (never dispatched: the speculative pass's table is for programs that decode)
   else {
      if (Lr != 16u * CH) {   // ragged rows (the span kernel's): any length 2 <= Lr < 16 * CH on the power-of-two instantiations
         if constexpr (CH != 16) {
            if (spans && Lr >= 2u && Lr < 16u * CH) return go(&fx_search_one<CH, true, 0, BSCH, true, GEN, true>);
         }
         return hipErrorInvalidValue;
      }
      if (spans) return go(&fx_search_one<CH, true, 0, BSCH, false, GEN, true>);
      if constexpr (CH == 16) return go(&fx_search_one<CH, false, 0, BSCH, false, GEN, true>);
      return hipErrorInvalidValue;   // (the span kernel answers searches with spans only)
   }
}
#define FX_ONE_MARKED_SIG (const uint8_t*, int64_t, const uint8_t*, FastParams, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t, uint32_t, hipStream_t, const uint32_t*, uint32_t, const uint8_t*, uint32_t)

// every (CH, SCH, BSCH, GEN) the dispatch code of fxamd.hip can ask for
#define FX_ONE_COMBOS_G(X, CH, G) X(CH, 0, 0, G) X(CH, 1, 0, G) X(CH, 2, 0, G) X(CH, 0, 1, G) X(CH, 0, 2, G) X(CH, 0, 3, G) X(CH, 1, 1, G) X(CH, 1, 2, G) X(CH, 2, 1, G) X(CH, 2, 2, G)
#define FX_ONE_COMBOS(X, CH) FX_ONE_COMBOS_G(X, CH, false) FX_ONE_COMBOS_G(X, CH, true)
#define FX_ONE_ALL(X) \
   FX_ONE_COMBOS(X, 1) FX_ONE_COMBOS(X, 2) FX_ONE_COMBOS(X, 4) FX_ONE_COMBOS(X, 8) FX_ONE_COMBOS(X, 12) FX_ONE_COMBOS(X, 16)
#define FX_ONE_SIG (const uint8_t*, int64_t, const uint8_t*, FastParams, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t, uint32_t, uint32_t, hipStream_t, uint32_t, bool)
