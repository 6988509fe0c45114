// fx_search_multi: m patterns against the same rows in ONE pass over the rows (the reference's operators are
// elemental over `pattern` too, src/forgex.F90:74,163).  A tile of 64 rows is staged in LDS once; the tables of all m patterns
// (8-state v_perm scheme or, round 6, the nibble scheme of automata with 9..16 states: 4 KB each) sit in LDS next to the tiles, and the wave that owns the tile scans it once per pattern
// (fx_scan_tile), writing pattern p's results to flags[p*n + row] (from / to likewise).  HBM traffic: the rows once plus m result
// sets, instead of m times the rows.  Patterns that have byte-level tables in the nibble format (FXP_F_BYTE_W16; bsch 2, or 3 with the
// forward automaton in the v_perm format) bring them along (8 KB per pattern instead of 4): a tile with a byte >= 0x80 is scanned for
// them right here, on the raw bytes, instead of being deferred to a pass of their own that reads the rows again.
// The kernel is the FIRST PASS of every pattern's multi-pass pipeline (what fx_search_fast's MODE 0 does), sharing the staging:
// a tile with a byte >= 0x80 is deferred for the patterns that have a later pass for such tiles (its rows are marked
// FX_NEEDS_GENERAL in that pattern's flags, the pattern's "deferred" word is set), and rows a pattern's tables cannot answer -- a
// byte >= 0x80 when the pattern has no such pass, the overlap state of a bordered prefix literal -- are appended to that
// pattern's worklist.  The host then enqueues each pattern's own (gated, usually empty) follow-up passes.
// The scans are VALU-bound (about 3.4 instructions per byte per pattern), so the pass costs about m times the compute of one
// pattern; what is saved is the memory time of the other m-1 passes.
#pragma once
#include "fx_one.hpp"

#define FX_MULTI_MAX 8
struct FxMultiArgs {
   const uint8_t* blob[FX_MULTI_MAX];   // the patterns' uploaded program images
   FastParams fp[FX_MULTI_MAX];         // class-level parameters of each (v_perm scheme)
   uint32_t slot[FX_MULTI_MAX];         // result slot of each: its flags start at flags + slot * n (from / to likewise)
   uint32_t* ctr[FX_MULTI_MAX];         // this call's counter group of each ([0] tiles deferred, [1] rows in its worklist)
   uint32_t* worklist[FX_MULTI_MAX];    // rows left to each pattern's fix-up
   uint32_t defer_tiles[FX_MULTI_MAX];  // 1: a later pass of this pattern takes whole tiles that hold a byte >= 0x80
   FastParams fpb[FX_MULTI_MAX];        // byte-level parameters of each (bsch != 0)
   uint32_t bsch[FX_MULTI_MAX];         // 0: no byte-level tables in this pass; 2: nibble tables; 3: nibble backwards, v_perm forwards (FXP_F_BYTE_A8)
   uint32_t any_bytes;                  // some pattern has bsch != 0: the table area holds 8 KB per pattern (else 4 KB)
   uint32_t inq[FX_MULTI_MAX];          // 1: exception rows of this pattern's byte-level scan are finished INSIDE this launch (its class-level tables decode UTF-8)
   uint32_t sch[FX_MULTI_MAX];          // class-level table scheme of each: 0 = 8-state v_perm tables, 2 = nibble tables (9..16 states; round 6) -- 4 KB of LDS per pattern either way
   uint32_t m;
};

template <int CH, bool SPANS, bool RAGGED, int WPB>
__global__ __launch_bounds__(64 * WPB) void fx_search_multi(const uint8_t* __restrict__ rows, int64_t n, FxMultiArgs a, uint8_t* __restrict__ flags,
                                                              int32_t* __restrict__ from, int32_t* __restrict__ to, uint32_t Lr) {
   const uint32_t L = RAGGED ? Lr : 16u * CH;
   extern __shared__ __attribute__((aligned(16))) uint4 tiles[];   // WPB waves x 64*(CH+1) cells, then m x (permR[256], permA[256] [, bwideR[256], bwideA / b8A[256]])
   uint2* tabs = reinterpret_cast<uint2*>(tiles + WPB * 64 * (CH + 1));
   const uint32_t tstride = a.any_bytes ? 1024u : 512u;   // 8-byte entries per pattern
   // behind the tables (any_bytes only): per wave one queue of 64 exception rows (row, pattern) of ALL patterns, and the patterns'
   // parameters where a lane can index them by its own pattern
   uint32_t* xq_all = reinterpret_cast<uint32_t*>(tabs + (size_t)a.m * tstride);
   uint32_t* xpar = xq_all + WPB * 128;   // [FX_MULTI_MAX][16]: R_start, A_init, hit_min, acc_min, inv, inv_on, lit_len, slot, blob lo, blob hi
   if (a.any_bytes && threadIdx.x < a.m) {
      const uint32_t q = threadIdx.x;
      xpar[q * 16 + 0] = a.fp[q].R_start;
      xpar[q * 16 + 1] = a.fp[q].A_init;
      xpar[q * 16 + 2] = a.fp[q].hit_min;
      xpar[q * 16 + 3] = a.fp[q].acc_min;
      xpar[q * 16 + 4] = a.fp[q].inv;
      xpar[q * 16 + 5] = a.fp[q].inv_on;
      xpar[q * 16 + 6] = a.fp[q].lit_len;
      xpar[q * 16 + 7] = a.slot[q];
      xpar[q * 16 + 8] = (uint32_t)reinterpret_cast<uintptr_t>(a.blob[q]);
      xpar[q * 16 + 9] = (uint32_t)(reinterpret_cast<uintptr_t>(a.blob[q]) >> 32);
   }
   for (uint32_t p = 0; p < a.m; ++p) {
      const FxpHeader* h = reinterpret_cast<const FxpHeader*>(a.blob[p]);
      const uint2* gR = reinterpret_cast<const uint2*>(a.blob[p] + (a.sch[p] == 2u ? h->off_w16R : h->off_fastR));
      const uint2* gA = reinterpret_cast<const uint2*>(a.blob[p] + (a.sch[p] == 2u ? h->off_w16A : h->off_fastA));
      const uint2* bR = reinterpret_cast<const uint2*>(a.blob[p] + h->off_bw16R);
      const uint2* bA = reinterpret_cast<const uint2*>(a.blob[p] + (a.bsch[p] == 3u ? h->off_b8A : h->off_bw16A));
      for (uint32_t i = threadIdx.x; i < 256u; i += 64u * WPB) {
         tabs[p * tstride + i] = gR[i];
         tabs[p * tstride + 256u + i] = gA[i];
         if (a.bsch[p] != 0u) {
            tabs[p * tstride + 512u + i] = bR[i];
            tabs[p * tstride + 768u + i] = bA[i];
         }
      }
   }
   // the counter words of consecutive calls alternate: this call zeroes the other group of every pattern (see fx_search_fast)
   if (blockIdx.x == 0 && threadIdx.x < a.m) {
      uint32_t* other = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(a.ctr[threadIdx.x]) ^ 16u);
      other[0] = 0u;
      other[1] = 0u;
      other[2] = 0u;
      other[3] = 0u;
   }
   __syncthreads();
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint4* tile = tiles + wave * (64 * (CH + 1));
   tile[tile_cell(lane, CH)] = make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);   // end-of-row column (see fx_search_fast)
   const bool whole = RAGGED && (Lr & 15u) == 0u;
   if (whole)
      for (uint32_t k = Lr >> 4; k < (uint32_t)CH; ++k) tile[tile_cell(lane, k)] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   const int64_t n_tiles = (n + 63) >> 6;
   const int64_t wave_global = (int64_t)blockIdx.x * WPB + wave, wave_stride = (int64_t)gridDim.x * WPB;
   uint32_t deferred_any = 0;   // bit p: this wave deferred a tile for pattern p
   // ---- exception rows finished inside the launch: ONE queue per wave for all patterns.  A drain gathers the queued rows into the
   //      tile (lane r = entry r), decodes each row with ITS pattern's class map, scans it with ITS pattern's class-level tables
   //      (per-lane table base, per-lane parameters) and stores the result in its pattern's slot.  Drained when full -- after the
   //      current tile's last pattern: the tile buffer is free then -- and when the wave has finished its tiles.
   uint32_t* const xq = xq_all + wave * 128u;
   uint32_t xq_n = 0;
   auto drain = [&]() {
      if (xq_n == 0u) return;
      const bool on = lane < xq_n;
      const uint32_t qrow = on ? xq[lane] : 0u, qp = on ? xq[64u + lane] : 0u;
      xq_n = 0;
      if constexpr (!RAGGED) {
         const uint4* src = reinterpret_cast<const uint4*>(rows + (int64_t)qrow * (int64_t)(16 * CH));
         constexpr int GD = CH <= 12 ? CH : 8;
#pragma unroll 1
         for (int k0 = 0; k0 < CH; k0 += GD) {
            uint4 g4[GD];
#pragma unroll
            for (int i = 0; i < GD; ++i) g4[i] = (on && k0 + i < CH) ? src[k0 + i < CH ? k0 + i : 0] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < GD; ++i)
               if (k0 + i < CH) tile[tile_cell(lane, k0 + i)] = g4[i];
         }
         const uint32_t* par = xpar + qp * 16u;
         const uint8_t* blob = reinterpret_cast<const uint8_t*>(((uint64_t)par[9] << 32) | par[8]);
         const FxpHeader* h = reinterpret_cast<const FxpHeader*>(blob);
         const fxrow::ClassTables ct{reinterpret_cast<const uint16_t*>(blob + h->off_cls_page), reinterpret_cast<const uint16_t*>(blob + h->off_cls_pages),
                                     reinterpret_cast<const uint16_t*>(blob + h->off_bound_cls), reinterpret_cast<const int32_t*>(blob + h->off_bounds), h->n_bounds};
         const uint32_t sym_ffff = 128u + h->cls_ffff;
         {   // on-device UTF-8 decode of this lane's row, in place, into the pattern's fast-path symbol ids
            uint32_t prev = 0;
            uint4 cur = tile[tile_cell(lane, 0)];
            for (int k = 0; k < CH; ++k) {
               const uint4 nxt = k + 1 < CH ? tile[tile_cell(lane, k + 1)] : make_uint4(0, 0, 0, 0);
               const fxrow::Cell16 o = fxrow::translate_cell16(prev, cur.x, cur.y, cur.z, cur.w, nxt.x, ct, sym_ffff);
               tile[tile_cell(lane, k)] = make_uint4(o.x, o.y, o.z, o.w);
               prev = cur.w;
               cur = nxt;
            }
         }
         const FastParams P{par[0], par[1], par[2], par[3], par[4], par[5], 0u, 0u, par[6], 0u, 0u};
         const int64_t base = (int64_t)par[7] * n;
         auto emit = [&](const int64_t r, const bool live, const bool, const uint32_t flag, const int32_t fr, const int32_t tt, const bool) {
            if (live) {
               flags[base + r] = (uint8_t)flag;
               if (SPANS) {
                  from[base + r] = fr;
                  to[base + r] = tt;
               }
            }
         };
         FxScanCtx scx{tile, tb, lane, L, Lr, false, false, 0u};
         const uint2* tR = tabs + qp * tstride;
         bool except = false;
         (void)fx_scan_tile<CH, SPANS, false, 0, false, true, false, false, false>(scx, tR, tR + 256, nullptr, nullptr, P, (int64_t)qrow, on, false, except, emit);
      }
   };
   uint4 stage[CH];
   if (RAGGED) load_tile<CH>(stage, rows, wave_global << 6, n, lane, true, Lr);
   else load_tile<CH>(stage, rows, wave_global << 6, n, lane);
   for (int64_t t = wave_global; t < n_tiles;) {
      const int64_t row = (t << 6) + lane;
      const bool row_ok = row < n;
      if (RAGGED && (Lr & 15u) == 0u) store_tile_rt<CH>(stage, tile, lane, Lr >> 4);
      else if (RAGGED) store_tile_relayout<CH>(stage, tile, lane, Lr);
      else store_tile<CH>(stage, tile, lane);
      t += wave_stride;   // the ONE place the staging registers are reloaded
      if (RAGGED) load_tile<CH>(stage, rows, t << 6, n, lane, true, Lr);
      else load_tile<CH>(stage, rows, t << 6, n, lane);
      // once per tile: pad rows that are not whole chunks, and the OR of the row's own bytes (every pattern asks for it)
      FxScanCtx sc{tile, tb, lane, L, Lr, whole, false, 0u};
      if (RAGGED && !whole) sc.pre_na = pad_rows<CH>(tile, lane, Lr);
      else {
         uint32_t na = 0;
#pragma unroll
         for (int k = 0; k < CH; ++k) {
            if (!RAGGED || (uint32_t)k < (Lr >> 4)) {
               const uint4 c = tile[tile_cell(lane, k)];
               na |= c.x | c.y | c.z | c.w;
            }
         }
         sc.pre_na = na;
      }
      const bool tile_hi = __builtin_amdgcn_ballot_w64((sc.pre_na & 0x80808080u) != 0) != 0;
      uint64_t pend[FX_MULTI_MAX];   // per pattern: lanes whose row is an exception to be finished in this launch (wave-uniform)
      uint32_t pend_any = 0;
      for (uint32_t p = 0; p < a.m; ++p) {
         const int64_t base = (int64_t)a.slot[p] * n;
         const FxpHeader* h = reinterpret_cast<const FxpHeader*>(a.blob[p]);
         sc.raw = (h->flags & FXP_F_RAW_BYTES) != 0;
         const bool bytes_here = !RAGGED && tile_hi && !sc.raw && a.bsch[p] != 0u;   // scanned right here with the pattern's byte-level tables
         if (tile_hi && !sc.raw && !bytes_here && a.defer_tiles[p] != 0u) {   // left whole to this pattern's pass over deferred tiles
            if (row_ok) flags[base + row] = FX_NEEDS_GENERAL;
            deferred_any |= 1u << p;
            continue;
         }
         auto emit = [&](const int64_t r, const bool live, const bool, const uint32_t flag, const int32_t fr, const int32_t tt, const bool) {
            if (live) {
               flags[base + r] = (uint8_t)flag;
               if (SPANS) {
                  from[base + r] = fr;
                  to[base + r] = tt;
               }
            }
         };
         const uint2* tR = tabs + p * tstride;
         const uint2* tA = tR + 256;
         bool except = false;
         if constexpr (!RAGGED) {
            if (bytes_here) {   // raw bytes, nibble tables backwards; forwards nibble or (bsch 3) 8-state v_perm; INVALID rows are listed below
               const fx_nib* bR = reinterpret_cast<const fx_nib*>(tR + 512);
               if (a.bsch[p] == 3u)
                  (void)fx_scan_tile<CH, SPANS, false, 2, true, false, false, false, true, false, 0>(sc, bR, tR + 768, nullptr, nullptr, a.fpb[p], row, row_ok, true, except, emit);
               else
                  (void)fx_scan_tile<CH, SPANS, false, 2, true, false, false, false, true, false, 2>(sc, bR, reinterpret_cast<const fx_nib*>(tR + 768), nullptr, nullptr,
                                                                                                       a.fpb[p], row, row_ok, true, except, emit);
            }
         }
         if (!bytes_here) {
            if (a.sch[p] == 2u)   // (automata of 9..16 states: the nibble tables, one 64-bit shift per byte -- the same 4 KB of LDS)
               (void)fx_scan_tile<CH, SPANS, RAGGED, 2, false, false, false, true, true>(sc, reinterpret_cast<const fx_nib*>(tR), reinterpret_cast<const fx_nib*>(tA), nullptr,
                                                                                         nullptr, a.fp[p], row, row_ok, true, except, emit);
            else
               (void)fx_scan_tile<CH, SPANS, RAGGED, 0, false, false, false, true, true>(sc, tR, tA, nullptr, nullptr, a.fp[p], row, row_ok, true, except, emit);
         }
         // rows these tables cannot answer: finished inside the launch when the pattern's class-level tables can decode them (queued;
         // the queue is drained after this tile's last pattern when it would not take them), else marked and listed for the pattern's
         // row-level fix-up (one atomic per tile that has any)
         const bool inq_here = bytes_here && a.inq[p] != 0u;
         if (inq_here) {
            const uint64_t qm = __builtin_amdgcn_ballot_w64(except && row_ok);
            if (qm != 0) {
               pend_any |= 1u << p;
               pend[p] = qm;
            }
         }
         const bool listed = except && row_ok && !inq_here;
         const uint64_t em = __builtin_amdgcn_ballot_w64(listed);
         if (em != 0) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&a.ctr[p][1], (uint32_t)__builtin_popcountll(em));
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (listed) {
               a.worklist[p][wbase + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = (uint32_t)row;
               flags[base + row] = FX_NEEDS_GENERAL;
            }
         }
      }
      // this tile's exception rows: into the wave's queue (the tile is finished for every pattern: a drain may overwrite it)
      if (pend_any != 0u) {
         for (uint32_t p = 0; p < a.m; ++p) {
            if (!((pend_any >> p) & 1u)) continue;
            const uint64_t qm = pend[p];
            const uint32_t cnt = (uint32_t)__builtin_popcountll(qm);
            if (xq_n + cnt > 64u) drain();
            if ((qm >> lane) & 1ull) {
               const uint32_t slot = xq_n + (uint32_t)__builtin_popcountll(qm & ((1ull << lane) - 1ull));
               xq[slot] = (uint32_t)row;
               xq[64u + slot] = p;
            }
            xq_n += cnt;
         }
      }
   }
   drain();
   // one plain store per wave and pattern (the value only gates the pattern's pass over deferred tiles)
   if (lane == 0)
      for (uint32_t p = 0; p < a.m; ++p)
         if ((deferred_any >> p) & 1u) a.ctr[p][0] = 1u;
}

// waves per block: 8 when that keeps more waves on a CU than blocks of 4 (the tables are stored once per block)
template <int CH>
hipError_t launch_multi(const uint8_t* rows, int64_t n, const FxMultiArgs& a, uint8_t* flags, int32_t* from, int32_t* to, uint32_t Lr, hipStream_t st) {
   const size_t tab_b = (size_t)a.m * (a.any_bytes ? 8192 : 4096), xb = a.any_bytes ? (size_t)FX_MULTI_MAX * 64 : 0;   // (+ per wave 512 B of queue)
   const size_t t4 = (size_t)4 * 64 * (CH + 1) * 16 + tab_b + (a.any_bytes ? 4 * 512 : 0) + xb, t8 = (size_t)8 * 64 * (CH + 1) * 16 + tab_b + (a.any_bytes ? 8 * 512 : 0) + xb;
   const size_t cap = 160 * 1024;
   const int w4 = t4 <= cap ? 4 * (int)(cap / t4) : 0, w8 = t8 <= cap ? 8 * (int)(cap / t8) : 0;
   if (w4 == 0 && w8 == 0) return hipErrorInvalidValue;   // (the caller bounds m by multi_max_patterns)
   const bool eight = w8 > w4;
   const int wpb = eight ? 8 : 4;
   const size_t lds = eight ? t8 : t4;
   const int64_t n_tiles = (n + 63) >> 6;
   int64_t blocks = (n_tiles + wpb - 1) / wpb;
   const int64_t resident = (int64_t)(cap / lds);
   const int64_t gcap = 256 * resident * 4;
   if (blocks > gcap) blocks = gcap;
   const bool ragged = Lr != 16u * CH, spans = from && to;
#define FX_MULTI_LAUNCH(SP, RG, W)                                                                                                        \
   {                                                                                                                                      \
      const void* fn = reinterpret_cast<const void*>(&fx_search_multi<CH, SP, RG, W>);                                                    \
      if (lds > 64 * 1024) {                                                                                                              \
         hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
         if (e != hipSuccess) return e;                                                                                                   \
      }                                                                                                                                   \
      hipLaunchKernelGGL((fx_search_multi<CH, SP, RG, W>), dim3((unsigned)blocks), dim3(64 * W), lds, st, rows, n, a, flags, from, to, Lr); \
      return hipGetLastError();                                                                                                           \
   }
   if (eight) {
      if (spans) {
         if (ragged) FX_MULTI_LAUNCH(true, true, 8) else FX_MULTI_LAUNCH(true, false, 8)
      } else {
         if (ragged) FX_MULTI_LAUNCH(false, true, 8) else FX_MULTI_LAUNCH(false, false, 8)
      }
   } else {
      if (spans) {
         if (ragged) FX_MULTI_LAUNCH(true, true, 4) else FX_MULTI_LAUNCH(true, false, 4)
      } else {
         if (ragged) FX_MULTI_LAUNCH(false, true, 4) else FX_MULTI_LAUNCH(false, false, 4)
      }
   }
#undef FX_MULTI_LAUNCH
}
// how many patterns one launch can take for rows of this chunk count (tables + at least four tiles in 160 KB of LDS)
static inline int multi_max_patterns(int ch, bool any_bytes = false) {
   const size_t t4 = (size_t)4 * 64 * (ch + 1) * 16 + (any_bytes ? 4 * 512 + FX_MULTI_MAX * 64 : 0);
   int m = (int)((160 * 1024 - t4) / (any_bytes ? 8192 : 4096));
   return m > FX_MULTI_MAX ? FX_MULTI_MAX : (m < 0 ? 0 : m);
}
#define FX_MULTI_SIG (const uint8_t*, int64_t, const FxMultiArgs&, uint8_t*, int32_t*, int32_t*, uint32_t, hipStream_t)
