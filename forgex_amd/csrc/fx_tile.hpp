// Tile kernels of libforgex_amd.so (fx_search_fast, fx_match_fast) and their launchers; see fxamd.hip for the overview and
// DESIGN.md section 4.  This header is compiled into several translation units: fxamd.hip only DECLARES the launcher
// instantiations (extern template), fx_tile_inst.hip defines them for one chunk count per object file so that the few hundred
// kernel variants build in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "compile.hpp"
#include "program.h"
#include "row_engine.hpp"

// Test / experiment hooks (environment variables FXAMD_*), read ONCE per process -- the library's scalar callers make match calls in
// loops, and a dozen getenv per enqueue is host time on their path.  fxamd_reload_env() (C ABI, tests only) reads them again.
struct FxEnv {
   bool no_half, force_general, no_w16, no_byte_dfa, no_a8, no_spec, no_tiny, no_span, no_pack_first, no_adapt, multipass, no_cache, no_multi, multi_no_bytes, multi_inq, multi_serial,
      host_register, multi_w16, no_latch;
   int64_t slice_rows;                                      // rows per enqueue (a multiple of 64)
   int one_grid, one_round_mb, one_blocks, half_rounds, half_sch, span_lens;     // launch-grid experiments (0 = the built-in rule)
};
const FxEnv& fx_env();   // (fxamd.hip)

#define FX_NEEDS_GENERAL 0xFFu   // marker the fast kernel leaves in flags[] for rows with a byte >= 0x80

// =========================================================================================================
// tile staging: 64 rows x (16*CH) bytes, HBM -> LDS, transposed so each lane reads its own row conflict-free
// cell(R, k) = k*64 + (R ^ (k & 7))   [16-byte cells]; chunk k of row R.  Only the low three bits are swizzled: that is all the
// banking needs (a 16-byte cell spans 4 of the 32/64 banks, so 8 resp. 16 consecutive cells are conflict-free), and it leaves the
// row's upper bits additive, so the staging stores of one tile differ only in their immediate offsets (store_tile).
// =========================================================================================================
__device__ __forceinline__ uint32_t tile_cell(uint32_t R, uint32_t k) { return (k << 6) + (R ^ (k & 7u)); }

typedef uint32_t fx_u32x4 __attribute__((ext_vector_type(4)));
#ifndef FX_PREFETCH_DEPTH
#define FX_PREFETCH_DEPTH 1   // tiles of global loads in flight per wave in the first pass
#endif
#ifndef FX_DEFER_PREFETCH
#define FX_DEFER_PREFETCH 0   // (1: match compaction with the lane that owns a queue slot loading the slot's three 8-byte groups from global memory WHEN the row is queued --
                              //  they are in L2: the tile has just been loaded -- into six registers, so that the flush at the wave's end waits for no memory: the flush is 2 us
                              //  of a wave's 17 on config 2, profiles/r06_cfg2_stamps.md.  Measured and NOT kept: 18.2-18.5 -> 19.1-19.3 us per step, four interleaved pairs,
                              //  gpurun call r06_c5 -- the kernel is VALU-bound at four waves per SIMD, and the loads' row-end handling executed once per tile instead of once
                              //  per wave costs more issue time than the one round trip it hides)
#endif
#ifndef FX_DEFER_FWD
#define FX_DEFER_FWD 1   // match compaction in fx_search_one (fx_one.hpp): rows of SPARSE tiles that need the exact start + the forward
                         // pass are queued per wave and finished 64 at a time
#endif
#ifndef FX_DEFER_LONG
#define FX_DEFER_LONG 0  // the same in the segment-walking kernels of this file.  OFF: measured on config 3 (half of the rows match)
                         // 0.65-0.72 ms against 0.48-0.51 ms (profiles/r03_defer_ab.txt) -- the queued rows' from / to become scattered
                         // 4-byte stores (partial lines, written a second time next to the tile's own zeros) and their bytes are read
                         // again from L2 / HBM; the speculative forward pass on the half row still in LDS stays.
#endif
#ifndef FX_FWD_GB
#define FX_FWD_GB 2
#endif
#ifndef FX_HALF4
#define FX_HALF4 1   // the half-row kernel (256-byte rows, spans) tuned for FOUR waves per SIMD: a leaner backward loop (incremental max,
                     // one chunk of LDS prefetch), the forward window's lookups 16 at a time, ONE shared end-of-row cell per wave
                     // (8 KB + 16 B of tile per wave: four blocks per CU) -- 127 VGPRs, no scratch.  Measured against the three-wave
                     // build (156 VGPRs) in one allocation, profiles/r03_half4_ab.txt: kernel 0.500 -> 0.484, 0.491 -> 0.486,
                     // 0.472 -> 0.467 ms; the driver's protocol 0.524 -> 0.511, 0.522 -> 0.498, 0.511 -> 0.486 ms per step
#endif
#ifndef FX_HALF_WAVES
#define FX_HALF_WAVES 3   // waves per SIMD the half-row kernel (256-byte rows, spans) is compiled for
#endif
#ifndef FX_MATCH_LONG_P3
#define FX_MATCH_LONG_P3 1   // `.match.` over rows longer than 256 bytes, 8-state tables: whole segments with three lookup buffers (fx_match_row_pipe3)
#endif
#ifndef FX_ADAPT_CALLS
// The half-row pipeline of 256-byte rows on batches that are mostly UTF-8 (round 4): its first pass loads every tile only to defer it to
// the follow-up -- the rows are read twice.  A first pass that finds more than half of its tiles deferred (sampled: every 256th wave reports
// its tiles and how many of them it deferred, words [3] and [2] of the call's counter group) makes the follow-up set a persistent word
// behind the counter groups; while that word is not zero (it counts down: the next FX_ADAPT_CALLS calls on this scratch set) the first pass
// marks every tile for the follow-up WITHOUT loading it.  Results never depend on the word: the follow-up finishes whatever is marked,
// pure-ASCII tiles included (with the class-level tables); only the cost does, until the count-down ends and a first pass looks again.
#define FX_ADAPT_CALLS 8u
#endif
#ifndef FX_LONG_NT_LAST
#define FX_LONG_NT_LAST 1   // (round 6) a long row's FIRST segment -- the last one its backward pass loads -- takes the `nt` policy whatever the row length: its lines are not
                            // needed again (the lines it shares with the segment to its right were fetched by that segment's pass), so they should not push the
                            // still-to-be-shared lines of other waves out of L2.  Interleaved against a build without it (gpurun call r06_c7): 1024-byte rows
                            // 0.5020 / 0.5013 -> 0.4884 / 0.4896 ms; 400-byte rows unchanged (0.688-0.704 both ways: two segments, and what bounds them is not the
                            // allocation policy of the second one)
#endif
#ifndef FX_LONG_NT_MIN
// Rows longer than 256 bytes, search kernels: the segment loads take the nt policy only from this row length on.  Below it the lines stay
// in L2 for the re-walk and the forward pass (which read the row from global memory once its segments have left the tile) and for the
// neighbouring segment's load when a row length is not a multiple of the 128-byte line: measured in one allocation, nt against the
// default policy (profiles/r04_long_nt_ab.txt): 400-byte rows 1.085 -> 0.890 ms, 1024-byte rows 0.567 -> 0.535 ms, 4096-byte rows
// 0.492 -> 0.522 ms (worse), `.match.` over 1024-byte rows 0.408 -> 0.444 ms (worse: it reads every byte once -- it keeps nt).
#define FX_LONG_NT_MIN 2048u
#endif
#ifndef FX_LOAD_AUX
#define FX_LOAD_AUX 2   // cache policy bits of the tile loads: 2 = nt (rows are read once; measured 2-3 % over the default policy)
#endif
template <int CH>
__device__ __forceinline__ void load_tile(uint4 (&v)[CH], const uint8_t* __restrict__ rows, int64_t row0, int64_t n, uint32_t lane, bool enable = true,
                                          uint32_t row_bytes = 16u * CH) {
   // the tile's bytes are contiguous: 64*CH 16-byte pieces; piece p = q*64+lane -> row p/CH, chunk p%CH.  row0 is wave-uniform:
   // the tile is addressed through a buffer resource whose base is the tile and whose extent is the tile's valid bytes, so
   // each piece is ONE buffer_load_dwordx4 (scalar base, lane offset, immediate piece offset) and the pieces of rows >= n
   // come back as zero from the hardware range check instead of per-piece predication.
   const int64_t rows_left = n - row0;
   // enable == false (wave-uniform): a tile this pass skips -- zero valid bytes, the loads are issued and range-checked away
   // (row_bytes < 16*CH: rows of fewer whole chunks than the instantiation has -- the tile is still one contiguous run of
   //  64*row_bytes bytes, the pieces behind it are range-checked away; store_tile_rt sorts the pieces into rows)
   // (the extent is rounded up to whole dwords: the range check drops a dword that straddles it, and with rows of odd length the
   //  batch may end inside one -- at most 3 bytes behind the caller's last row are read, never used)
   const uint32_t valid = !enable ? 0u : ((rows_left >= 64 ? 64u * row_bytes : (rows_left > 0 ? (uint32_t)rows_left * row_bytes : 0u)) + 3u) & ~3u;
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)row0 * (uint64_t)row_bytes;
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
#pragma unroll
   for (int q = 0; q < CH; ++q) {
      const fx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u + (uint32_t)q * 1024u, 0, FX_LOAD_AUX);
      v[q] = make_uint4(t.x, t.y, t.z, t.w);
   }
}

// Long rows (ANY Lr > 16*CH, CH = 16 or 8): the tile is walked in SEGMENTS of 16*CH bytes (the last one shorter when Lr is not a
// multiple).  Piece q*64+lane of segment `seg` = row (64/CH) q + lane/CH, chunk lane%CH of that segment: CH lanes read 16*CH
// contiguous bytes of one row.  Same buffer resource trick: extent = the tile's valid bytes, the row / segment distance rides in
// the scalar offset (which the range check includes).  CH = 8 with 256-byte rows is the HALF-ROW staging of the headline
// configuration: 8 KB of LDS per wave instead of 16, so that three waves per SIMD fit.
template <int CH, bool TAILFIX = (CH == 16)>
__device__ __forceinline__ void load_tile_seg(uint4 (&v)[CH], const uint8_t* __restrict__ rows, int64_t row0, int64_t n, uint32_t lane, uint32_t Lr,
                                              uint32_t seg_byte, uint32_t k_lo, uint32_t k_hi, bool enable, bool short_seg = false, bool nt = true) {
   // the segment starts at row byte `seg_byte`; tile chunk k holds segment chunk clamp(k, k_lo, k_hi) - k_lo.  A whole segment has
   // (k_lo, k_hi) = (0, CH-1); the short last one has k_hi = its last (possibly partial) chunk and the chunks behind it repeat that
   // one (never walked).  Rows start at any byte: the pieces are unaligned buffer loads.
   static_assert(CH == 16 || CH == 8 || CH == 4, "segment walker: 16, 8 or 4 chunks per segment");
   const int64_t rows_left = n - row0;
   // (extent rounded up to whole dwords, as in load_tile)
   const uint32_t valid = !enable ? 0u : ((rows_left >= 64 ? 64u * Lr : (rows_left > 0 ? (uint32_t)rows_left * Lr : 0u)) + 3u) & ~3u;
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)row0 * (uint64_t)Lr;
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
   uint32_t kc = lane % CH;
   kc = kc < k_lo ? k_lo : (kc > k_hi ? k_hi : kc);
   const uint32_t voff = (lane / CH) * Lr + (kc - k_lo) * 16u;
   const uint32_t s0 = __builtin_amdgcn_readfirstlane(seg_byte);
   // (`nt` is wave-uniform -- a function of the row length: the cache policy is an immediate of the instruction, hence the two loops)
   if (nt) {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
         const fx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s0 + (uint32_t)((64 / CH) * q) * Lr, FX_LOAD_AUX);
         v[q] = make_uint4(t.x, t.y, t.z, t.w);
      }
   } else {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
         const fx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s0 + (uint32_t)((64 / CH) * q) * Lr, 0);
         v[q] = make_uint4(t.x, t.y, t.z, t.w);
      }
   }
   // The short last segment of the tile's LAST row: its final piece may reach past the tile, and the range check works dword by
   // dword on UNALIGNED dwords (rows start at any byte) -- the dword that holds the row's last bytes is dropped whole when it
   // straddles the extent (a 257-byte row's byte 256 came back as 0: a NUL to the automaton).  Those up to three bytes are
   // re-read one by one; nothing behind the tile's own bytes is touched.
   // (CH = 8 is the half-row staging of 256-byte rows: whole segments only -- the code is left out of that kernel, where its mere
   //  presence cost 2-3 %)
   if (TAILFIX && short_seg && enable) {
      const uint32_t tile_bytes = rows_left >= 64 ? 64u * Lr : (uint32_t)rows_left * Lr;
      const uint32_t last_row = (rows_left >= 64 ? 64u : (uint32_t)rows_left) - 1u;
      const uint32_t q_last = last_row / (64u / CH);
      const uint8_t* tb8 = reinterpret_cast<const uint8_t*>(base);
#pragma unroll
      for (int q = 0; q < CH; ++q) {
         if ((uint32_t)q != q_last) continue;   // (wave-uniform)
         const uint32_t ps = voff + s0 + (uint32_t)((64 / CH) * q) * Lr;
         uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
         for (int i = 0; i < 4; ++i) {
            const uint32_t o = ps + 4u * (uint32_t)i;
            if (o < tile_bytes && o + 4u > valid) {
               uint32_t r = 0;
               for (uint32_t j = 0; j < 4u; ++j)
                  if (o + j < tile_bytes) r |= (uint32_t)tb8[o + j] << (8u * j);
               w[i] = r;
            }
         }
         v[q] = make_uint4(w[0], w[1], w[2], w[3]);
      }
   }
}

// symbol 255 (inert at the end of a row in every scheme) behind the row end, for lane r's own row.  Returns the OR of the row's
// own bytes (the loader delivered zeros behind the row end), which the ragged kernels use for their ">= 0x80 anywhere" test.
template <int CH>
__device__ __forceinline__ uint32_t pad_rows(uint4* tile, uint32_t lane, uint32_t Lr) {
   uint32_t na = 0;
#pragma unroll 1
   for (uint32_t k = 0; k < (uint32_t)CH; ++k) {
      uint4 c = tile[tile_cell(lane, k)];
      na |= c.x | c.y | c.z | c.w;
      const uint32_t b = 16u * k;
      if (b + 16u <= Lr) continue;   // wave-uniform
      auto pad = [&](uint32_t wv, uint32_t at) -> uint32_t {   // bytes of the word at row offset `at` that lie behind the row end -> 0xFF
         if (at + 4u <= Lr) return wv;
         if (at >= Lr) return 0xFFFFFFFFu;
         return wv | (0xFFFFFFFFu << (8u * (Lr - at)));
      };
      c.x = pad(c.x, b + 0u);
      c.y = pad(c.y, b + 4u);
      c.z = pad(c.z, b + 8u);
      c.w = pad(c.w, b + 12u);
      tile[tile_cell(lane, k)] = c;
   }
   return na;
}

template <int CH>
__device__ __forceinline__ void store_tile(const uint4 (&v)[CH], uint4* tile, uint32_t lane) {
   if constexpr ((CH & (CH - 1)) == 0) {
      // piece q*64+lane = row q*RPI + r, chunk k (r = lane / CH, k = lane % CH).  With B = max(RPI, 8) the row splits into a
      // multiple of B, which the swizzle leaves alone (an immediate offset), and a rest < B: B/RPI base addresses in all.
      constexpr uint32_t RPI = 64 / CH, B = RPI > 8 ? RPI : 8;
      const uint32_t r = lane / CH, k = lane % CH;
#pragma unroll
      for (int q = 0; q < CH; ++q) {
         const uint32_t hi = ((uint32_t)q * RPI) & ~(B - 1u), lo = ((uint32_t)q * RPI) & (B - 1u);
         tile[(k << 6) + hi + ((lo + r) ^ (k & 7u))] = v[q];
      }
   } else {
#pragma unroll
      for (int q = 0; q < CH; ++q) {
         uint32_t p = q * 64 + lane;
         tile[tile_cell(p / CH, p % CH)] = v[q];
      }
   }
}

// rows of c < CH whole chunks (c = row length / 16, wave-uniform at run time): piece q*64+lane = row p / c, chunk p % c; the chunk
// columns c..CH-1 are never written here (they hold the inert pad symbol, put there once per kernel)
template <int CH>
__device__ __forceinline__ void store_tile_rt(const uint4 (&v)[CH], uint4* tile, uint32_t lane, uint32_t c) {
   const uint32_t inv = 0xFFFFFFFFu / c + 1u;   // p / c == (p * inv) >> 32 for p < 2^16
#pragma unroll
   for (int q = 0; q < CH; ++q) {
      const uint32_t p = q * 64 + lane;
      const uint32_t R = (uint32_t)(((uint64_t)p * inv) >> 32), k = p - R * c;
      if (p < 64u * c) tile[tile_cell(R, k)] = v[q];
   }
}

// rows of ANY other length (Lr % 16 != 0): the tile is still one contiguous run of 64*Lr bytes, loaded with the coalesced
// loader; the pieces go to LDS as a LINEAR image first, then lane r pulls its own row out of it -- bytes r*Lr .. r*Lr+Lr-1, at
// any alignment: aligned dword reads and a per-lane v_alignbyte -- and writes it back in the transposed cell layout every
// other part of the kernel expects (in place: a wave's LDS operations complete in order, all rows are in registers before the
// first cell is overwritten).  Bytes behind the row end come out as zero (pad_rows turns them into the inert symbol).
template <int CH>
__device__ __forceinline__ void store_tile_relayout(uint4 (&v)[CH], uint4* tile, uint32_t lane, uint32_t Lr) {
#pragma unroll
   for (int q = 0; q < CH; ++q) tile[q * 64 + lane] = v[q];
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   const uint32_t base = lane * Lr, sh = base & 3u;   // (16k keeps the byte phase)
   const uint32_t* dw = reinterpret_cast<const uint32_t*>(tb + (base & ~3u));
#pragma unroll
   for (int k = 0; k < CH; ++k) {
      if (16u * k < Lr) {   // wave-uniform
         uint32_t d[5];
#pragma unroll
         for (int i = 0; i < 5; ++i) d[i] = dw[4 * k + i];
         uint32_t w[4];
#pragma unroll
         for (int i = 0; i < 4; ++i) {
            w[i] = fxrow::fx_alignbyte(d[i + 1], d[i], sh);
            const uint32_t at = 16u * k + 4u * i;   // bytes of this word that lie behind the row end -> 0
            if (at >= Lr) w[i] = 0u;
            else if (at + 4u > Lr) w[i] &= ~(0xFFFFFFFFu << (8u * (Lr - at)));
         }
         v[k] = make_uint4(w[0], w[1], w[2], w[3]);
      } else {
         v[k] = make_uint4(0, 0, 0, 0);
      }
   }
#pragma unroll
   for (int k = 0; k < CH; ++k) tile[tile_cell(lane, k)] = v[k];
}

// =========================================================================================================
// Ragged rows, round 4 (fx_search_one): rows of ANY length 2 <= Lr < 16*CH stay LEFT-ALIGNED in their cells and nothing is
// padded with a symbol.  The bytes behind the text hold what follows the text in the wrapped string -- the trailing NUL, then KILL
// symbols -- so every left-to-right walk (forward pass, windows, re-walks, the speculative pass) is the aligned kernels' code; the
// right-to-left pass starts at the row's last byte: chunks behind the text are skipped and the chunk the row ends in is walked over its
// valid bytes only (chain8_back_n).  No inert symbol is involved, so the byte-level tables run on ragged rows as well, and the work
// follows the row length, not the instantiation's chunk count.  Staging: the tile is still one contiguous run of 64*Lr bytes; CH lanes
// read 16*CH bytes of one row (CH = the row's chunk count rounded up to a power of two) -- UNALIGNED 16-byte buffer loads at the row
// stride, straight into the swizzled cells with the aligned kernels' store.  The linear LDS image + per-lane relayout of rounds 1-3 (store_tile_relayout: 4-way bank conflicts at
// Lr = 255, unused chunk columns scanned at Lr = 100 / 132) remains only in the multi-pass kernels of this file.
// =========================================================================================================
struct FxTail {
   uint32_t Lr;    // row length in bytes
   uint32_t kt;    // the chunk position Lr falls in: Lr >> 4 (chunks 0 .. kt-1 are whole text)
   uint32_t nb;    // text bytes in chunk kt: Lr & 15 (0: the row ends on a chunk boundary, chunk kt holds the NUL and KILL symbols only)
   uint32_t nch;   // chunks that hold text: (Lr + 15) >> 4
};
__device__ __forceinline__ FxTail fx_tail_of(const uint32_t Lr) {
   const uint32_t nch = (Lr + 15u) >> 4;
   return FxTail{Lr, Lr >> 4, Lr & 15u, nch};
}
// The same, re-derived HERE from the row length through an opaque scalar move: what a use site computes from it (the tail masks, the
// wave-uniform chunk guards -- each a 64-bit lane mask --, strides) is loop-invariant, and the compiler otherwise hoists all of it out of the
// tile loop: ~100 scalars live across the scan bodies, spilled to lanes of 2-3 VGPRs that the ragged instantiations do not have at three
// waves per SIMD (scratch in the 128-byte ragged kernels).  A few s_ instructions per tile instead.
__device__ __forceinline__ FxTail fx_tail_here(const FxTail& T) {
   uint32_t Lr = T.Lr;
   asm volatile("" : "+s"(Lr));
   return fx_tail_of(Lr);
}
// tile loads: CH (a power of two here: the dispatch rounds a ragged row's chunk count up to one) lanes share a row -- lane = (r0, k) =
// (lane / CH, lane % CH) reads the 16 bytes at row byte 16k of row q * (64 / CH) + r0 in instruction q: voff = r0 * Lr + 16k per lane,
// the rows' distance q * (64 / CH) * Lr in the scalar offset (the range check includes it).  Lanes whose chunk holds no text (k >= nch)
// ask for an address behind the extent and fetch nothing.  The extent is the tile's bytes + 3 (a dword is dropped whole when it
// straddles the extent, and the last row's last dword does unless Lr % 4 == 0) -- except for the batch's last tile, whose extent is exact
// and whose straddling dword is rebuilt from byte loads (fx_patch_tail_piece): nothing behind the caller's last byte is read.
// The batch's LAST tile gets its exact extent -- nothing behind the caller's last byte is read, whatever page it ends on.  The range
// check drops a dword WHOLE when it straddles the extent, and the last row's last dword does unless (Lr + room) % 4 == 0.  Exactly ONE dword of
// the tile that holds text can be affected: dword m = (Lr + room) >> 2 of the tile's last row (every other row's text dwords end at or before
// the extent; the last row's dwords behind m hold no text).  fx_last_dword says which one and rebuilds it from byte loads -- `row0` the tile's
// first row, `rows_in_tile` what a full tile holds; returns false when nothing is to be done (every tile but the batch's last, and that one
// when the extent falls on a dword border of its last row).  The callers patch the tile IN LDS right after their staging store, behind this
// wave-uniform test.  (Round 5 patched the staging REGISTERS in the loaders, piece by piece: 10..40 more live VGPRs on the hot path of every
// ragged instantiation -- 25 of 58 one-launch kernels of 128-byte instantiations spilled at three waves per SIMD)
// (The tiny-row kernels' tiles are lane SPANS of whole rows, possibly a partial last one: the same in units of span bytes.)
struct FxLastDword {
   uint32_t row;    // unit of the tile (a row; the tiny-row kernels: a lane span) the dword lies in
   uint32_t m;      // dword of that unit (byte offset 4 m in it)
   uint32_t word;   // its text bytes (zeros behind the text)
};
// the tile = up to `cap` bytes from `tb` on, of which `left` exist in the batch; its pieces' dwords sit at multiples of 4 from the start of each
// `unit` bytes (a row; a lane span of whole rows).  All arguments wave-uniform.
__device__ __forceinline__ bool fx_last_dword_bytes(FxLastDword& d, const uint8_t* __restrict__ tb, const int64_t left, const uint32_t cap, const uint32_t unit) {
   if (left <= 0) return false;
   const int64_t behind = left - (int64_t)cap;   // bytes of the batch behind this tile
   if (behind >= 3) return false;                // (the extent is the tile's bytes + 3: nothing straddles)
   const uint32_t tile_bytes = behind >= 0 ? cap : (uint32_t)left;
   const uint32_t valid = tile_bytes + (behind > 0 ? (uint32_t)behind : 0u);
   const uint32_t u = (tile_bytes - 1u) / unit;  // the unit the tile's last byte lies in
   const uint32_t rem = valid - u * unit;        // bytes from that unit's first byte to the extent
   if ((rem & 3u) == 0u) return false;
   const uint32_t m = rem >> 2;
   const uint32_t text_end = tile_bytes - u * unit;   // bytes of the unit that belong to the tile
   if (4u * m >= text_end) return false;         // (the straddling dword holds no text)
   const uint8_t* p = tb + (uint64_t)u * unit + 4u * m;
   uint32_t w = 0;
   for (uint32_t j = 0; j < 4u && 4u * m + j < text_end; ++j) w |= (uint32_t)p[j] << (8u * j);
   d.row = u;
   d.m = m;
   d.word = w;
   return true;
}
// ... for tiles of whole rows: `row0` the tile's first row, `rows_in_tile` what a full tile holds
__device__ __forceinline__ bool fx_last_dword(FxLastDword& d, const uint8_t* __restrict__ rows, const int64_t row0, const int64_t n, const uint32_t rows_in_tile,
                                              const uint32_t Lr) {
   if (n - row0 <= 0) return false;
   return fx_last_dword_bytes(d, rows + row0 * (int64_t)Lr, (n - row0) * (int64_t)Lr, rows_in_tile * Lr, Lr);
}
template <int CH>
__device__ __forceinline__ void load_tile_rag(uint4 (&v)[CH], const uint8_t* __restrict__ rows, int64_t row0, int64_t n, uint32_t lane, const FxTail& T_,
                                              bool enable = true) {
   const FxTail T = fx_tail_here(T_);
   static_assert((CH & (CH - 1)) == 0, "ragged rows: the chunk count of the instantiation is a power of two");
   const int64_t rows_left = n - row0;
   // (a tile that is followed by at least 3 more bytes of the batch reads up to 3 of them: the last row's last dword straddles the tile's end
   //  unless Lr % 4 == 0; the batch's last tile -- wave-uniform -- ends its extent at the batch's last byte)
   const uint32_t tile_bytes = rows_left <= 0 ? 0u : (rows_left >= 64 ? 64u : (uint32_t)rows_left) * T.Lr;
   const int64_t room = rows_left > 64 ? (rows_left - 64) * (int64_t)T.Lr : 0;   // bytes of the batch behind this tile
   const bool last_tile = room < 3;
   const uint32_t valid = (!enable || rows_left <= 0) ? 0u : (last_tile ? tile_bytes + (uint32_t)room : tile_bytes + 3u);
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)row0 * (uint64_t)T.Lr;
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
   constexpr uint32_t RPI = 64 / CH;
   const uint32_t r0 = lane / CH, k = lane % CH;
   const uint32_t voff = k < T.nch ? r0 * T.Lr + 16u * k : 0x7FFFFFF0u;
   const uint32_t step = __builtin_amdgcn_readfirstlane(RPI * T.Lr);
#pragma unroll
   for (int q = 0; q < CH; ++q) {
      const fx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (uint32_t)q * step, FX_LOAD_AUX);
      v[q] = make_uint4(t.x, t.y, t.z, t.w);
   }
}
template <int CH>
__device__ __forceinline__ void store_tile_rag(const uint4 (&v)[CH], uint4* tile, uint32_t lane, const FxTail& T_) {
   const FxTail T = fx_tail_here(T_);
   if (lane % CH < T.nch) store_tile<CH>(v, tile, lane);   // (the chunks behind the text keep their KILL symbols)
}
// ... and the batch's last tile: the one text dword the range check dropped (fx_last_dword), written into the row's cell (one lane)
__device__ __forceinline__ void fx_rag_fix_last(uint4* tile, const uint8_t* __restrict__ rows, const int64_t row0, const int64_t n, const uint32_t lane, const FxTail& T_) {
   const FxTail T = fx_tail_here(T_);
   FxLastDword d;
   if (!fx_last_dword(d, rows, row0, n, 64u, T.Lr)) return;   // wave-uniform
   if (lane == 0) reinterpret_cast<uint32_t*>(tile)[(tile_cell(d.row, d.m >> 2) << 2) + (d.m & 3u)] = d.word;
}
// what follows the text in lane r's own cells: the trailing NUL at byte Lr, KILL symbols (0xFE) behind it.  Chunks behind chunk kt are
// written ONCE per kernel (the loader never touches them); chunk kt -- the text's last bytes, rewritten with every tile -- is patched here.
__device__ __forceinline__ uint32_t fx_tail_word(const uint32_t w, const uint32_t at, const uint32_t nb) {   // dword at chunk byte `at` of chunk kt
   if (at + 4u <= nb) return w;
   if (at >= nb) return at == nb ? 0xFEFEFE00u : 0xFEFEFEFEu;
   const uint32_t sh = 8u * (nb - at);   // 8, 16 or 24: that many low bits are text
   return (w & ~(0xFFFFFFFFu << sh)) | (0xFEFEFE00u << sh);
}
__device__ __forceinline__ void fx_tail_patch(uint4* tile, uint32_t lane, const FxTail& T_) {
   const FxTail T = fx_tail_here(T_);
   if (T.nb == 0u) return;
   uint4 c = tile[tile_cell(lane, T.kt)];
   c.x = fx_tail_word(c.x, 0u, T.nb);
   c.y = fx_tail_word(c.y, 4u, T.nb);
   c.z = fx_tail_word(c.z, 8u, T.nb);
   c.w = fx_tail_word(c.w, 12u, T.nb);
   tile[tile_cell(lane, T.kt)] = c;
}
template <int CH>
__device__ __forceinline__ void fx_tail_init(uint4* tile, uint32_t lane, const FxTail& T) {   // once per kernel
   for (uint32_t k = T.nch; k < (uint32_t)CH; ++k)
      tile[tile_cell(lane, k)] = (k == T.kt) ? make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu) : make_uint4(0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
}
// the text bytes of chunk kt OR-ed (the ">= 0x80 anywhere" tests look at text only)
__device__ __forceinline__ uint32_t fx_tail_or(const uint4 c, const uint32_t nb) {
   auto m = [&](uint32_t w, uint32_t at) -> uint32_t { return at + 4u <= nb ? w : (at >= nb ? 0u : (w & ~(0xFFFFFFFFu << (8u * (nb - at))))); };
   return m(c.x, 0u) | m(c.y, 4u) | m(c.z, 8u) | m(c.w, 12u);
}
// one row gathered from global memory into lane r's cells (exception queues): whole chunks are plain unaligned 16-byte loads; the chunk
// the row ends in is read as the row's LAST 16 bytes and shifted down (nothing behind the row is read), rows shorter than 16 bytes byte by byte
struct __attribute__((packed, aligned(1))) fx_u4_unaligned {
   uint32_t x, y, z, w;
};
__device__ __forceinline__ uint4 fx_shr128_bytes(const uint4 v, const uint32_t nbytes) {   // nbytes wave-uniform, 1..15
   const uint32_t d = nbytes >> 2, b = nbytes & 3u;
   uint32_t a0, a1, a2, a3;
   if (d == 0u) { a0 = v.x; a1 = v.y; a2 = v.z; a3 = v.w; }
   else if (d == 1u) { a0 = v.y; a1 = v.z; a2 = v.w; a3 = 0u; }
   else if (d == 2u) { a0 = v.z; a1 = v.w; a2 = 0u; a3 = 0u; }
   else { a0 = v.w; a1 = 0u; a2 = 0u; a3 = 0u; }
   if (b == 0u) return make_uint4(a0, a1, a2, a3);
   return make_uint4(fxrow::fx_alignbyte(a1, a0, b), fxrow::fx_alignbyte(a2, a1, b), fxrow::fx_alignbyte(a3, a2, b), fxrow::fx_alignbyte(0u, a3, b));
}
template <int CH>
__device__ __forceinline__ void gather_row_rag(uint4* tile, uint32_t lane, const uint8_t* __restrict__ rp, bool on, const FxTail& T_) {
   const FxTail T = fx_tail_here(T_);
#pragma unroll 1
   for (uint32_t k = 0; k < T.kt; ++k) {
      uint4 c = make_uint4(0, 0, 0, 0);
      if (on) {
         const fx_u4_unaligned u = *reinterpret_cast<const fx_u4_unaligned*>(rp + 16u * k);
         c = make_uint4(u.x, u.y, u.z, u.w);
      }
      tile[tile_cell(lane, k)] = c;
   }
   if (T.nb != 0u) {
      uint4 c = make_uint4(0, 0, 0, 0);
      if (T.Lr >= 16u) {
         if (on) {
            const fx_u4_unaligned u = *reinterpret_cast<const fx_u4_unaligned*>(rp + T.Lr - 16u);
            c = make_uint4(u.x, u.y, u.z, u.w);
         }
         c = fx_shr128_bytes(c, 16u - T.nb);
      } else if (on) {
         uint64_t lo = 0, hi = 0;   // (no run-time indexed array: that would live in scratch)
         for (uint32_t j = 0; j < T.Lr; ++j) {
            const uint64_t b = rp[j];
            if (j < 8u) lo |= b << (8u * j);
            else hi |= b << (8u * (j - 8u));
         }
         c = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
      }
      tile[tile_cell(lane, T.kt)] = c;
   }
}

// =========================================================================================================
// fast search kernel
// =========================================================================================================
// every value is a state id replicated into all four bytes (id * 0x01010101): v_perm_b32 then advances four identical
// copies of the automaton and `state >= hit_min` can compare whole registers without masking
struct FastParams {
   uint32_t R_start, A_init, hit_min, acc_min;
   uint32_t inv;            // BYTES modes: the INVALID state (structurally invalid UTF-8): the row is left to the decode path;
                            // inv_on: the overlap state of R (FXP_F_OVERLAP_SINK): the row is left to the general engine
   uint32_t inv_on;
   uint32_t defer_tiles;    // first pass: tiles holding a byte >= 0x80 are deferred whole (a later pass handles them)
   uint32_t gate_word;      // marked-tile passes: which of the call's two words says whether there is anything to do
   uint32_t lit_len;        // > 0: literal INDEX search (FXP_F_RAW_BYTES): no forward pass, the match is lit_len bytes from the start
   uint32_t spec;           // byte-level tables of fx_search_one: bit 0 = FXP_F_SPEC_FWD (the speculative forward pass from the row's first character is
                            // sound), bit 1 = FXP_F_NEEDS_NONASCII (a row without a byte >= 0x80 holds no match)
   uint32_t out_mode;       // first passes that write PACKED results themselves (round 5: the half-row kernel of 256-byte rows, the span kernel):
                            // 0 = flags u8[n], from / to int32[n]; 1 / 2 / 4 = flag bit words + spans of that many bytes, and "this 64-row tile is left
                            // to the follow-up" goes to a byte per tile (`marks`) instead of the rows' flag bytes
   uint32_t latch;          // 1: the LATCHED format of R (FXP_F_R_LATCH, program.h; round 6) -- the launcher picks the kernel's LATCH instantiation, the R
                            // table is off_fastRL, hit_min = 4 (x 0x01010101: "latched"), and hit_base = the first hit state among the base states 0..3
   uint32_t hit_base;
};
#define FX_LATCH_MASK 0x03030303u   // state & mask = the base state, latch cleared

// 8 independent table lookups for 8 bytes.  Three table schemes share the kernels (template parameter SCH):
//   0 v_perm       (<= 8 states): F = uint2 = the 8 next-state bytes of the symbol; step = ONE v_perm_b32.
//   2 wide         (<= 16 states): F = fx_nib = 16 next-state nibbles; step = a 64-bit shift by 4*state and a mask (see fxstep below).
//   1 chain        (larger automata): F = 2 * column of the symbol's class (uint16 map, state-independent, pipelined the same
//                  way); the state is the byte offset of its row in a class-indexed uint16 table held in LDS and the step is a
//                  dependent ds_read_u16 of T[state + F] (the destination's row offset).
template <class F, class TabT>
__device__ __forceinline__ void lookup8(F* __restrict__ f, uint32_t lo, uint32_t hi, const TabT* __restrict__ tab) {
#pragma unroll
   for (int i = 0; i < 8; ++i) f[i] = tab[((i < 4 ? lo : hi) >> ((i & 3) * 8)) & 0xFFu];
}
__device__ __forceinline__ uint32_t fxstep(uint2 f, uint32_t st, const uint8_t*) { return __builtin_amdgcn_perm(f.y, f.x, st); }
// 16-state scheme ("wide"): F = 16 next-state NIBBLES (8 bytes, one ds_read_b64); the state is a plain id 0..15 and one step is
// (entry >> 4*state) & 15 -- v_lshlrev_b32, v_lshrrev_b64, v_and_b32.  (A 16-byte-per-symbol v_perm format -- two v_perm_b32 and
// an AND -- moved twice the LDS bytes per input byte and ran at 0.55x the steps per second: tools/ubench/step_rate.hip.)
struct __attribute__((aligned(8))) fx_nib {
   uint32_t x, y;
};
__device__ __forceinline__ uint32_t fxstep(fx_nib f, uint32_t st, const uint8_t*) {
   return (uint32_t)(((((uint64_t)f.y) << 32) | f.x) >> (st << 2)) & 15u;
}
__device__ __forceinline__ uint32_t fxstep(uint32_t f, uint32_t st, const uint8_t* T) {
   return *reinterpret_cast<const uint16_t*>(T + st + f);
}
// table scheme SCH: 0 = v_perm (<= 8 states), 1 = LDS chain, 2 = wide (<= 16 states, nibble tables: one 64-bit shift per byte)
template <int SCH>
struct FxF {
   using type = uint2;
};
template <>
struct FxF<1> {
   using type = uint32_t;
};
template <>
struct FxF<2> {
   using type = fx_nib;
};

// Right-to-left state chain over 8 bytes.  All four bytes of `state` carry the same state id (v_perm_b32 advances four
// identical copies), so whole registers compare like ids and no masking is needed.  Hit states have the LARGEST ids, so
// the group's "any hit" is max(states) >= hit_min: one v_max3_u32 per two bytes instead of a compare+select per byte.
// (LEAN: the running maximum instead of eight kept states -- four more v_max per group, five fewer live registers)
template <class F, bool LEAN = false, bool LATCH = false>
__device__ __forceinline__ uint32_t chain8_back(const F (&f)[8], uint32_t& state, const uint8_t* T) {
   if constexpr (LATCH) {   // (latched format of R: the group's last state says whether any of its states was a hit -- no maximum; the caller clears the latch)
#pragma unroll
      for (int i = 7; i >= 0; --i) state = fxstep(f[i], state, T);
      return state;
   }
   if constexpr (LEAN) {
      uint32_t m = 0;
#pragma unroll
      for (int i = 7; i >= 1; i -= 2) {
         const uint32_t s1 = fxstep(f[i], state, T);
         state = fxstep(f[i - 1], s1, T);
         m = max(max(m, s1), state);
      }
      return m;
   }
   uint32_t st[8];
#pragma unroll
   for (int i = 7; i >= 0; --i) {
      state = fxstep(f[i], state, T);
      st[i] = state;
   }
   uint32_t m0 = max(max(st[0], st[1]), st[2]);
   uint32_t m1 = max(max(st[3], st[4]), st[5]);
   uint32_t m2 = max(st[6], st[7]);
   return max(max(m0, m1), m2);
}

// the same over only the first nv (1..7, wave-uniform) bytes of the group -- the end of a long row's last segment
template <class F, bool LATCH = false>
__device__ __forceinline__ uint32_t chain8_back_n(const F (&f)[8], uint32_t& state, const uint8_t* T, uint32_t nv) {
   uint32_t mx = 0;
#pragma unroll
   for (int i = 7; i >= 0; --i)
      if ((uint32_t)i < nv) {
         state = fxstep(f[i], state, T);
         if (!LATCH) mx = max(mx, state);
      }
   return LATCH ? state : mx;
}

// ---- on-device UTF-8 decode for the fast kernel (FXP_F_FAST_UTF8) -------------------------------------------------------
// A tile that holds any byte >= 0x80 is rewritten IN REGISTERS, before it is stored to LDS, into fast-path symbol ids:
// ASCII bytes stay; the first byte of a structurally valid multi-byte character becomes 128 + class(code point); its
// continuation bytes become 255 (SKIP); every byte of an invalid sequence becomes 128 + class(U+FFFF) -- the reference's
// strict stepping (utf8_m.f90:44-140,168-246) and arithmetic decode (:338-430), decided per position from a +-3 byte
// window (a lead byte is always a character start; a continuation byte is inside a character iff the nearest
// non-continuation byte within 3 to its left is a lead whose whole sequence is continuation bytes).
// Symbol stream of one row for the forward pass, starting at ANY byte index j: text bytes, then 0x00 for the trailing NUL
// at index L, then 0xFE (the symbol id whose table row is all-dead) -- so end-of-row needs no per-byte test.
#ifndef FX_LIVE_PRED
#define FX_LIVE_PRED 1   // (0: the round-4 behaviour, for A/B builds)
#endif
template <bool RAGGED, bool LONG = false>
__device__ __forceinline__ void group_words(uint32_t& lo, uint32_t& hi, const uint8_t* tb, uint32_t lane, uint32_t p, uint32_t L, const uint8_t* eor = nullptr,
                                            const bool live = true) {
   if (LONG) {
      // long rows: `tb` is the row itself in global memory (the LDS tile only ever holds one 256-byte segment); any L >= 8
      // (live: a lane whose walk is over reads nothing -- round 5: 86 % of config 3's tiles of 1024-byte rows hold a match longer than the
      //  32-symbol window, and the 60-odd dead lanes riding along fetched a line each per round: 1.16 x the algorithmic bytes)
      if (FX_LIVE_PRED != 0 && !live) {
         lo = 0xFEFEFEFEu;
         hi = 0xFEFEFEFEu;
      } else if (p + 8u <= L) {
         const uint2 r = *reinterpret_cast<const uint2*>(tb + p);   // (rows start at any byte: an unaligned 8-byte load)
         lo = r.x;
         hi = r.y;
      } else if (p >= L) {
         lo = p == L ? 0xFEFEFE00u : 0xFEFEFEFEu;
         hi = 0xFEFEFEFEu;
      } else {
         // the row ends inside this group: its last 8 bytes, shifted down to position p; then the NUL, then KILL symbols
         const uint2 r = *reinterpret_cast<const uint2*>(tb + L - 8u);
         const uint32_t nb = 8u * (L - p);   // bits of text: 8 .. 56
         const uint64_t v = ((((uint64_t)r.y << 32) | r.x) >> (64u - nb)) | (0xFEFEFEFEFEFEFE00ull << nb);
         lo = (uint32_t)v;
         hi = (uint32_t)(v >> 32);
      }
      return;
   }
   if (!RAGGED) {
      // whole chunks: index L.. lives in the row's extra chunk column (NUL, then KILL symbols); anything further reads its KILL half
      const uint32_t pc = p < L + 8u ? p : L + 8u;   // p and L are multiples of 8
      // (eor: ONE end-of-row cell shared by the wave's rows instead of a chunk column of them)
      const uint8_t* src = (eor != nullptr && pc >= L) ? eor + (pc & 8u) : tb + (tile_cell(lane, pc >> 4) << 4) + (pc & 8u);
      const uint2 r = *reinterpret_cast<const uint2*>(src);
      lo = r.x;
      hi = r.y;
      return;
   }
   const uint32_t pc = p < L ? p : 0u;   // p is a multiple of 8, L is any length
   const uint2 r = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, pc >> 4) << 4) + (pc & 15u));
   // 0x00 at index L (the trailing NUL), 0xFE (the symbol id whose table row is all-dead) behind it -- at byte granularity
   auto word = [&](uint32_t wv, uint32_t at) -> uint32_t {
      if (at + 4u <= L) return wv;
      if (at >= L) return at == L ? 0xFEFEFE00u : 0xFEFEFEFEu;
      const uint32_t sh = 8u * (L - at);   // 8, 16 or 24: that many low bits are text
      return (wv & ~(0xFFFFFFFFu << sh)) | (0xFEFEFE00u << sh);
   };
   lo = word(r.x, p);
   hi = word(r.y, p + 4u);
}
template <bool RAGGED, bool LONG = false>
__device__ __forceinline__ void fetch32(uint32_t (&o)[8], const uint8_t* tb, uint32_t lane, uint32_t j, uint32_t L, const uint8_t* eor = nullptr,
                                        const bool live = true) {
   const uint32_t base = j & ~7u, sh = j & 7u;
   uint32_t d[10];
#pragma unroll
   for (int g = 0; g < 5; ++g) group_words<RAGGED, LONG>(d[2 * g], d[2 * g + 1], tb, lane, base + 8u * g, L, eor, live);
   const uint32_t up = 0u - ((sh >> 2) & 1u);   // all ones when the stream starts in the odd dword (bit-select, not indexing)
   uint32_t e[9];
#pragma unroll
   for (int k = 0; k < 9; ++k) e[k] = (up & d[k + 1]) | (~up & d[k]);
#pragma unroll
   for (int k = 0; k < 8; ++k) o[k] = __builtin_amdgcn_alignbyte(e[k + 1], e[k], sh & 3u);
}
// the same for NG 8-symbol groups (a shorter first window for short rows)
template <bool RAGGED, int NG>
__device__ __forceinline__ void fetch_groups(uint32_t (&o)[2 * NG], const uint8_t* tb, uint32_t lane, uint32_t j, uint32_t L) {
   const uint32_t base = j & ~7u, sh = j & 7u;
   uint32_t d[2 * NG + 2];
#pragma unroll
   for (int g = 0; g < NG + 1; ++g) group_words<RAGGED, false>(d[2 * g], d[2 * g + 1], tb, lane, base + 8u * g, L);
   const uint32_t up = 0u - ((sh >> 2) & 1u);   // all ones when the stream starts in the odd dword (bit-select, not indexing)
   uint32_t e[2 * NG + 1];
#pragma unroll
   for (int k = 0; k < 2 * NG + 1; ++k) e[k] = (up & d[k + 1]) | (~up & d[k]);
#pragma unroll
   for (int k = 0; k < 2 * NG; ++k) o[k] = __builtin_amdgcn_alignbyte(e[k + 1], e[k], sh & 3u);
}
template <bool RAGGED, bool LONG = false>
__device__ __forceinline__ void fetch8(uint32_t (&o)[2], const uint8_t* tb, uint32_t lane, uint32_t j, uint32_t L) {
   const uint32_t base = j & ~7u, sh = j & 7u;
   uint32_t d[4];
   group_words<RAGGED, LONG>(d[0], d[1], tb, lane, base, L);
   group_words<RAGGED, LONG>(d[2], d[3], tb, lane, base + 8u, L);
   const uint32_t up = 0u - ((sh >> 2) & 1u);
   uint32_t e[3];
#pragma unroll
   for (int k = 0; k < 3; ++k) e[k] = (up & d[k + 1]) | (~up & d[k]);
   o[0] = __builtin_amdgcn_alignbyte(e[1], e[0], sh & 3u);
   o[1] = __builtin_amdgcn_alignbyte(e[2], e[1], sh & 3u);
}

// Chain tables -> LDS: the 256-entry symbol map, then T_R (nr entries), then T_A (na entries), all 16-bit.  Four entries per thread are READ before
// any is stored (round 5: a load -> store loop made every 256 entries a round trip to L2 of their own at the start of every block; measured on
// the 17-state pattern over 256- and 128-byte rows: 0.717 -> 0.715 ms, 0.492 -> 0.495 ms -- nothing either way, gpurun call r05_c29: the
// other blocks of a CU cover a block's start-up).
__device__ __forceinline__ void fx_stage_chain(uint16_t* __restrict__ dst, const uint16_t* __restrict__ g, const uint16_t* __restrict__ gr,
                                               const uint16_t* __restrict__ ga, const uint32_t nr, const uint32_t na) {
   const uint32_t total = 256u + nr + na;
   for (uint32_t base = threadIdx.x; base < total; base += 1024u) {
      uint16_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const uint32_t i = base + 256u * (uint32_t)u;
         v[u] = i < 256u ? g[i] : (i < 256u + nr ? gr[i - 256u] : (i < total ? ga[i - 256u - nr] : (uint16_t)0));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const uint32_t i = base + 256u * (uint32_t)u;
         if (i < total) dst[i] = v[u];
      }
   }
}

// ---- match compaction: one queued row per lane finished from GLOBAL memory ----------------------------------------------------------
// Lane `on`: row bytes at rp (length L >= 8, any alignment), leftmost hit in 8-byte group g entered in reverse state e.  Re-walks the
// group for the exact start, then walks forward from it (first window of 8*NW symbols, lookups issued 8*GB at a time, then 8 symbols per
// round while any lane is alive).  Returns the wrapped start index s (>= 2) and max_match mm (0 = none; lit_len != 0: s + lit_len, no
// walk).  The same arithmetic as the in-tile path (fx_scan_tile / fx_search_fast), with the row read through group_words<.., LONG>.
// (PRE: the window's NW + 1 groups were copied into LDS when the row was queued -- `pre`[64 k + lane] = group k of this lane's row; only a
//  match longer than the window still reads the row.  `prr` (round 6): the same three groups in this lane's REGISTERS, loaded from global memory
//  when the row took its slot -- fx_scan_tile, FX_DEFER_PREFETCH: the flush then waits for nothing.)
template <int S_, int NW, int GB, int S_A = S_, bool PRE = false, class TabT, class TabTA>
__device__ __forceinline__ void fx_finish_from_global(const TabT* __restrict__ tabR, const TabTA* __restrict__ tabA, const uint8_t* TRp, const uint8_t* TAp,
                                                      const FastParams& P, const uint8_t* rp, const uint32_t L, const uint32_t lane, const bool on,
                                                      const uint32_t g, const uint32_t e, uint32_t& s_out, uint32_t& mm_out, const uint2* pre = nullptr,
                                                      const uint32_t* prr = nullptr) {
   using F = typename FxF<S_>::type;
   using FA = typename FxF<S_A>::type;
   static_assert(NW % GB == 0, "window groups: a multiple of the lookup batch");
   // Round 4: the forward window's loads are issued BEFORE the re-walk.  The window starts in the hit group itself (j = 8 g + loc, so its
   // aligned base is 8 g whatever `loc` turns out to be), which the compiler cannot know: the loads used to wait for the re-walk's result --
   // a second round trip to L2 / HBM at the end of a wave that has nothing to overlap it with (config 2: the flush is the kernel's tail).
   uint32_t d[2 * NW + 2];
   if constexpr (PRE) {
      static_assert(NW == 2, "the queue holds three groups per row");
#pragma unroll
      for (int q = 0; q < NW + 1; ++q) {
         const uint2 r = pre[64 * q + (int)lane];   // (slots behind the queue's end hold an earlier flush's groups: their lanes are off)
         d[2 * q] = r.x;
         d[2 * q + 1] = r.y;
      }
   } else if (NW == 2 && prr != nullptr) {
#pragma unroll
      for (int k = 0; k < 2 * NW + 2; ++k) d[k] = prr[k];
   } else {
      const uint32_t base0 = on ? g * 8u : 0u;
#pragma unroll
      for (int q = 0; q < NW + 1; ++q) group_words<false, true>(d[2 * q], d[2 * q + 1], rp, lane, base0 + 8u * q, L, nullptr, on);
   }
   // the hit group = the window's first group; the row may end inside it: only its text bytes are walked (the loader put the NUL / KILL
   // symbols behind them)
   const uint2 rw = on ? make_uint2(d[0], d[1]) : make_uint2(0, 0);
   const uint32_t nv = (on && g * 8u + 8u > L) ? L - g * 8u : 8u;
   uint32_t s;
   {
      F f[8];
      lookup8(f, rw.x, rw.y, tabR);
      uint32_t st = e, loc = 0;
#pragma unroll
      for (int i = 7; i >= 0; --i) {
         const uint32_t nx = fxstep(f[i], st, TRp);
         const bool ok = (uint32_t)i < nv;   // (per lane)
         st = ok ? nx : st;
         loc = ok && nx >= P.hit_min ? (uint32_t)i : loc;
      }
      s = g * 8u + 2u + loc;
   }
   s_out = s;
   uint32_t mm = P.lit_len != 0 ? s + P.lit_len : 0u;
   uint32_t cur = (on && P.lit_len == 0) ? P.A_init : 0u;
   uint32_t j = s - 2u;
   if (__builtin_amdgcn_ballot_w64(cur != 0) != 0) {
      uint32_t o[2 * NW];
      {
         const uint32_t sh = j & 7u;   // (the aligned base of j is 8 g: `d` holds the window's groups already)
         const uint32_t up = 0u - ((sh >> 2) & 1u);
         uint32_t ee[2 * NW + 1];
#pragma unroll
         for (int k = 0; k < 2 * NW + 1; ++k) ee[k] = (up & d[k + 1]) | (~up & d[k]);
#pragma unroll
         for (int k = 0; k < 2 * NW; ++k) o[k] = __builtin_amdgcn_alignbyte(ee[k + 1], ee[k], sh & 3u);
      }
      uint32_t gl = 0xFFFFFFFFu, el = 0, blo = 0, bhi = 0;
#pragma unroll
      for (int gb = 0; gb < NW; gb += GB) {
         FA f[8 * GB];
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
            const bool hit = mx >= P.acc_min;
            gl = hit ? (uint32_t)(gb + q) : gl;
            el = hit ? entry : el;
            blo = hit ? o[2 * (gb + q)] : blo;
            bhi = hit ? o[2 * (gb + q) + 1] : bhi;
         }
      }
      {
         FA fr8[8];
         lookup8(fr8, blo, bhi, tabA);
         uint32_t st = el, loc = 0;
#pragma unroll
         for (int i = 0; i < 8; ++i) {
            st = fxstep(fr8[i], st, TAp);
            loc = st >= P.acc_min ? (uint32_t)i : loc;
         }
         mm = gl != 0xFFFFFFFFu ? j + 8u * gl + loc + 3u : mm;
      }
      j += 8u * NW;
      if (__builtin_amdgcn_ballot_w64(cur != 0) != 0) {   // matches longer than the window: 8 symbols per round, the next group read one round ahead
         const uint32_t sh = j & 7u, up = 0u - ((sh >> 2) & 1u);
         uint32_t gbp = j & ~7u;
         uint32_t t0[2], t1[2];
         group_words<false, true>(t0[0], t0[1], rp, lane, gbp, L, nullptr, cur != 0);
         group_words<false, true>(t1[0], t1[1], rp, lane, gbp + 8u, L, nullptr, cur != 0);
         do {
            uint32_t t2[2];
            group_words<false, true>(t2[0], t2[1], rp, lane, gbp + 16u, L, nullptr, cur != 0);
            const uint32_t e0 = (up & t0[1]) | (~up & t0[0]), e1 = (up & t1[0]) | (~up & t0[1]), e2 = (up & t1[1]) | (~up & t1[0]);
            const uint32_t o0 = __builtin_amdgcn_alignbyte(e1, e0, sh & 3u), o1 = __builtin_amdgcn_alignbyte(e2, e1, sh & 3u);
            FA f8[8];
            lookup8(f8, o0, o1, tabA);
            uint32_t loc = 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
               cur = fxstep(f8[i], cur, TAp);
               loc = cur >= P.acc_min ? (uint32_t)i : loc;
            }
            mm = loc != 8u ? j + loc + 3u : mm;
            j += 8u;
            gbp += 8u;
            t0[0] = t1[0]; t0[1] = t1[1];
            t1[0] = t2[0]; t1[1] = t2[1];
         } while (__builtin_amdgcn_ballot_w64(cur != 0) != 0);
      }
   }
   mm_out = mm;
}

// ---- optional phase stamps (debug builds only: make stamp) --------------------------------------------------------------
// -DFX_STAMP (`make stamp-fast STAMP_OBJ=<chunks>_1`, tools/stamp_one.py --fast): every wave accumulates s_memtime deltas per phase of fx_search_fast and ADDS them to
// its own row of fx_stamp_buf at its end (plain stores: per-wave atomics to one set of words measured themselves -- fx_one.hpp, ONE_STAMP).
#ifdef FX_STAMP
#define FX_STAMP_MAX_WAVES 65536
#define FX_STAMP_SLOTS 12
static __device__ unsigned long long fx_stamp_buf[FX_STAMP_MAX_WAVES * FX_STAMP_SLOTS];   // per wave: 0..7 phase ticks, 8 tiles, 9 -, 10 lifetime, 11 launches seen
#define STAMP_DECL const unsigned long long _st_t0 = __builtin_amdgcn_s_memtime(); unsigned long long _st_t = _st_t0, _st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i)                                                     \
   do {                                                              \
      const unsigned long long _n = __builtin_amdgcn_s_memtime();    \
      _st_acc[i] += _n - _st_t;                                      \
      _st_t = _n;                                                    \
   } while (0)
#define STAMP_FLUSH                                                                                                   \
   do {                                                                                                               \
      const unsigned long long _life = __builtin_amdgcn_s_memtime() - _st_t0;                                         \
      const int64_t _wg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);                                               \
      if (lane == 0 && _wg < FX_STAMP_MAX_WAVES) {                                                                    \
         unsigned long long* _row = fx_stamp_buf + _wg * FX_STAMP_SLOTS;                                              \
         for (int _i = 0; _i < 8; ++_i) _row[_i] += _st_acc[_i];                                                      \
         _row[8] += n_seen;                                                                                           \
         _row[10] += _life;                                                                                           \
         _row[11] += 1ull;                                                                                            \
      }                                                                                                               \
   } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// the tile of 64 rows is ONE contiguous run of 64*Lr bytes whatever the row length: fully coalesced 16-byte pieces
#define LOAD_TILE(st, r0)                                              \
   do {                                                                \
      if (RAGGED) load_tile<CH>(st, rows, (r0), n, lane, true, Lr);    \
      else load_tile<CH>(st, rows, (r0), n, lane);                     \
   } while (0)
// staged pieces -> LDS tile (RAGGED with whole chunks: run-time piece-to-row map)
#define STORE_TILE(st)                                                                      \
   do {                                                                                     \
      if (RAGGED && (Lr & 15u) == 0u) store_tile_rt<CH>(st, tile, lane, Lr >> 4);           \
      else if (RAGGED) store_tile_relayout<CH>(st, tile, lane, Lr);                         \
      else store_tile<CH>(st, tile, lane);                                                  \
   } while (0)
// prefetch of a tile that may lie behind the last one: the aligned loader needs no guard (zero valid bytes -> every piece is
// range-checked away), and an unguarded load keeps the staging registers free of control-flow merges
#define PREFETCH_TILE(st, tn, en)                                      \
   do {                                                                \
      if (!RAGGED) load_tile<CH>(st, rows, (tn) << 6, n, lane, (en));  \
      else load_tile<CH>(st, rows, (tn) << 6, n, lane, (en), Lr);      \
   } while (0)

// segment sg of a long row: bytes [SEGB sg, SEGB sg + SEGB) of every row (SEGB = 16*CH); the LAST segment is shorter when Lr % SEGB != 0 and sits
// left-aligned in the tile: its chunks behind the row end repeat the last one (loaded, never walked)
#define PREFETCH_SEG(st, tn, sg, en) \
   load_tile_seg<CH, (CH == 16 || NOHALF)>(st, rows, (tn) << 6, n, lane, Lr, (sg) * SEGB, 0u, (((sg) + 1u) * SEGB > Lr) ? (((Lr % SEGB) + 15u) >> 4) - 1u : (uint32_t)CH - 1u, (en), (((sg) + 1u) * SEGB > Lr), CH == 8 || (CH == 16 && (Lr >= FX_LONG_NT_MIN || (FX_LONG_NT_LAST != 0 && (sg) == 0u))))
#define PREFETCH_SEG_FWD(st, tn, sg, en) \
   load_tile_seg<CH, (CH == 16 || NOHALF)>(st, rows, (tn) << 6, n, lane, Lr, (sg) * SEGB, 0u, (((sg) + 1u) * SEGB > Lr) ? (((Lr % SEGB) + 15u) >> 4) - 1u : (uint32_t)CH - 1u, (en), (((sg) + 1u) * SEGB > Lr), CH != 4)

// FIXUP = false: first pass over the caller's rows.  Tiles holding a byte >= 0x80 are not scanned here: with
//                 FXP_F_FAST_UTF8 the whole tile is marked FX_NEEDS_GENERAL (flags) and left to the second pass, otherwise
//                 the offending rows are appended one by one to the worklist of the general engine's fix-up.
// FIXUP = true:  second pass: only marked tiles are loaded, decoded from UTF-8 to symbol ids in LDS, then scanned.
// MODE 0: first pass (class-level tables), as above.            MODE 1: the decode second pass (FIXUP), marked tiles only.
// MODE 2: byte-level tables (FXP_F_BYTE_DFA) over ALL tiles: raw bytes are the symbols, nothing is decoded or deferred; rows
//         whose backward pass ends in the INVALID state go to the worklist (decode pass or fx_fixup_list).
// MODE 3: the same over the tiles a MODE 0 pass deferred.
// MODE 4: the decode pass over a WORKLIST of row indices (the exception rows a BYTES pass appended): each lane gathers its own
//         row into its LDS cells, results are scattered back to the rows' own slots.
// n_deferred points at this call's two words: [0] "a first pass deferred tiles", [1] number of exception rows in `worklist`.
// LONG: rows longer than 256 bytes (any length up to 64 KiB), CH = 16: the backward pass walks the row segment by segment through the
// same LDS tile, the short forward pass reads its bytes straight from global memory.  First-pass and BYTES modes only.
// (no end-of-row chunk column in LDS for the segment-walking kernels once their forward pass reads global memory only)
// Round 5 (FX_LONG_CAP): the segment walkers with spans keep TWO more chunk columns per row -- the first 32 bytes of the segment to the
// RIGHT of the one in the tile (or what follows the row's end: NUL, KILL symbols) -- so that the 40 bytes from a hit group on can be
// captured into registers while its segment is in LDS: the exact start and the forward window then read registers, not global memory.
#ifndef FX_LONG_CAP
#define FX_LONG_CAP 1
#endif
template <int CH, bool SPANS, bool LONG, bool NOHALF = false>
constexpr int fx_tile_cols() {
   if (FX_LONG_CAP != 0 && LONG && SPANS && CH == 16) return CH + 2;
   return (!LONG || (CH <= 8 && SPANS && FX_DEFER_LONG == 0 && FX_HALF4 == 0)) ? CH + 1 : CH;
}
// NOHALF (round 4; LONG with CH = 8 only): rows longer than 256 bytes walked in 128-byte segments -- the half-row kernel's lean loop and
// 8 KB of tile per wave (four waves per SIMD) WITHOUT its in-LDS finish: exact start and forward pass from global memory, as with the
// 256-byte segments.  For the chain tables only, whose one dependent LDS read per byte is latency-bound (17-state pattern over 1024-byte
// rows: profiles/r04_half_chain_ab.txt); with the v_perm tables the backward pass is not what long rows wait for (0.562 -> 0.524 ms at
// 1024 bytes, but 1.07 -> 1.17 ms at 400 and 0.493 -> 0.506 ms at 4096: not dispatched).
template <int CH, bool SPANS, int MODE, int SCH, bool RAGGED, bool LONG = false, bool NOHALF = false, bool LATCH = false>
__global__ __launch_bounds__(256, (LONG && CH == 8 && SPANS && FX_DEFER_LONG != 0) ? FX_HALF_WAVES : ((LONG && CH <= 8 && SPANS && FX_HALF4 != 0) ? 4 : 1)) void fx_search_fast(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog,
                                                        FastParams fp, uint8_t* __restrict__ flags, int32_t* __restrict__ from,
                                                        int32_t* __restrict__ to, uint32_t* __restrict__ n_deferred, uint32_t class_map_in_lds,
                                                        uint32_t Lr, uint32_t* __restrict__ clear_next, uint32_t* __restrict__ worklist) {
   // RAGGED: Lr = true row length (16 <= Lr < 16*CH, Lr % 4 == 0); such rows are padded with symbol 255 in LDS.  The aligned
   // instantiation keeps the row length a compile-time constant (the hot path).
   const uint32_t L = (RAGGED || LONG) ? Lr : 16u * CH;
   constexpr uint32_t SEGB = 16u * CH;                   // bytes of one LDS tile row = one segment of a long row
   const uint32_t S = LONG ? ((Lr + SEGB - 1u) / SEGB) : 1u;   // segments per row, the last one shorter when Lr % SEGB != 0
   constexpr bool ragged = RAGGED;
   static_assert(!NOHALF || (LONG && CH == 8), "NOHALF: the 128-byte segment walker of long rows");
   constexpr bool HALFROW = LONG && (CH == 8 || CH == 4) && !NOHALF;   // 256-byte (128-byte) rows staged as two halves of CH chunks (the launcher guarantees Lr == 32 * CH)
   // Match compaction (segment-walking kernels with spans): the exact start and the forward pass are per-ROW work that only rows with a
   // hit need, but a wave pays for them per TILE -- at full price when one lane in 64 has a hit.  Such rows are queued (row, hit group,
   // state entering it) in a per-wave LDS queue instead, and when 64 have gathered -- and once more at the end -- every lane takes one
   // queued row: re-walks its hit group and walks forward, reading the row's bytes from global memory (L2: the tile was just read).
   // The tile pass itself stores their flag (a hit inside the text always yields a span: flag 1); from / to follow at the flush.
   constexpr bool DEFER = FX_DEFER_LONG != 0 && LONG && SPANS;
   constexpr bool HALF4 = FX_HALF4 != 0 && LONG && (CH == 8 || CH == 4) && SPANS && !DEFER;   // the four-waves-per-SIMD tuning of the half-row kernel (see FX_HALF4)
   static_assert(!LONG || ((CH == 16 || CH == 8 || CH == 4) && !RAGGED && (MODE == 0 || MODE == 2 || MODE == 3)), "long rows: CH 16, 8 or 4, first-pass / byte-level modes");
   constexpr bool CHAIN = SCH == 1, WIDE = SCH == 2;
   static_assert(!LATCH || (HALF4 && SCH == 0 && MODE == 0 && CH == 8), "the latched format of R: the half-row first pass of 256-byte rows on the 8-state tables");
   // a state of a RE-WALK (the latch was cleared at the group's start and may have been set by an earlier step): is its base state a hit state?  (group and
   // leading-NUL tests compare with fp.hit_min as ever: in the latched format that is "latched", and they look at states whose predecessor was unlatched)
   auto is_hit = [&](const uint32_t st) -> bool { return LATCH ? (st & FX_LATCH_MASK) >= fp.hit_base : st >= fp.hit_min; };
   constexpr bool LIST = MODE == 4, FIXUP = MODE == 1 || LIST, BYTES = MODE == 2 || MODE == 3, MARKED = MODE == 1 || MODE == 3;
   static_assert(!BYTES || (SCH != 0 && !RAGGED), "byte-level tables: chain or wide v_perm scheme, whole chunks");
   static_assert(!LIST || !RAGGED, "the worklist pass gathers whole-chunk rows");
   if ((MARKED || LIST) && n_deferred[fp.gate_word] == 0) return;   // nothing was left for this pass
   // the "something was deferred" words of consecutive calls alternate: this call's first pass zeroes the NEXT call's word (no
   // memset node per call; nobody reads that word before the next call's second pass)
   if (!MARKED && !LIST && blockIdx.x == 0 && threadIdx.x == 0) {
      clear_next[0] = 0u;
      clear_next[1] = 0u;
      clear_next[2] = 0u;
      clear_next[3] = 0u;
   }
   using F = typename FxF<SCH>::type;
   __shared__ uint2 permR[SCH == 0 ? 256 : 1];
   __shared__ uint2 permA[SCH == 0 ? 256 : 1];
   __shared__ fx_nib wideR[WIDE ? 256 : 1];
   __shared__ fx_nib wideA[WIDE ? 256 : 1];
   extern __shared__ __attribute__((aligned(16))) uint4 tiles[];   // 4 waves x 64*CH cells [+ chain tables] [+ class map]
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave id in an SGPR: tile indices stay scalar
   const int64_t n_tiles = (n + 63) >> 6;
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   // (see FX_ADAPT_CALLS) bit 1 of defer_tiles: this first pass takes part; the persistent word sits behind the two counter groups
   const bool adapt = MODE == 0 && (fp.defer_tiles & 2u) != 0u;
   if (adapt) {
      const uint32_t* hintw = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(n_deferred) & ~uintptr_t(31)) + 32u);
      if (__builtin_amdgcn_readfirstlane(hintw[0]) != 0u) {   // mostly UTF-8 lately: every tile to the follow-up, unread
         for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
            const int64_t row = (t << 6) + lane;
            if (HALFROW && MODE == 0 && SCH == 0 && SPANS && fp.out_mode != 0u) {
               if (lane == 0) reinterpret_cast<uint8_t*>(worklist)[t] = 1u;
            } else if (row < n) flags[row] = FX_NEEDS_GENERAL;
         }
         if (lane == 0) n_deferred[0] = 1u;
         return;
      }
   }
   uint32_t n_def = 0, n_seen = 0;   // (wave-uniform) tiles this wave deferred / was given
   // Start-up: this thread's entries of the 256-entry tables are READ first, then the first tiles' global loads go out, and only then are the
   // entries written to LDS: the block's two start-up latencies -- tables from L2, rows from HBM -- overlap instead of adding up, and the wait
   // for the table entries (the older loads: vmcnt counts in order) does not wait for the rows.  (The marked-tile passes read the flags first;
   // the worklist pass gathers per tile.)
   uint2 t_r = make_uint2(0, 0), t_a = make_uint2(0, 0);
   if (!CHAIN) {
      const FxpHeader* h0 = reinterpret_cast<const FxpHeader*>(prog);
      t_r = reinterpret_cast<const uint2*>(prog + (WIDE ? (BYTES ? h0->off_bw16R : h0->off_w16R) : (LATCH ? h0->off_fastRL : h0->off_fastR)))[threadIdx.x];
      t_a = reinterpret_cast<const uint2*>(prog + (WIDE ? (BYTES ? h0->off_bw16A : h0->off_w16A) : h0->off_fastA))[threadIdx.x];
   }
   __builtin_amdgcn_sched_barrier(0);   // (the table reads stay ahead of the tiles' loads: their addresses wait for the header's offsets)
   constexpr int DEPTH = FX_PREFETCH_DEPTH;
   constexpr bool EARLY = !MARKED && !LIST;
   uint4 stage[LIST ? 1 : DEPTH][CH];
   bool live[LIST ? 1 : DEPTH];
   if constexpr (EARLY) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
         live[d] = true;
         if constexpr (LONG) PREFETCH_SEG(stage[d], wave_global + d * wave_stride, S - 1u, true);
         else PREFETCH_TILE(stage[d], wave_global + d * wave_stride, true);
      }
   }
   __shared__ uint32_t fwd_q[DEFER ? 4 * 64 * 2 : 1];   // match compaction: per wave 64 x (row, hit group | entry state << 16)
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   // chain scheme: symbol -> 2*column map (512 B), then T_R, then T_A, behind the tiles
   constexpr int COLS = fx_tile_cols<CH, SPANS, LONG, NOHALF>();   // chunk columns per row in LDS: the row's chunks [+ the end-of-row column / the look-ahead columns]
   constexpr bool CAP = FX_LONG_CAP != 0 && LONG && SPANS && CH == 16;   // (see fx_tile_cols)
   // ... for rows below FX_LONG_NT_MIN bytes (wave-uniform): measured in one allocation against a build without it (gpurun call r05_c12): rows of
   // 400 B 0.889 -> 0.850 ms, 1024 B 0.536 -> 0.518 ms, but 4096 B 0.489 -> 0.498 ms (the look-ahead copy per segment costs more than the rare
   // re-read saves); the 128-byte segment walker of the chain tables (NOHALF) loses its fourth wave per SIMD to the two columns: 0.66 -> 0.78 ms, not built
   const bool cap_on = CAP && Lr < FX_LONG_NT_MIN;
   // (HALF4: the four shared end-of-row cells come first behind the tiles, see below)
   uint16_t* cmap = reinterpret_cast<uint16_t*>(tiles + 4 * 64 * COLS + (HALF4 ? 4 : 0));
   const uint32_t tr_bytes = BYTES ? h->byte_TR_bytes : h->chain_TR_bytes, ta_bytes = BYTES ? h->byte_TA_bytes : h->chain_TA_bytes;
   const uint32_t chain_bytes = CHAIN ? ((512u + tr_bytes + ta_bytes + 15u) & ~15u) : 0u;
   const uint8_t* TRp = reinterpret_cast<const uint8_t*>(cmap) + 512;
   const uint8_t* TAp = TRp + (CHAIN ? tr_bytes : 0u);
   if (CHAIN) {
      const uint16_t* g = reinterpret_cast<const uint16_t*>(prog + (BYTES ? h->off_byte_cls : h->off_chain_cls));
      const uint16_t* gr = reinterpret_cast<const uint16_t*>(prog + (BYTES ? h->off_byte_TR : h->off_chain_TR));
      const uint16_t* ga = reinterpret_cast<const uint16_t*>(prog + (BYTES ? h->off_byte_TA : h->off_chain_TA));
      const uint32_t nr = tr_bytes / 2, na = ta_bytes / 2;
      fx_stage_chain(cmap, g, gr, ga, nr, na);
   } else if (WIDE) {
      reinterpret_cast<uint2*>(wideR)[threadIdx.x] = t_r;
      reinterpret_cast<uint2*>(wideA)[threadIdx.x] = t_a;
   } else {
      permR[threadIdx.x] = t_r;   // 256 threads = 256 symbol ids (ids >= 128 are all-dead rows unless FXP_F_FAST_UTF8)
      permA[threadIdx.x] = t_a;
   }
   __syncthreads();
   // symbol -> F tables of the two directions (the chain scheme shares one class map)
   using TabT = typename std::conditional<CHAIN, uint16_t, typename std::conditional<WIDE, fx_nib, uint2>::type>::type;
   const TabT* tabR = CHAIN ? reinterpret_cast<const TabT*>(cmap) : (WIDE ? reinterpret_cast<const TabT*>(wideR) : reinterpret_cast<const TabT*>(permR));
   const TabT* tabA = CHAIN ? reinterpret_cast<const TabT*>(cmap) : (WIDE ? reinterpret_cast<const TabT*>(wideA) : reinterpret_cast<const TabT*>(permA));
   const bool raw = BYTES || (h->flags & FXP_F_RAW_BYTES) != 0;   // literal search, byte-level tables: bytes are symbols, nothing is decoded or deferred
   const bool utf8 = !raw && (FIXUP || fp.defer_tiles != 0);   // first pass: defer whole tiles that hold a byte >= 0x80
   // second pass only: BMP class map (page index + pages) for the in-LDS UTF-8 decode, placed behind the four tiles
   const uint16_t* page_p = reinterpret_cast<const uint16_t*>(prog + h->off_cls_page);
   const uint16_t* pages_p = reinterpret_cast<const uint16_t*>(prog + h->off_cls_pages);
   if (FIXUP && class_map_in_lds) {
      uint16_t* l16 = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(tiles + 4 * 64 * COLS) + chain_bytes);
      const uint32_t n16 = 1024u + h->n_pages * 64u;
      {   // 16-byte pieces (blob offsets and the LDS offset are multiples of 16): one round trip instead of one per 256 entries
         uint4* l4 = reinterpret_cast<uint4*>(l16);
         for (uint32_t i = threadIdx.x; i < n16 / 8u; i += 256u)
            l4[i] = i < 128u ? reinterpret_cast<const uint4*>(page_p)[i] : reinterpret_cast<const uint4*>(pages_p)[i - 128u];
      }
      __syncthreads();
      page_p = l16;
      pages_p = l16 + 1024;
   }
   const fxrow::ClassTables ct{page_p, pages_p, reinterpret_cast<const uint16_t*>(prog + h->off_bound_cls),
                               reinterpret_cast<const int32_t*>(prog + h->off_bounds), h->n_bounds};
   const uint32_t sym_ffff = 128u + h->cls_ffff;
   // one extra chunk column per row holds what follows the text: the trailing NUL (symbol 0), then KILL symbols (0xFE, whose table
   // row is all-dead), so the forward pass reads "past the end" like any other position.  Written once, never overwritten.
   uint4* tile = tiles + wave * (64 * COLS);
   if constexpr (COLS > CH && !CAP) tile[tile_cell(lane, CH)] = make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
   // HALF4: one shared end-of-row cell per wave behind the four tiles (the speculative forward pass on the half row in LDS reads it)
   uint4* const eor_cell = tiles + 4 * 64 * COLS + wave;
   if (HALF4 && lane == 0) *eor_cell = make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
   const uint8_t* const eor8 = HALF4 ? reinterpret_cast<const uint8_t*>(eor_cell) : nullptr;
   // rows of fewer WHOLE chunks than the instantiation has: their unused chunk columns hold the inert symbol 255 from here on
   // (the staging stores never touch them; the decode passes re-pad per tile because they rewrite the cells)
   const bool whole = RAGGED && (Lr & 15u) == 0u;
   if (whole)
      for (uint32_t k = Lr >> 4; k < (uint32_t)CH; ++k) tile[tile_cell(lane, k)] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);

   bool any_deferred = false;   // wave-uniform: this wave deferred at least one tile to the second pass
   STAMP_DECL;
   // One tile: `stage` holds its global loads (issued DEPTH tiles ago); once they are stored to LDS the same registers take the
   // loads of tile t_next, which stay in flight while this and the following DEPTH-1 tiles are scanned.
   // marked-tile passes: does tile t hold a row the pass before left behind (FX_NEEDS_GENERAL)?  wave-uniform
   auto tile_marked = [&](const int64_t t) -> bool {
      const int64_t rr = (t << 6) + lane;
      return __builtin_amdgcn_ballot_w64(rr < n && flags[rr] == FX_NEEDS_GENERAL) != 0;
   };
   const uint32_t list_count = LIST ? n_deferred[1] : 0u;
   // PACKED results straight from the half-row first pass (round 5): the tile's flag word is the wave's ballot, spans are stored narrow,
   // and a deferred tile is recorded in a byte per tile (the follow-up reads those instead of the rows' flag bytes)
   constexpr bool PACK_OK = HALFROW && MODE == 0 && SCH == 0 && SPANS && !DEFER;
   const uint32_t out_mode = PACK_OK ? fp.out_mode : 0u;
   uint8_t* const marks = reinterpret_cast<uint8_t*>(worklist);   // (packed calls only: the half-row first pass lists no rows)
   // Forward pass over the symbol stream of one row from text index j (state `cur`, 0 = this lane does not walk): first 32 symbols
   // straight-line -- five aligned 8-byte row reads, a byte shift to start exactly at j, all 32 table lookups issued before the
   // chain; per 8-byte group only "any accept" (v_max3) + entry state are kept and the last accepting group is re-walked for the
   // exact byte -- then 8 symbols per round trip while any lane is alive.  `src`: the LDS tile (FROM_GLOBAL false) or the row itself
   // in global memory; Lx: the length of what `src` holds (the virtual end-of-row symbols follow it).  mm: max_match so far, updated.
   // (capd: the 40 bytes from the hit group on, captured while their segment was in LDS -- CAP; lanes that start at the leading NUL read the
   //  row's first bytes from global memory as before)
   auto forward_pass = [&](auto from_global, const uint8_t* src, const uint32_t Lx, uint32_t cur, uint32_t& mm, uint32_t j, const uint32_t* capd = nullptr,
                           const bool use_cap = false) {
      constexpr bool FG = decltype(from_global)::value;
      if (__builtin_amdgcn_ballot_w64(cur != 0) == 0) return;
      uint32_t o[8];
      const uint8_t* const eor = FG ? nullptr : eor8;
      if (capd != nullptr) {
         if (__builtin_amdgcn_ballot_w64(cur != 0 && !use_cap) != 0) fetch32<RAGGED, FG>(o, src, lane, j, Lx, eor, cur != 0 && !use_cap);
         else
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = 0u;
         const uint32_t sh = j & 7u, up = 0u - ((sh >> 2) & 1u);
         uint32_t e[9];
#pragma unroll
         for (int k = 0; k < 9; ++k) e[k] = (up & capd[k + 1]) | (~up & capd[k]);
#pragma unroll
         for (int k = 0; k < 8; ++k) o[k] = use_cap ? __builtin_amdgcn_alignbyte(e[k + 1], e[k], sh & 3u) : o[k];
      } else
         fetch32<RAGGED, FG>(o, src, lane, j, Lx, eor, cur != 0);
#ifndef FX_FWD_GB_LDS
#define FX_FWD_GB_LDS 4
#endif
      constexpr int GB = (FG && DEFER) ? FX_FWD_GB : (HALF4 ? 2 : FX_FWD_GB_LDS);   // 8-symbol groups whose lookups are issued together (fewer in the flush: registers)
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
            const bool hit = mx >= fp.acc_min;
            gl = hit ? (uint32_t)(gb + g) : gl;
            el = hit ? entry : el;
            blo = hit ? o[2 * (gb + g)] : blo;
            bhi = hit ? o[2 * (gb + g) + 1] : bhi;
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
      STAMP(4);
      // matches longer than the window: 8 symbols per round trip.  The stream is a rolling window of two aligned 8-byte groups
      // (t0, t1); the group after them is read one round ahead, so a round waits for its table lookups only.  Wave-uniform: dead
      // lanes (state 0 is absorbing and below acc_min) ride along.
      if (__builtin_amdgcn_ballot_w64(cur != 0) != 0) {
         const uint32_t sh = j & 7u, up = 0u - ((sh >> 2) & 1u);
         uint32_t gb = j & ~7u;
         uint32_t t0[2], t1[2];
         group_words<RAGGED, FG>(t0[0], t0[1], src, lane, gb, Lx, eor, cur != 0);
         group_words<RAGGED, FG>(t1[0], t1[1], src, lane, gb + 8u, Lx, eor, cur != 0);
         do {
            uint32_t t2[2];
            group_words<RAGGED, FG>(t2[0], t2[1], src, lane, gb + 16u, Lx, eor, cur != 0);
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
         } while (__builtin_amdgcn_ballot_w64(cur != 0) != 0);
      }
   };
   // ---- match compaction: the per-wave queue of rows with a hit, and its flush (see DEFER above) ------------------------------------
   uint32_t* const fq = fwd_q + (DEFER ? wave * 128u : 0u);
   uint32_t fq_n = 0;   // entries in the queue (wave-uniform)
   auto flush_queue = [&]() {
      if (fq_n == 0u) return;
      const bool on = lane < fq_n;
      const uint32_t qrow = on ? fq[lane] : 0u, ge = on ? fq[64u + lane] : 0u;
      fq_n = 0;
      const uint32_t g = ge & 0xFFFFu, e = SCH == 0 ? (ge >> 16) * 0x01010101u : (ge >> 16);
      const uint8_t* rp = rows + (int64_t)qrow * (int64_t)L;
      // exact byte of the leftmost hit: re-walk the hit group (8 bytes of the row, from global memory; the row may end inside the group)
      uint2 rw = make_uint2(0, 0);
      uint32_t nv = 8;
      if (on) {
         if (g * 8u + 8u <= L) rw = *reinterpret_cast<const uint2*>(rp + g * 8u);
         else {
            const uint2 r = *reinterpret_cast<const uint2*>(rp + L - 8u);
            nv = L - g * 8u;
            const uint64_t v = (((uint64_t)r.y << 32) | r.x) >> (64u - 8u * nv);
            rw = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
         }
      }
      F f[8];
      lookup8(f, rw.x, rw.y, tabR);
      uint32_t st = e, loc = 0;
#pragma unroll
      for (int i = 7; i >= 0; --i) {
         const uint32_t nx = fxstep(f[i], st, TRp);
         const bool ok = (uint32_t)i < nv;   // (per lane)
         st = ok ? nx : st;
         loc = ok && nx >= fp.hit_min ? (uint32_t)i : loc;
      }
      const uint32_t s = g * 8u + 2u + loc;   // wrapped index of the start
      uint32_t mm = fp.lit_len != 0 ? s + fp.lit_len : 0u;
      forward_pass(std::true_type{}, rp, (uint32_t)L, (on && fp.lit_len == 0) ? fp.A_init : 0u, mm, s - 2u);
      if (on) {   // api_internal_m.F90:140-148 (a start inside the text: from = s - 1 >= 1)
         const int32_t tt = mm >= L + 2u ? (int32_t)L : (int32_t)mm - 2;
         const bool okm = mm != 0 && tt > 0;
         from[qrow] = okm ? (int32_t)(s - 1u) : 0;
         to[qrow] = okm ? tt : 0;
         if (!okm) flags[qrow] = 0;   // (cannot happen: a hit at s means a non-empty match starts there)
      }
   };
   // `live`: the tile in `stage` is to be scanned (always, except in the marked-tile passes); on return it says so for t_next
   auto do_tile = [&](uint4 (&stage)[CH], bool& live, const int64_t t, const int64_t t_next) {
      const int64_t row0 = t << 6;
      n_seen += 1u;
      STAMP(7);
      int64_t row = row0 + lane;   // the row this lane owns and whether it exists
      bool row_ok = row < n;
      bool defer_early = false;
      if (LIST) {
         // worklist pass: lane r gathers row worklist[64 t + r] straight into its own cells (one row per lane: nothing to transpose)
         const uint32_t slot = (uint32_t)row0 + lane;
         row_ok = slot < list_count;
         row = row_ok ? (int64_t)worklist[slot] : 0;
         const uint4* src = reinterpret_cast<const uint4*>(rows + row * (int64_t)(16 * CH));
#pragma unroll
         for (int k = 0; k < CH; ++k) stage[k] = row_ok ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
         for (int k = 0; k < CH; ++k) tile[tile_cell(lane, k)] = stage[k];
      }
      uint32_t state = fp.R_start;
      uint32_t gsel = 0xFFFFFFFFu, esel = 0;   // leftmost 8-byte group holding a hit, and the state entering it
      uint32_t na = 0;
      uint32_t s_half = 0, mm_half = 0;   // half-row staging: start / max_match resolved while the right half was in LDS
      uint32_t cap[CAP ? 10 : 1];          // CAP: the five 8-byte groups from the leftmost hit group on (row bytes, then NUL, then KILL symbols)
#pragma unroll
      for (int i = 0; i < (CAP ? 10 : 1); ++i) cap[i] = 0u;
      for (uint32_t seg = S - 1u;; --seg) {   // one pass unless LONG: the row's 256-byte segments, right to left
         if (!LIST) {
            const bool process = live;
            // cheap sampled look at the staged bytes: a tile that shows a byte >= 0x80 here is deferred without being scanned
            // (tiles whose only such bytes hide in the unsampled registers are caught after the backward pass below)
            if (MODE == 0 && utf8 && seg == S - 1u) {
               const uint32_t smp = stage[0].x | stage[0].w | stage[CH / 2].y | stage[CH - 1].z;
               defer_early = __builtin_amdgcn_ballot_w64((smp & 0x80808080u) != 0) != 0;
            }
            if (process) STORE_TILE(stage);
            if constexpr (CAP) {
               // the row's LAST segment (the first one staged): what follows the text -- the trailing NUL, then KILL symbols -- behind its
               // seg_len bytes: the chunk the row ends in is patched, up to three chunks behind it are written (the look-ahead columns when
               // the segment is whole).  Wave-uniform positions, the lane's own cells.
               if (cap_on && process && seg == S - 1u) {
                  const uint32_t sl = Lr - seg * SEGB;   // 1 .. SEGB
                  const uint32_t kt = sl >> 4, nb = sl & 15u;
                  if (nb != 0u) {
                     uint4 c = tile[tile_cell(lane, kt)];
                     c.x = fx_tail_word(c.x, 0u, nb);
                     c.y = fx_tail_word(c.y, 4u, nb);
                     c.z = fx_tail_word(c.z, 8u, nb);
                     c.w = fx_tail_word(c.w, 12u, nb);
                     tile[tile_cell(lane, kt)] = c;
                  } else tile[tile_cell(lane, kt)] = make_uint4(0xFEFEFE00u, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
                  for (uint32_t k = kt + 1u; k <= kt + 3u && k < (uint32_t)COLS; ++k) tile[tile_cell(lane, k)] = make_uint4(0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu, 0xFEFEFEFEu);
               }
            }
            STAMP(0);
            // (wave-private tile: LDS operations of one wave complete in order, no barrier needed)
            // the ONE place the staging registers are reloaded (a second load site would meet this one in a register merge at the
            // loop's back edge: copies behind a vmcnt(0)).  A tile the pass skips is "loaded" with zero valid bytes.
            const bool last = !LONG || !process || defer_early || seg == 0u;   // nothing more of this tile is wanted
            if (last) live = MARKED ? tile_marked(t_next) : true;
            if constexpr (LONG) PREFETCH_SEG(stage, last ? t_next : t, last ? S - 1u : seg - 1u, last ? live : true);
            else PREFETCH_TILE(stage, t_next, live);
            if (!process) return;
         }
         if (defer_early) {
            if (PACK_OK && out_mode != 0u) {
               if (lane == 0) marks[t] = 1u;
            } else if (row_ok) flags[row] = FX_NEEDS_GENERAL;
            any_deferred = true;
            n_def += 1u;
            return;
         }
         if (FIXUP) {
            // On-device UTF-8 decode, in place in LDS: lane r rewrites its own row cell by cell into fast-path symbol ids
            // (fxrow::translate_cell16).  The 4 bytes before / after a cell are taken from the ORIGINAL neighbours: the
            // previous cell's last dword is kept in a register, the next cell is read before anything overwrites it.
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
         if (ragged && (!whole || FIXUP)) na |= pad_rows<CH>(tile, lane, Lr);

         // ---- right-to-left pass: reverse unanchored DFA; the LAST hit seen is the leftmost start ----
         // software pipeline in 8-byte groups: the 8 lookups of the next group are in flight (lgkmcnt <= 15 stays
         // expressible) while the state chain of the current group runs.  Per group only "did any state hit" and the
         // group's entry state are kept; the exact byte is recovered afterwards by re-walking ONE group per row.
         STAMP(1);
#ifdef FX_EXP_NOCOMPUTE
         {   // experiment: memory path only (loads, LDS staging, outputs), no automaton work
            uint32_t acc = na;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
               const uint4 c = tile[tile_cell(lane, k)];
               acc |= c.x | c.y | c.z | c.w;
            }
            na = acc;
            if (LONG && seg != 0u) continue;
            if (row0 + lane < n) {
               flags[row0 + lane] = (uint8_t)(acc & 1u);
               if (SPANS) {
                  from[row0 + lane] = (int32_t)acc;
                  to[row0 + lane] = (int32_t)(acc >> 1);
               }
            }
            return;
         }
#endif
         // LONG: bytes of this segment (256, or what is left of the row in its last segment): groups behind the row end are not
         // walked, the group the row ends in is walked over its valid bytes only -- no pad symbol is needed, so the byte-level
         // tables work on rows of any length
         const uint32_t seg_len = LONG ? (Lr - seg * SEGB < SEGB ? Lr - seg * SEGB : SEGB) : SEGB;
         const uint32_t gbase = LONG ? seg * (SEGB / 8u) : 0u;   // 8-byte groups to the left of this segment
         // the chunk loop in two instantiations: every group whole (always, unless LONG and this is a row's short last segment) or
         // with the per-group byte counts
         // (round 5: the loop records the hit group's number INSIDE the segment -- an inline constant of the select; with `gbase + 2 k` every group
         //  first moved a scalar into a vector register: 16 v_mov_b32 per 128 bytes of the half-row kernel)
         uint32_t gloc = 0xFFFFFFFFu;
         auto walk = [&](auto whole_groups) {
            constexpr bool WG = decltype(whole_groups)::value;
            F fa[8], fb[8];
            // (a row's short last segment: the walk starts at the chunk the row ends in -- chunks behind it are not even looked up; round 5:
            //  400-byte rows walked their 144-byte segment's seven empty chunks through the table reads)
            const uint32_t kc = WG ? (uint32_t)CH : ((seg_len + 15u) >> 4);   // chunks that hold text: 1 .. CH, wave-uniform
            uint4 wk = tile[tile_cell(lane, kc - 1u)], wn = make_uint4(0, 0, 0, 0);
            if (CH >= 2 && !HALF4 && kc >= 2u) wn = tile[tile_cell(lane, kc - 2u)];
            lookup8(fa, wk.z, wk.w, tabR);
#pragma unroll
            for (int k = CH - 1; k >= 0; --k) {
               if (!WG && (uint32_t)k >= kc) continue;
               // valid bytes of this chunk's upper / lower group (8 unless LONG and the row ends here)
               const uint32_t nhi = WG ? 8u : (seg_len >= 16u * k + 16u ? 8u : (seg_len > 16u * k + 8u ? seg_len - (16u * k + 8u) : 0u));
               const uint32_t nlo = WG ? 8u : (seg_len >= 16u * k + 8u ? 8u : (seg_len > 16u * k ? seg_len - 16u * k : 0u));
               if (LONG) {
                  if (nhi == 8u) na |= wk.x | wk.y | wk.z | wk.w;
                  else {   // (wave-uniform) only the row's own bytes count
                     const uint32_t w4[4] = {wk.x, wk.y, wk.z, wk.w};
#pragma unroll
                     for (int i = 0; i < 4; ++i) {
                        const uint32_t at = 16u * k + 4u * i;
                        if (at + 4u <= seg_len) na |= w4[i];
                        else if (at < seg_len) na |= w4[i] & ~(0xFFFFFFFFu << (8u * (seg_len - at)));
                     }
                  }
               } else if (!ragged || (whole && !FIXUP && (uint32_t)k < (Lr >> 4))) na |= wk.x | wk.y | wk.z | wk.w;
               lookup8(fb, wk.x, wk.y, tabR);
               __builtin_amdgcn_sched_barrier(0);
               if (nhi != 0u) {
                  const uint32_t entry = state;
                  const uint32_t mx = nhi == 8u ? chain8_back<F, HALF4, LATCH>(fa, state, TRp) : chain8_back_n<F, LATCH>(fa, state, TRp, nhi);
                  gloc = mx >= fp.hit_min ? (uint32_t)(2 * k + 1) : gloc;
                  esel = mx >= fp.hit_min ? entry : esel;
                  if (LATCH) state &= FX_LATCH_MASK;
                  asm volatile("" : "+v"(esel));   // select now: otherwise all 2*CH entry states stay live until after the loop
               }
               __builtin_amdgcn_sched_barrier(0);
               if (k >= 1) {
                  if (HALF4) wk = tile[tile_cell(lane, k - 1)];   // (no second chunk of prefetch: registers)
                  else wk = wn;
                  lookup8(fa, wk.z, wk.w, tabR);
                  if (k >= 2 && !HALF4) wn = tile[tile_cell(lane, k - 2)];
               }
               __builtin_amdgcn_sched_barrier(0);
               if (nlo != 0u) {
                  const uint32_t entry = state;
                  const uint32_t mx = nlo == 8u ? chain8_back<F, HALF4, LATCH>(fb, state, TRp) : chain8_back_n<F, LATCH>(fb, state, TRp, nlo);
                  gloc = mx >= fp.hit_min ? (uint32_t)(2 * k) : gloc;
                  esel = mx >= fp.hit_min ? entry : esel;
                  if (LATCH) state &= FX_LATCH_MASK;
                  asm volatile("" : "+v"(esel));
               }
               __builtin_amdgcn_sched_barrier(0);
            }
         };
         if constexpr (!LONG) walk(std::true_type{});
         else {
            if (seg_len == SEGB) walk(std::true_type{});
            else walk(std::false_type{});
         }
         gsel = gloc != 0xFFFFFFFFu ? gbase + gloc : gsel;   // (a hit in this segment is further left than any recorded so far)
         if constexpr (CAP) {
            // a hit in THIS segment is the leftmost so far: the 40 bytes from its group on go to registers while they are in LDS (its own
            // chunks and, past the segment's right edge, the look-ahead columns)
            const bool newhit = gloc != 0xFFFFFFFFu;   // (a hit in THIS segment)
            if (cap_on && __builtin_amdgcn_ballot_w64(newhit) != 0) {
               const uint32_t gl = newhit ? gloc : 0u;
#pragma unroll
               for (int q = 0; q < 5; ++q) {
                  const uint32_t p = (gl + (uint32_t)q) << 3;   // local byte position: < 16 * (CH + 2)
                  const uint2 r = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, p >> 4) << 4) + (p & 8u));
                  cap[2 * q] = newhit ? r.x : cap[2 * q];
                  cap[2 * q + 1] = newhit ? r.y : cap[2 * q + 1];
               }
            }
            // the segment's first 32 bytes become the look-ahead of the one to its left (before that one is stored over them)
            if (cap_on && seg != 0u) {
               const uint4 h0 = tile[tile_cell(lane, 0)], h1 = tile[tile_cell(lane, 1)];
               tile[tile_cell(lane, CH)] = h0;
               tile[tile_cell(lane, CH + 1)] = h1;
            }
         }
         if constexpr (HALFROW && !DEFER) {
            // Half-row staging (two segments per row): the right half is in LDS now and will be overwritten by the left one.  A row
            // whose leftmost hit SO FAR lies here gets its exact start and -- speculatively: a hit in the left half supersedes it --
            // its forward pass now, from LDS (this half + the end-of-row column), instead of from global memory afterwards.
            if (seg == 1u && __builtin_amdgcn_ballot_w64(gsel != 0xFFFFFFFFu) != 0) {
               const uint32_t gl8 = gsel != 0xFFFFFFFFu ? gsel - SEGB / 8u : 0u;   // group inside this half
               const uint2 rw = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, gl8 >> 1) << 4) + ((gl8 & 1u) << 3));
               F f[8];
               lookup8(f, rw.x, rw.y, tabR);
               uint32_t st = esel, loc = 8;
#pragma unroll
               for (int i = 7; i >= 0; --i) {
                  st = fxstep(f[i], st, TRp);
                  loc = is_hit(st) ? (uint32_t)i : loc;
               }
               s_half = gsel != 0xFFFFFFFFu ? gsel * 8u + 2u + loc : 0u;
               if (SPANS && fp.lit_len == 0) {
                  uint32_t mmh = 0;
                  // coordinates of this half: text index j - SEGB, length SEGB; max_match moves back by SEGB afterwards
                  forward_pass(std::false_type{}, tb, SEGB, s_half != 0u ? fp.A_init : 0u, mmh, s_half != 0u ? s_half - 2u - SEGB : 0u);
                  mm_half = mmh != 0u ? mmh + SEGB : 0u;
               }
            }
         }
         if (!LONG || seg == 0u) break;
      }
      STAMP(2);
      uint32_t s = 0;          // wrapped start index (1 = leading NUL, j+2 for text byte j), 0 = none
      if constexpr (DEFER) {
         // match compaction: the exact start is found at the flush; here only "a start inside the text" (2) / "at the leading NUL" (1)
         s = gsel != 0xFFFFFFFFu ? 2u : 0u;
         const F fz = tabR[0];   // leading NUL
         state = fxstep(fz, state, TRp);
         s = state >= fp.hit_min ? 1u : s;
         if (!row_ok) s = 0;
      } else {
         // exact byte of the leftmost hit: re-walk the selected group (every lane walks exactly one group)
         // (half-row staging: a hit group of the right half was resolved while that half was in LDS -- s_half; the re-walk here is
         //  for hit groups of the left half, which is what the tile holds now)
         const bool here = !HALFROW || gsel < SEGB / 8u;
         const uint32_t g = (gsel != 0xFFFFFFFFu && here) ? gsel : 0u;
         uint2 rw;
         uint32_t nv = 8;   // LONG: valid bytes of the group (the row may end inside it)
         if (CAP && cap_on) {   // (captured while its segment was in LDS)
            rw = make_uint2(cap[0], cap[1]);
            if (g * 8u + 8u > L) nv = L - g * 8u;   // the row ends inside the group: only its text bytes were walked
         } else if (LONG && !HALFROW) {   // (its segment left the tile: from global memory)
            rw = make_uint2(0, 0);
            if (row_ok && (FX_LIVE_PRED == 0 || gsel != 0xFFFFFFFFu)) {   // (lanes without a hit read nothing)
               const uint8_t* rp = rows + row * (int64_t)L;
               if (g * 8u + 8u <= L) rw = *reinterpret_cast<const uint2*>(rp + g * 8u);
               else {   // the row's last 8 bytes, shifted down to the group's place (nothing behind the row is read)
                  const uint2 r = *reinterpret_cast<const uint2*>(rp + L - 8u);
                  nv = L - g * 8u;
                  const uint64_t v = (((uint64_t)r.y << 32) | r.x) >> (64u - 8u * nv);
                  rw = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
               }
            }
         } else rw = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, g >> 1) << 4) + ((g & 1u) << 3));
         uint32_t loc = 8;
         // (half-row staging: skipped when no lane's leftmost hit lies in the left half -- config 3: every match sits in the right one)
         if (!HALFROW || __builtin_amdgcn_ballot_w64(gsel != 0xFFFFFFFFu && here) != 0) {
            F f[8];
            lookup8(f, rw.x, rw.y, tabR);
            uint32_t st = esel;
#pragma unroll
            for (int i = 7; i >= 0; --i) {
               const uint32_t nx = fxstep(f[i], st, TRp);
               const bool on = !LONG || HALFROW || (uint32_t)i < nv;   // (per lane)
               st = on ? nx : st;
               loc = on && is_hit(nx) ? (uint32_t)i : loc;
            }
         }
         s = gsel != 0xFFFFFFFFu ? (here ? g * 8u + 2u + loc : s_half) : 0u;
         const F fz = tabR[0];   // leading NUL
         state = fxstep(fz, state, TRp);
         s = state >= fp.hit_min ? 1u : s;
         if (LONG && !row_ok) s = 0;   // (no row: the forward pass would read global memory)
      }
      // where the forward pass finds the row's bytes (lanes past the last row of a long-row batch read row 0: every lane fetches)
      const uint8_t* fsrc = LONG ? rows + (row_ok ? row : 0) * (int64_t)L : tb;
      // Bytes >= 0x80 in the first pass: without UTF-8 tables the ROW goes to the general kernel's fix-up; with them the whole
      // TILE is deferred to the second pass (wave-uniform; the raw-byte scan above is discarded and the
      // forward walk below is skipped).
      const bool row_hi = MODE == 0 && !raw && (na & 0x80808080u) != 0;
      const bool defer_tile = MODE == 0 && utf8 && __builtin_amdgcn_ballot_w64(row_hi) != 0;
      // byte-level tables: the backward pass ended in the INVALID state -> structurally invalid UTF-8, the row-level fix-up redoes it
      const bool exception = (BYTES || (MODE == 0 && fp.inv_on != 0)) && state == fp.inv;
      const bool nonascii = (row_hi && !utf8) || defer_tile || exception;
      if (BYTES || MODE == 0) {
         // exception rows are appended to the worklist of the decode pass: one atomic per tile that has any.  A first pass
         // without decode tables lists its own leftovers (rows with bytes >= 0x80, overlap rows) for the row-level fix-up.
         const bool listed = (BYTES ? exception : (worklist != nullptr && (exception || (row_hi && !utf8)))) && row_ok;
         const uint64_t em = __builtin_amdgcn_ballot_w64(listed);
         if (em != 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&n_deferred[1], (uint32_t)__builtin_popcountll(em));
            base = __builtin_amdgcn_readfirstlane(base);
            if (listed) worklist[base + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = (uint32_t)row;
         }
      }

      // ---- left-to-right pass from the leftmost start: anchored DFA, longest accept (api_internal_m.F90:119-148) ----
      // flags only: a start inside the text always gives to >= from >= 1, so only starts at the leading NUL need the walk.
      // The row is extended virtually: position L holds the trailing NUL (byte 0 -> F[0]), later positions kill the state;
      // an accept after consuming position `pos` gives max_match = pos + 3 for text bytes and for the trailing NUL alike.
      // (half-row staging: a start in the right half had its forward pass while that half was in LDS -- mm_half)
      const bool fwd_here = !HALFROW || DEFER || s != s_half || s == 0u;
      // match compaction: rows with a start inside the text are queued (their flag is known: such a start always yields a span);
      // only starts at the leading NUL (^-anchored patterns) walk forward here
      const bool queued = DEFER && s >= 2u && !nonascii;
      uint32_t cur = (s != 0 && fwd_here && !queued && !nonascii && (SPANS || s == 1) && fp.lit_len == 0) ? fp.A_init : 0u;
      uint32_t mm = (fp.lit_len != 0 && s != 0) ? s + fp.lit_len : 0u;   // max_match (wrapped index of the byte after the match)
      uint32_t j = s >= 2 ? s - 2 : 0;      // 0-based text index of the next byte to consume
      if (s == 1) {
         const F f = tabA[0];
         cur = fxstep(f, cur, TAp);
         mm = cur >= fp.acc_min ? 2u : 0u;
      }
      STAMP(3);
      if constexpr (CAP) forward_pass(std::integral_constant<bool, LONG>{}, fsrc, (uint32_t)L, cur, mm, j, cap, cap_on && s >= 2u);   // (always the array itself: a pointer that may be null would send it to scratch memory)
      else forward_pass(std::integral_constant<bool, LONG>{}, fsrc, (uint32_t)L, cur, mm, j);
      if (HALFROW && !DEFER && !fwd_here && fp.lit_len == 0) mm = mm_half;
      STAMP(5);
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
      if (queued) flag = 1;
      if (nonascii) flag = FX_NEEDS_GENERAL;
      any_deferred = any_deferred || defer_tile;
      n_def += defer_tile ? 1u : 0u;
      if (PACK_OK && out_mode != 0u) {
         // (a deferred tile's word is written by the follow-up; rows behind the batch's end contribute a zero bit)
         const uint64_t m = __builtin_amdgcn_ballot_w64(row_ok && flag == 1u);
         if (lane == 0) {
            marks[t] = defer_tile ? 1u : 0u;
            if (!defer_tile) reinterpret_cast<uint64_t*>(flags)[t] = m;
         }
         if (row_ok && !defer_tile) {
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
      } else if (row_ok) {
         flags[row] = (uint8_t)flag;
         if (SPANS && !queued) {
            from[row] = fr;
            to[row] = tt;
         }
      }
      if constexpr (DEFER) {
         const uint64_t qm = __builtin_amdgcn_ballot_w64(queued);
         if (qm != 0) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(qm);
            if (fq_n + cnt > 64u) flush_queue();
            if (queued) {
               const uint32_t slot = fq_n + (uint32_t)__builtin_popcountll(qm & ((1ull << lane) - 1ull));
               fq[slot] = (uint32_t)row;
               fq[64u + slot] = gsel | ((SCH == 0 ? (esel & 0xFFu) : esel) << 16);
            }
            fq_n += cnt;
         }
      }
   };
   if constexpr (LIST) {
      live[0] = true;
      for (int64_t t = wave_global; (uint64_t)(t << 6) < list_count; t += wave_stride) do_tile(stage[0], live[0], t, t);
   } else {
      // DEPTH tiles of global loads in flight per wave (HBM latency under load is several microseconds).  The marked-tile passes
      // run the same pipeline: a tile they skip costs one read of its flags and 16 loads that are range-checked away.
      if constexpr (!EARLY) {
#pragma unroll
         for (int d = 0; d < DEPTH; ++d) {
            live[d] = MARKED ? tile_marked(wave_global + d * wave_stride) : true;
            if constexpr (LONG) PREFETCH_SEG(stage[d], wave_global + d * wave_stride, S - 1u, live[d]);
            else PREFETCH_TILE(stage[d], wave_global + d * wave_stride, live[d]);
         }
      }
      for (int64_t t = wave_global;;) {   // (leaving the loop from the middle keeps the staging registers free of merges)
         bool done = false;
#pragma unroll
         for (int d = 0; d < DEPTH; ++d) {
            if (done || t >= n_tiles) {
               done = true;
               continue;
            }
            do_tile(stage[d], live[d], t, t + DEPTH * wave_stride);
            t += wave_stride;
         }
         if (done) break;
      }
   }
   if constexpr (DEFER) flush_queue();
   STAMP(6);
   STAMP_FLUSH;
   // one plain store per wave (not an atomic per tile: 16k same-address atomics cost ~0.2 ms); the value only gates the second pass
   if (MODE == 0 && any_deferred && lane == 0) n_deferred[0] = 1u;
   if (adapt && (wave_global & 255) == 0 && lane == 0) {   // the sample FX_ADAPT_CALLS describes: two atomics from every 256th wave (same-address atomics cost ~12 ns each)
      atomicAdd(&n_deferred[3], n_seen);
      if (n_def != 0u) atomicAdd(&n_deferred[2], n_def);
   }
}

// =========================================================================================================
// fx_match_fast: `.match.` on the tile kernel.  One forward pass of the anchored automaton over every byte of the row from
// M_start (the state after the optional leading NUL, api_internal_m.F90:280-289), verdict = the state's FINAL bit (accept at
// ci = n+2 or after the trailing NUL, :296-302), behind the reference's literal / prefix / suffix gate (forgex.F90:207-213,
// api_internal_m.F90:199-233) which is evaluated on the raw row bytes.  Same staging, schemes and UTF-8 second pass as the
// search kernel.
// =========================================================================================================
template <class F>
__device__ __forceinline__ void chain8_fwd(const F (&f)[8], uint32_t& state, const uint8_t* T) {
#pragma unroll
   for (int i = 0; i < 8; ++i) state = fxstep(f[i], state, T);
}
// only the first nv (wave-uniform) bytes of the group: a long row ends inside it
template <class F>
__device__ __forceinline__ void chain8_fwd_n(const F (&f)[8], uint32_t& state, const uint8_t* T, uint32_t nv) {
#pragma unroll
   for (int i = 0; i < 8; ++i)
      if ((uint32_t)i < nv) state = fxstep(f[i], state, T);
}

// A whole tile row (CH chunks of the wave's LDS tile) through the anchored automaton on the 8-state v_perm tables with THREE lookup buffers:
// every group's lookups are issued two chains ahead of their use (a chain of eight v_perm_b32 is 32 cycles; one chain ahead, the lookups come
// back late at two or three waves per SIMD).  Rolled trips of three chunks, then the CH % 3 chunks whose lookups are already in flight.
// The formulation of fx_match_tile (fx_one.hpp; profiles/r03_pipe3_ab.txt: 10 M x 256 B 0.476-0.488 -> 0.415-0.430 ms) for the segment loop
// of fx_match_fast (round 4: rows longer than 256 bytes).  `na` collects the OR of the row's words.
template <int CH>
__device__ __forceinline__ void fx_match_row_pipe3(const uint4* tile, const uint32_t lane, const uint2* __restrict__ tabA, uint32_t& st, uint32_t& na) {
   uint2 fa[8], fb[8], fc[8];
   auto cellc = [&](const uint32_t c) { return tile[tile_cell(lane, c < (uint32_t)CH ? c : (uint32_t)CH - 1u)]; };
   uint4 w0 = cellc(0), w1 = cellc(1), w2 = cellc(2);
   lookup8(fa, w0.x, w0.y, tabA);
   lookup8(fb, w0.z, w0.w, tabA);
   constexpr uint32_t TRIPS = (uint32_t)CH / 3u, REST = (uint32_t)CH % 3u;
#pragma unroll 1
   for (uint32_t c = 0; c < 3u * TRIPS; c += 3u) {
      na |= w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w | w2.x | w2.y | w2.z | w2.w;
      lookup8(fc, w1.x, w1.y, tabA);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fa, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fa, w1.z, w1.w, tabA);
      w0 = cellc(c + 3u);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fb, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fb, w2.x, w2.y, tabA);
      w1 = cellc(c + 4u);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fc, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fc, w2.z, w2.w, tabA);
      w2 = cellc(c + 5u);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fa, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fa, w0.x, w0.y, tabA);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fb, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fb, w0.z, w0.w, tabA);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fc, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
   }
   if constexpr (REST >= 1) {   // chunk 3 TRIPS: its lookups are in fa / fb, its words in w0
      na |= w0.x | w0.y | w0.z | w0.w;
      if constexpr (REST == 2) {
         na |= w1.x | w1.y | w1.z | w1.w;
         lookup8(fc, w1.x, w1.y, tabA);
      }
      chain8_fwd(fa, st, nullptr);
      if constexpr (REST == 2) lookup8(fa, w1.z, w1.w, tabA);
      chain8_fwd(fb, st, nullptr);
      if constexpr (REST == 2) {
         chain8_fwd(fc, st, nullptr);
         chain8_fwd(fa, st, nullptr);
      }
   }
}

// the same for the first 3 * trips chunks of a row whose chunk count is a run-time value (ragged rows of fx_match_tile): whole trips only,
// the caller's two-buffer loop takes the chunks behind them (the lookups issued ahead for a next trip are dropped)
template <int CH>
__device__ __forceinline__ void fx_match_trips_pipe3(const uint4* tile, const uint32_t lane, const uint2* __restrict__ tabA, uint32_t& st, uint32_t& na,
                                                     const uint32_t trips) {
   uint2 fa[8], fb[8], fc[8];
   auto cellc = [&](const uint32_t c) { return tile[tile_cell(lane, c < (uint32_t)CH ? c : (uint32_t)CH - 1u)]; };
   uint4 w0 = cellc(0), w1 = cellc(1), w2 = cellc(2);
   lookup8(fa, w0.x, w0.y, tabA);
   lookup8(fb, w0.z, w0.w, tabA);
#pragma unroll 1
   for (uint32_t c = 0; c < 3u * trips; c += 3u) {
      na |= w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w | w2.x | w2.y | w2.z | w2.w;
      lookup8(fc, w1.x, w1.y, tabA);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fa, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fa, w1.z, w1.w, tabA);
      w0 = cellc(c + 3u);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fb, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fb, w2.x, w2.y, tabA);
      w1 = cellc(c + 4u);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fc, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fc, w2.z, w2.w, tabA);
      w2 = cellc(c + 5u);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fa, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fa, w0.x, w0.y, tabA);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fb, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      lookup8(fb, w0.z, w0.w, tabA);
      __builtin_amdgcn_sched_barrier(0);
      chain8_fwd(fc, st, nullptr);
      __builtin_amdgcn_sched_barrier(0);
   }
}

// 2 = verdict is TRUE, 0 = verdict is FALSE, 1 = the automaton decides (fxrow::match_gate on the row's bytes in the LDS tile)
__device__ __forceinline__ uint32_t match_gate(const FxpHeader* h, const uint8_t* __restrict__ prog, const uint8_t* tb, uint32_t lane, uint32_t L) {
   auto row = [&](uint32_t j) -> uint32_t { return tb[(tile_cell(lane, j >> 4) << 4) + (j & 15u)]; };
   return fxrow::match_gate(h, prog, row, L);
}

// MODE as in fx_search_fast (BYTES modes: a row whose walk ends inside a character or in the INVALID state -- FINAL column 2 -- is
// left to the row-level fix-up)
// (round 4: LONG with CH = 8 is the HALF-row staging of 256-byte rows for the chain tables -- 8 KB of tile per wave, four waves per SIMD:
//  their one dependent LDS read per byte is latency-bound, `.match.` of a 23-state pattern over config-3 rows 0.94 -> 0.66 ms, profiles/r04_half_chain_ab.txt)
//  NOHALF: rows longer than 256 bytes in 128-byte segments, chain tables -- as fx_search_fast's)
template <int CH, int MODE, int SCH, bool RAGGED, bool LONG = false, bool NOHALF = false>
__global__ __launch_bounds__(256, (LONG && CH <= 8) ? 4 : 1) void fx_match_fast(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog,
                                                       FastParams fp, uint8_t* __restrict__ flags, uint32_t* __restrict__ n_deferred,
                                                       uint32_t class_map_in_lds, uint32_t Lr, uint32_t* __restrict__ clear_next,
                                                       uint32_t* __restrict__ worklist) {
   const uint32_t L = (RAGGED || LONG) ? Lr : 16u * CH;   // true row length; pads (symbol 255) behind it are the identity for A
   constexpr uint32_t SEGB = 16u * CH;                       // bytes of one LDS tile row = one segment of a long row
   const uint32_t S = LONG ? ((Lr + SEGB - 1u) / SEGB) : 1u;   // LONG: segments per row (the last one shorter when Lr % SEGB != 0, at any byte), left to right
   constexpr bool ragged = RAGGED;
   static_assert(!LONG || ((CH == 16 || CH == 8 || CH == 4) && !RAGGED && (MODE == 0 || MODE == 2 || MODE == 3)), "long rows: CH 16 (8 / 4: half rows), first-pass / byte-level modes");
   constexpr bool CHAIN = SCH == 1, WIDE = SCH == 2;
   constexpr bool LIST = MODE == 4, FIXUP = MODE == 1 || LIST, BYTES = MODE == 2 || MODE == 3, MARKED = MODE == 1 || MODE == 3;
   static_assert(!BYTES || (SCH != 0 && !RAGGED), "byte-level tables: chain or wide v_perm scheme, whole chunks");
   static_assert(!LIST || !RAGGED, "the worklist pass gathers whole-chunk rows");
   if ((MARKED || LIST) && n_deferred[fp.gate_word] == 0) return;
   if (!MARKED && !LIST && blockIdx.x == 0 && threadIdx.x == 0) {
      clear_next[0] = 0u;
      clear_next[1] = 0u;
      clear_next[2] = 0u;
      clear_next[3] = 0u;
   }
   using F = typename FxF<SCH>::type;
   __shared__ uint2 permA[SCH == 0 ? 256 : 1];
   __shared__ fx_nib wideA[WIDE ? 256 : 1];
   extern __shared__ __attribute__((aligned(16))) uint4 tiles[];
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   uint16_t* cmap = reinterpret_cast<uint16_t*>(tiles + 4 * 64 * CH);
   const uint32_t ta_bytes = BYTES ? h->byte_TA_bytes : h->chain_TA_bytes;
   const uint32_t chain_bytes = CHAIN ? ((512u + ta_bytes + 15u) & ~15u) : 0u;
   const uint8_t* TAp = reinterpret_cast<const uint8_t*>(cmap) + 512;
   if (CHAIN) {
      const uint16_t* g = reinterpret_cast<const uint16_t*>(prog + (BYTES ? h->off_byte_cls : h->off_chain_cls));
      const uint16_t* ga = reinterpret_cast<const uint16_t*>(prog + (BYTES ? h->off_byte_TA : h->off_chain_TA));
      const uint32_t na = ta_bytes / 2;
      fx_stage_chain(cmap, g, ga, ga, 0u, na);
   } else if (WIDE) {
      wideA[threadIdx.x] = reinterpret_cast<const fx_nib*>(prog + (BYTES ? h->off_bw16A : h->off_w16A))[threadIdx.x];
   } else {
      permA[threadIdx.x] = reinterpret_cast<const uint2*>(prog + h->off_fastA)[threadIdx.x];
   }
   __syncthreads();
   using TabT = typename std::conditional<CHAIN, uint16_t, typename std::conditional<WIDE, fx_nib, uint2>::type>::type;
   const TabT* tabA = CHAIN ? reinterpret_cast<const TabT*>(cmap) : (WIDE ? reinterpret_cast<const TabT*>(wideA) : reinterpret_cast<const TabT*>(permA));
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const bool utf8 = !BYTES && (FIXUP || fp.defer_tiles != 0);   // first pass: defer whole tiles that hold a byte >= 0x80
   const uint16_t* page_p = reinterpret_cast<const uint16_t*>(prog + h->off_cls_page);
   const uint16_t* pages_p = reinterpret_cast<const uint16_t*>(prog + h->off_cls_pages);
   if (FIXUP && class_map_in_lds) {
      uint16_t* l16 = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(tiles + 4 * 64 * CH) + chain_bytes);
      const uint32_t n16 = 1024u + h->n_pages * 64u;
      {   // 16-byte pieces (blob offsets and the LDS offset are multiples of 16): one round trip instead of one per 256 entries
         uint4* l4 = reinterpret_cast<uint4*>(l16);
         for (uint32_t i = threadIdx.x; i < n16 / 8u; i += 256u)
            l4[i] = i < 128u ? reinterpret_cast<const uint4*>(page_p)[i] : reinterpret_cast<const uint4*>(pages_p)[i - 128u];
      }
      __syncthreads();
      page_p = l16;
      pages_p = l16 + 1024;
   }
   const fxrow::ClassTables ct{page_p, pages_p, reinterpret_cast<const uint16_t*>(prog + h->off_bound_cls),
                               reinterpret_cast<const int32_t*>(prog + h->off_bounds), h->n_bounds};
   const uint32_t sym_ffff = 128u + h->cls_ffff;
   uint4* tile = tiles + wave * (64 * CH);
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   const bool whole = RAGGED && (Lr & 15u) == 0u;   // fewer WHOLE chunks than the instantiation: pad columns written once (see fx_search_fast)
   if (whole)
      for (uint32_t k = Lr >> 4; k < (uint32_t)CH; ++k) tile[tile_cell(lane, k)] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
   const int64_t n_tiles = (n + 63) >> 6;
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   bool any_deferred = false;
   auto tile_marked = [&](const int64_t t) -> bool {
      const int64_t rr = (t << 6) + lane;
      return __builtin_amdgcn_ballot_w64(rr < n && flags[rr] == FX_NEEDS_GENERAL) != 0;
   };
   const uint32_t list_count = LIST ? n_deferred[1] : 0u;
   uint4 stage[CH];
   bool live = true;   // the tile in `stage` is to be scanned (always, except in the marked-tile passes)
   if (!LIST) {
      live = MARKED ? tile_marked(wave_global) : true;
      if constexpr (LONG) PREFETCH_SEG_FWD(stage, wave_global, 0u, live);
      else PREFETCH_TILE(stage, wave_global, live);
   }
   for (int64_t t = wave_global; LIST ? (uint64_t)(t << 6) < list_count : t < n_tiles; t += wave_stride) {
      const int64_t row0 = t << 6;
      int64_t row = row0 + lane;   // the row this lane owns and whether it exists
      bool row_ok = row < n;
      bool defer_early = false;
      if (LIST) {
         // worklist pass: lane r gathers row worklist[64 t + r] straight into its own cells
         const uint32_t slot = (uint32_t)row0 + lane;
         row_ok = slot < list_count;
         row = row_ok ? (int64_t)worklist[slot] : 0;
         const uint4* src = reinterpret_cast<const uint4*>(rows + row * (int64_t)(16 * CH));
#pragma unroll
         for (int k = 0; k < CH; ++k) stage[k] = row_ok ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
         for (int k = 0; k < CH; ++k) tile[tile_cell(lane, k)] = stage[k];
      }
      uint32_t st = fp.A_init;   // = M_start
      uint32_t na = 0;
      uint32_t gate = 1u;
      bool skip = false;
      for (uint32_t seg = 0;; ++seg) {   // one pass unless LONG: the row's 256-byte segments, left to right
         if (!LIST) {
            const bool process = live;
            if (MODE == 0 && utf8 && seg == 0u) {
               const uint32_t smp = stage[0].x | stage[0].w | stage[CH / 2].y | stage[CH - 1].z;
               defer_early = __builtin_amdgcn_ballot_w64((smp & 0x80808080u) != 0) != 0;
            }
            if (process) STORE_TILE(stage);
            const bool last = !LONG || !process || defer_early || seg + 1u == S;   // nothing more of this tile is wanted
            if (last) live = MARKED ? tile_marked(t + wave_stride) : true;
            // the one reload site of the staging registers
            if constexpr (LONG) PREFETCH_SEG_FWD(stage, last ? t + wave_stride : t, last ? 0u : seg + 1u, last ? live : true);
            else PREFETCH_TILE(stage, t + wave_stride, live);
            if (!process) {
               skip = true;
               break;
            }
         }
         if (defer_early) {
            if (row_ok) flags[row] = FX_NEEDS_GENERAL;
            any_deferred = true;
            skip = true;
            break;
         }
         if (seg == 0u) {   // on the raw bytes, before any decode (long rows: straight from global memory)
            if (LONG) {
               const uint8_t* rp = rows + row * (int64_t)L;
               auto rowb = [&](uint32_t j) -> uint32_t { return rp[j]; };
               gate = row_ok ? fxrow::match_gate(h, prog, rowb, L) : 0u;
            } else {
               gate = match_gate(h, prog, tb, lane, L);
            }
         }
         if (FIXUP) {
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
         if (ragged && (!whole || FIXUP)) na |= pad_rows<CH>(tile, lane, Lr);
         // ---- left-to-right pass over the whole row, lookups of the next 8-byte group in flight during the chain ----
         // LONG: bytes of this segment (the row may end inside its last one, at any byte: the groups are walked over their valid bytes)
         const uint32_t seg_len = LONG ? (Lr - seg * SEGB < SEGB ? Lr - seg * SEGB : SEGB) : 16u * CH;
         // whole segments of long rows on the 8-state tables (round 4): three lookup buffers, as `.match.` over rows of up to 256 bytes has
         // had since round 3 (fx_match_tile)
         bool piped = false;
         if constexpr (FX_MATCH_LONG_P3 != 0 && LONG && SCH == 0 && CH == 16) {
            if (seg_len == 256u) {
               fx_match_row_pipe3<CH>(tile, lane, tabA, st, na);
               piped = true;
            }
         }
         F fa[8], fb[8];
         uint4 wk = tile[tile_cell(lane, 0)], wn = make_uint4(0, 0, 0, 0);
         if (CH >= 2) wn = tile[tile_cell(lane, 1)];
         if (!piped) lookup8(fa, wk.x, wk.y, tabA);
         const int nch = piped ? 0 : (LONG ? (int)((seg_len + 15u) >> 4) : CH);   // chunks of this segment
#pragma unroll 1   // rolled on purpose: fully unrolled, the state-independent lookups of ALL chunks get hoisted (512 VGPRs + scratch)
         for (int k = 0; k < nch; ++k) {
            const uint32_t nlo = !LONG ? 8u : (seg_len >= 16u * k + 8u ? 8u : seg_len - 16u * k);   // (>= 1: k < nch)
            const uint32_t nhi = !LONG ? 8u : (seg_len >= 16u * k + 16u ? 8u : (seg_len > 16u * k + 8u ? seg_len - (16u * k + 8u) : 0u));
            if (LONG) {
               if (nhi == 8u) na |= wk.x | wk.y | wk.z | wk.w;
               else {
                  const uint32_t w4[4] = {wk.x, wk.y, wk.z, wk.w};
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                     const uint32_t at = 16u * k + 4u * i;
                     if (at + 4u <= seg_len) na |= w4[i];
                     else if (at < seg_len) na |= w4[i] & ~(0xFFFFFFFFu << (8u * (seg_len - at)));
                  }
               }
            } else if (!ragged || (whole && !FIXUP && (uint32_t)k < (Lr >> 4))) na |= wk.x | wk.y | wk.z | wk.w;
            lookup8(fb, wk.z, wk.w, tabA);
            __builtin_amdgcn_sched_barrier(0);
            if (nlo == 8u) chain8_fwd(fa, st, TAp);
            else chain8_fwd_n(fa, st, TAp, nlo);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < CH) {
               wk = wn;
               lookup8(fa, wk.x, wk.y, tabA);
               if (k + 2 < CH) wn = tile[tile_cell(lane, k + 2)];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nhi == 8u) chain8_fwd(fb, st, TAp);
            else if (nhi != 0u) chain8_fwd_n(fb, st, TAp, nhi);
            __builtin_amdgcn_sched_barrier(0);
         }
         if (!LONG || seg + 1u == S) break;
      }
      if (skip) continue;
      uint32_t fin;
      if (CHAIN) fin = *reinterpret_cast<const uint16_t*>(TAp + st + 2u * ((BYTES ? h->byte_n_classes : h->n_classes) + 2u));   // FINAL column
      else if (WIDE) {
         const uint32_t* fm = BYTES ? h->bw16_finalM : h->w16_finalM;   // byte j = verdict of state j
         fin = (fm[(st >> 2) & 3u] >> ((st & 3u) * 8u)) & 3u;
      } else fin = __builtin_amdgcn_perm(h->fast_finalM[1], h->fast_finalM[0], st) & 1u;
      uint32_t flag = gate == 2u ? 1u : (gate == 0u ? 0u : (st != 0 && fin == 1u ? 1u : 0u));
      const bool row_hi = MODE == 0 && (na & 0x80808080u) != 0;
      const bool defer_tile = MODE == 0 && utf8 && __builtin_amdgcn_ballot_w64(row_hi) != 0;
      const bool exception = BYTES && gate == 1u && st != 0 && fin == 2u;   // inside a character / INVALID at the end of the row
      if (BYTES || MODE == 0) {   // exception rows are appended to the worklist of the decode pass / of the row-level fix-up
         const bool listed = (BYTES ? exception : (worklist != nullptr && row_hi && !utf8)) && row_ok;
         const uint64_t em = __builtin_amdgcn_ballot_w64(listed);
         if (em != 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&n_deferred[1], (uint32_t)__builtin_popcountll(em));
            base = __builtin_amdgcn_readfirstlane(base);
            if (listed) worklist[base + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = (uint32_t)row;
         }
      }
      if ((row_hi && !utf8) || defer_tile || exception) flag = FX_NEEDS_GENERAL;
      any_deferred = any_deferred || defer_tile;
      if (row_ok) flags[row] = (uint8_t)flag;
   }
   if (MODE == 0 && any_deferred && lane == 0) n_deferred[0] = 1u;
}

// MODE: 0 first pass, 1 decode second pass, 2 byte-level tables over all tiles, 3 byte-level tables over marked tiles
// n_deferred: this call's group of four words ([0] tiles deferred, [1] exception rows left, [2], [3] spare); the other call parity's group is 16 bytes away
template <int CH, int MODE, int SCH>
hipError_t launch_fast(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, int32_t* from,
                              int32_t* to, uint32_t* n_deferred, uint32_t class_map_bytes, uint32_t chain_bytes, uint32_t Lr, hipStream_t st, uint32_t* worklist, int64_t grid_tiles) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(n_deferred) ^ 16u);   // the other parity's group of four words (32-byte aligned block)
   const int64_t n_tiles = grid_tiles > 0 ? grid_tiles : (n + 63) >> 6;   // (worklist pass: the host only knows an upper bound)
   int64_t blocks = (n_tiles + 3) / 4;
   // grid-stride beyond the cap (guide §6 G11); a whole number of rounds of what is resident (two blocks per CU at 256-byte rows,
   // four with half-row staging), so that the last round fills the chip too.  Half-row kernel: TWELVE rounds (12288 blocks, three tiles
   // per wave at 10 M rows) -- measured in one allocation, five interleaved repetitions each (profiles/r03_half4_ab.txt): 3 rounds
   // 0.506 ms, 8: 0.478, 12: 0.474, 16: 0.482: the hardware's block scheduler balances the tail better than a long static stride
   // (the gated passes -- marked tiles, worklist -- usually find nothing to do: a grid of what is resident, so that an empty pass is
   //  one round of blocks that leave at once)
   const int64_t cap = MODE == 4 ? 256 : ((MODE == 1 || MODE == 3) ? 256 * 2 : ((CH <= 8 && Lr > 16u * CH) ? 256 * (FX_HALF4 != 0 && FX_DEFER_LONG == 0 ? 4 * 12 : 3 * (FX_DEFER_LONG != 0 ? FX_HALF_WAVES : 3)) : 256 * 8));
   int64_t gcap = cap;
   if (CH <= 8 && Lr > 16u * CH && MODE == 0 && FX_HALF4 != 0 && FX_DEFER_LONG == 0) {
      // what a block costs (table staging, first tile without overlap) against what a finer grid gains at the tail: a round of about
      // 50 us (11 rounds at 10 M rows; measured: 3 rounds 0.506 ms, 8: 0.478, 12: 0.474, 16: 0.482); FXAMD_HALF_ROUNDS: experiment hook
      int64_t rounds = fx_env().half_rounds;
      if (rounds <= 0) {   // one round per 225 MB of rows: a round of about 50 us (the same rule as the one-launch kernel's, fx_one.hpp)
         rounds = (n * (int64_t)Lr) / ((int64_t)225 << 20);
         if (rounds < 3) rounds = 3;
         if (rounds > 64) rounds = 64;
      }
      gcap = (int64_t)256 * 4 * rounds;
   }
   if (blocks > gcap) blocks = gcap;
   // decode passes: the BMP class map rides behind the tiles when it fits
   const uint32_t map_lds = ((MODE == 1 || MODE == 4) && class_map_bytes <= 24u * 1024u) ? class_map_bytes : 0u;
   const size_t lds = (size_t)4 * 64 * (CH + 1) * 16 + chain_bytes + map_lds;   // + the end-of-row chunk column
   const bool ragged = Lr != 16u * CH;
   const bool spans = from && to;
   if (Lr > 16u * CH) {   // long rows: segment-walking instantiation (CH = 16: any length; CH = 8, first pass with the 8-state tables:
                          // half-row staging of 256-byte rows), first-pass / byte-level modes only
      if constexpr (CH == 8 && SCH == 1 && (MODE == 0 || MODE == 2)) {
         if (Lr != 256u) {   // rows longer than 256 bytes on the chain tables: 128-byte segments, finished from global memory (NOHALF)
            const size_t lds8 = (size_t)4 * 64 * CH * 16 + chain_bytes + map_lds + 64;
            if (lds8 > 64 * 1024) {
               const hipError_t e = hipFuncSetAttribute(spans ? reinterpret_cast<const void*>(&fx_search_fast<CH, true, MODE, SCH, false, true, true>)
                                                              : reinterpret_cast<const void*>(&fx_search_fast<CH, false, MODE, SCH, false, true, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
               if (e != hipSuccess) return e;
            }
            if (spans) hipLaunchKernelGGL((fx_search_fast<CH, true, MODE, SCH, false, true, true>), dim3((unsigned)blocks), dim3(256), lds8, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
            else hipLaunchKernelGGL((fx_search_fast<CH, false, MODE, SCH, false, true, true>), dim3((unsigned)blocks), dim3(256), lds8, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
            return hipGetLastError();
         }
      }
      if constexpr ((CH == 16 && (MODE == 0 || MODE == 2 || MODE == 3)) || (CH == 8 && MODE == 0) || (CH == 4 && MODE == 0 && SCH == 1)) {
         constexpr int CHN = SCH;
         if (CH <= 8 && Lr != 32u * CH) return hipErrorInvalidValue;   // (half rows: rows of twice the tile's bytes only)
         const size_t lds = (size_t)4 * 64 * (spans ? fx_tile_cols<CH, true, true>() : fx_tile_cols<CH, false, true>()) * 16 + chain_bytes + map_lds +
                            ((FX_HALF4 != 0 && CH <= 8 && spans) ? 64 : 0);   // (+ the four shared end-of-row cells)
         const void* fn = spans ? reinterpret_cast<const void*>(&fx_search_fast<CH, true, MODE, CHN, false, true>)
                                : reinterpret_cast<const void*>(&fx_search_fast<CH, false, MODE, CHN, false, true>);
         if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
         }
         if constexpr (CH == 8 && MODE == 0 && SCH == 0 && FX_HALF4 != 0 && FX_DEFER_LONG == 0) {
            if (spans && fp.latch != 0u) {   // (the latched format of R: FXP_F_R_LATCH programs; same LDS, same grid)
               hipLaunchKernelGGL((fx_search_fast<CH, true, MODE, CHN, false, true, false, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
               return hipGetLastError();
            }
         }
         if (fp.latch != 0u) return hipErrorInvalidValue;   // (never dispatched: only the half-row first pass with spans reads the latched format)
         if (spans) hipLaunchKernelGGL((fx_search_fast<CH, true, MODE, CHN, false, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
         else hipLaunchKernelGGL((fx_search_fast<CH, false, MODE, CHN, false, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
         return hipGetLastError();
      } else {
         return hipErrorInvalidValue;   // (never dispatched)
      }
   }
   if constexpr (MODE >= 2) {   // whole-chunk rows only: byte-level tables (2, 3: chain scheme) and the worklist decode pass (4)
      constexpr int CHN = SCH;
      if (ragged) return hipErrorInvalidValue;   // (never dispatched)
      const void* fn = spans ? reinterpret_cast<const void*>(&fx_search_fast<CH, true, MODE, CHN, false>)
                             : reinterpret_cast<const void*>(&fx_search_fast<CH, false, MODE, CHN, false>);
      if (lds > 64 * 1024) {
         hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
         if (e != hipSuccess) return e;
      }
      if (spans) hipLaunchKernelGGL((fx_search_fast<CH, true, MODE, CHN, false>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
      else hipLaunchKernelGGL((fx_search_fast<CH, false, MODE, CHN, false>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
      return hipGetLastError();
   } else {
      // (ragged rows run on the power-of-two instantiations, as in the one-launch kernel -- fxamd.hip, chunks_of: round 6 retired the ragged variants of 3, 6
      //  and 12 chunks, 100-odd kernels that answered rows of 33..47 / 65..95 / 129..191 bytes 2-5 % faster than the next power of two does)
      constexpr bool RAG_OK = (CH & (CH - 1)) == 0;
      if (ragged && !RAG_OK) return hipErrorInvalidValue;
      const void* fn = spans ? reinterpret_cast<const void*>(&fx_search_fast<CH, true, MODE, SCH, false>) : reinterpret_cast<const void*>(&fx_search_fast<CH, false, MODE, SCH, false>);
      if constexpr (RAG_OK) {
         if (ragged) fn = spans ? reinterpret_cast<const void*>(&fx_search_fast<CH, true, MODE, SCH, true>) : reinterpret_cast<const void*>(&fx_search_fast<CH, false, MODE, SCH, true>);
      }
      if (lds > 64 * 1024) {   // beyond the default dynamic-LDS window: raise the kernel's limit (idempotent)
         hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
         if (e != hipSuccess) return e;
      }
      if (ragged) {
         if constexpr (RAG_OK) {
            if (spans) hipLaunchKernelGGL((fx_search_fast<CH, true, MODE, SCH, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
            else hipLaunchKernelGGL((fx_search_fast<CH, false, MODE, SCH, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
         }
      } else {
         if (spans) hipLaunchKernelGGL((fx_search_fast<CH, true, MODE, SCH, false>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
         else hipLaunchKernelGGL((fx_search_fast<CH, false, MODE, SCH, false>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, from, to, n_deferred, map_lds, Lr, clear_next, worklist);
      }
      return hipGetLastError();
   }
}

template <int CH, int MODE, int SCH>
hipError_t launch_match(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, uint32_t* n_deferred,
                               uint32_t class_map_bytes, uint32_t chain_bytes, uint32_t Lr, hipStream_t st, uint32_t* worklist, int64_t grid_tiles) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(n_deferred) ^ 16u);
   const int64_t n_tiles = grid_tiles > 0 ? grid_tiles : (n + 63) >> 6;
   int64_t blocks = (n_tiles + 3) / 4;
   if (blocks > (MODE == 4 ? 256 : 256 * 8)) blocks = MODE == 4 ? 256 : 256 * 8;
   const uint32_t map_lds = ((MODE == 1 || MODE == 4) && class_map_bytes <= 24u * 1024u) ? class_map_bytes : 0u;
   const size_t lds = (size_t)4 * 64 * CH * 16 + chain_bytes + map_lds;
   const bool ragged = Lr != 16u * CH;
   if (Lr > 16u * CH) {   // long rows (CH = 16), or 256-byte rows staged as half rows (CH = 8: first pass on the chain tables)
      if constexpr (CH == 8 && SCH == 1 && (MODE == 0 || MODE == 2)) {
         if (Lr != 256u) {   // rows longer than 256 bytes on the chain tables: 128-byte segments (NOHALF: the loader's tail fix for short last segments)
            if (lds > 64 * 1024) {
               const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_match_fast<CH, MODE, SCH, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
               if (e != hipSuccess) return e;
            }
            int64_t rounds = (n * (int64_t)Lr) / ((int64_t)225 << 20);
            rounds = rounds < 3 ? 3 : (rounds > 64 ? 64 : rounds);
            blocks = (n_tiles + 3) / 4;
            if (blocks > 256 * 4 * rounds) blocks = 256 * 4 * rounds;
            hipLaunchKernelGGL((fx_match_fast<CH, MODE, SCH, false, true, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, n_deferred, map_lds, Lr, clear_next, worklist);
            return hipGetLastError();
         }
      }
      if constexpr ((CH == 16 && (MODE == 0 || MODE == 2 || MODE == 3)) || (CH == 8 && MODE == 0 && SCH == 1)) {
         constexpr int CHN = SCH;
         if (CH == 8 && Lr != 256u) return hipErrorInvalidValue;   // (half rows: 256-byte rows only -- 128-byte rows in 64-byte halves gained 1.6 %: not dispatched)
         const void* fn = reinterpret_cast<const void*>(&fx_match_fast<CH, MODE, CHN, false, true>);
         if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
         }
         if (CH <= 8) {   // whole rounds of the four resident blocks per CU, as the half-row search kernel's grid
            int64_t rounds = (n * (int64_t)Lr) / ((int64_t)225 << 20);
            rounds = rounds < 3 ? 3 : (rounds > 64 ? 64 : rounds);
            blocks = (n_tiles + 3) / 4;
            if (blocks > 256 * 4 * rounds) blocks = 256 * 4 * rounds;
         }
         hipLaunchKernelGGL((fx_match_fast<CH, MODE, CHN, false, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, n_deferred, map_lds, Lr, clear_next, worklist);
         return hipGetLastError();
      } else {
         return hipErrorInvalidValue;
      }
   }
   if constexpr (MODE >= 2) {
      constexpr int CHN = SCH;
      if (ragged) return hipErrorInvalidValue;
      if (lds > 64 * 1024) {
         hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_match_fast<CH, MODE, CHN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
         if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL((fx_match_fast<CH, MODE, CHN, false>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, n_deferred, map_lds, Lr, clear_next, worklist);
      return hipGetLastError();
   } else {
      constexpr bool RAG_OK = (CH & (CH - 1)) == 0;   // (ragged rows: the power-of-two instantiations only, see launch_fast)
      if (ragged && !RAG_OK) return hipErrorInvalidValue;
      const void* fn = reinterpret_cast<const void*>(&fx_match_fast<CH, MODE, SCH, false>);
      if constexpr (RAG_OK) {
         if (ragged) fn = reinterpret_cast<const void*>(&fx_match_fast<CH, MODE, SCH, true>);
      }
      if (lds > 64 * 1024) {
         hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
         if (e != hipSuccess) return e;
      }
      if (ragged) {
         if constexpr (RAG_OK) hipLaunchKernelGGL((fx_match_fast<CH, MODE, SCH, true>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, n_deferred, map_lds, Lr, clear_next, worklist);
      } else hipLaunchKernelGGL((fx_match_fast<CH, MODE, SCH, false>), dim3((unsigned)blocks), dim3(256), lds, st, rows, n, d_blob, fp, flags, n_deferred, map_lds, Lr, clear_next, worklist);
      return hipGetLastError();
   }
}


// every (CH, MODE, SCH) the dispatch code of fxamd.hip can ask for
#define FX_TILE_COMBOS(X, CH) \
   X(CH, 0, 0) X(CH, 0, 1) X(CH, 0, 2) X(CH, 1, 0) X(CH, 1, 1) X(CH, 1, 2) X(CH, 4, 0) X(CH, 4, 1) X(CH, 4, 2) X(CH, 2, 1) X(CH, 2, 2) X(CH, 3, 1) X(CH, 3, 2)
#define FX_TILE_ALL(X) \
   FX_TILE_COMBOS(X, 1) FX_TILE_COMBOS(X, 2) FX_TILE_COMBOS(X, 4) FX_TILE_COMBOS(X, 8) FX_TILE_COMBOS(X, 12) FX_TILE_COMBOS(X, 16)
#define FX_TILE_SIG_FAST \
   (const uint8_t*, int64_t, const uint8_t*, FastParams, uint8_t*, int32_t*, int32_t*, uint32_t*, uint32_t, uint32_t, uint32_t, hipStream_t, uint32_t*, int64_t)
#define FX_TILE_SIG_MATCH (const uint8_t*, int64_t, const uint8_t*, FastParams, uint8_t*, uint32_t*, uint32_t, uint32_t, uint32_t, hipStream_t, uint32_t*, int64_t)
