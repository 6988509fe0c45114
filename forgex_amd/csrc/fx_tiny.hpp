// fx_match_tiny: `.match.` over TINY rows -- 2 to 32 bytes (round 4; BASELINE config 1's shape: `\d{3}-\d{4}` over 8-byte rows).
//
// A tile of the one-launch kernel is 64 rows, whatever their length: at 8 bytes per row a wave stages 512 bytes per trip and pays the trip's
// fixed work -- staging, tail patch, gate, verdict, loop -- for 8 bytes per lane (profiles/r04b_match_cfg1x_summary.txt: 13.5 vector and
// 14.3 scalar instructions per input byte, 0.22 of the HBM peak).  Here a lane owns a SPAN of 64 consecutive bytes = 64 / L whole rows: the
// tile is the aligned 64-byte-row tile (4 KB of contiguous bytes per wave and trip, the coalesced loader and the swizzled store of the
// other kernels), and the lane walks its rows one after the other out of its own four cells -- per row a fresh start state (M_start: the
// state after the optional leading NUL, api_internal_m.F90:280-289), L steps, the FINAL verdict (accept at ci = n + 2 or after the trailing
// NUL, :296-302) -- and stores its 64 / L verdict bytes with ONE store.  Class-level tables (8-state v_perm or 16-state nibbles); rows with
// a byte >= 0x80 are listed for the row-level fix-up of the multi-pass pipelines (fx_fixup_list: the general row procedure), as are all
// rows of programs with a literal gate that the general procedure evaluates (forgex.F90:207-213, api_internal_m.F90:199-233: evaluated here
// per row when the program has one).
#pragma once
#include "fx_tile.hpp"

// ---- what the two kernels share -------------------------------------------------------------------------------------------------------------
// Rows of ANY length 2 <= L <= 32 (round 4, second version: the first took 4 / 8 / 16 / 32 only).  A lane owns RPL = 64 / L whole rows = a
// span of LS = RPL * L <= 64 consecutive bytes; a wave's trip covers 64 spans = 64 * LS contiguous bytes.  LS == 64: the aligned 64-byte-row
// tile (coalesced 16-byte pieces, store_tile).  Else the spans are ragged 64-byte "rows": four lanes share a span and read its pieces at the
// span stride -- the loader of the pad-free ragged scheme (fx_tile.hpp, "Ragged rows, round 4") with an explicit byte extent, because the
// batch may end inside the last span.  A lane then holds its span in 16 registers and cuts row j out of them at the compile-time offset j * L.
template <int L>
struct FxTiny {
   static_assert(L >= 2 && L <= 32, "tiny rows");
   static constexpr int RPL = 64 / L;          // rows per lane
   static constexpr int LS = RPL * L;          // bytes per lane span
   static constexpr int NCH = (LS + 15) / 16;  // chunks that hold span bytes
   static constexpr int NW = (L + 3) / 4;      // dwords per row
};
// the tile loads of trip t: `total` = bytes of the batch
template <int L>
__device__ __forceinline__ void fx_tiny_load(uint4 (&stage)[4], const uint8_t* __restrict__ rows, const int64_t total, const int64_t t, const uint32_t lane) {
   using T = FxTiny<L>;
   const int64_t off0 = t * (int64_t)(64 * T::LS);
   const int64_t left = total - off0;
   // (a dword is dropped whole when it straddles the extent, and with rows that are not a multiple of 4 bytes a span's last dword may: a
   //  tile that is followed by another gets 3 more bytes of extent; the batch's LAST tile gets its exact extent and the straddling dword
   //  is rebuilt from byte loads -- nothing behind the caller's last byte is read)
   const uint32_t tile_bytes = left <= 0 ? 0u : (uint32_t)(left >= 64 * T::LS ? 64 * T::LS : left);
   const int64_t room = left > 64 * T::LS ? left - 64 * T::LS : 0;   // bytes of the batch behind this tile
   const bool last_tile = room < 3;                                  // wave-uniform
   const uint32_t valid = left <= 0 ? 0u : (L % 4 == 0 ? tile_bytes : (last_tile ? tile_bytes + (uint32_t)room : tile_bytes + 3u));
   const uint64_t base = reinterpret_cast<uint64_t>(rows) + (uint64_t)(left > 0 ? off0 : 0);
   const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
   const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                         __builtin_amdgcn_readfirstlane(valid), 0x00020000);
   if constexpr (T::LS == 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
         const fx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u + (uint32_t)q * 1024u, 0, FX_LOAD_AUX);
         stage[q] = make_uint4(v.x, v.y, v.z, v.w);
      }
   } else {
      const uint32_t r0 = lane >> 2, k = lane & 3u;   // four lanes per span, sixteen spans per instruction
      const uint32_t voff = k < (uint32_t)T::NCH ? r0 * (uint32_t)T::LS + 16u * k : 0x7FFFFFF0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
         const fx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (uint32_t)(q * 16 * T::LS), FX_LOAD_AUX);
         stage[q] = make_uint4(v.x, v.y, v.z, v.w);
      }
   }
}
// ... into LDS; `t`: the trip the registers hold.  The batch's last tile: the one text dword its exact extent cut off is rebuilt in LDS (fx_last_dword_bytes,
// fx_tile.hpp; round 5 patched the staging registers in the loader) -- units are the lane spans of LS bytes: span u = cells of row u, chunk m / 4
template <int L>
__device__ __forceinline__ void fx_tiny_store(const uint4 (&stage)[4], uint4* tile, const uint32_t lane, const uint8_t* __restrict__ rows, const int64_t total, const int64_t t) {
   using T = FxTiny<L>;
   if (T::LS == 64 || (lane & 3u) < (uint32_t)T::NCH) store_tile<4>(stage, tile, lane);
   if constexpr (L % 4 != 0) {
      const int64_t off0 = t * (int64_t)(64 * T::LS);
      FxLastDword d;
      if (fx_last_dword_bytes(d, rows + off0, total - off0, 64u * (uint32_t)T::LS, (uint32_t)T::LS)) {   // wave-uniform
         if (lane == 0) reinterpret_cast<uint32_t*>(tile)[(tile_cell(d.row, d.m >> 2) << 2) + (d.m & 3u)] = d.word;
      }
   }
}
// the lane's span as 16 dwords (+ one of slack for the byte shifts), and row j of it as NW dwords (bytes behind the row in the last one are
// whatever follows: the walks look at the row's L bytes only)
template <int L>
__device__ __forceinline__ void fx_tiny_span(uint32_t (&d)[17], const uint4* tile, const uint32_t lane) {
#pragma unroll
   for (int c = 0; c < 4; ++c) {
      const uint4 v = c < FxTiny<L>::NCH ? tile[tile_cell(lane, (uint32_t)c)] : make_uint4(0, 0, 0, 0);
      d[4 * c] = v.x;
      d[4 * c + 1] = v.y;
      d[4 * c + 2] = v.z;
      d[4 * c + 3] = v.w;
   }
   d[16] = 0;
}
template <int L, int J>
__device__ __forceinline__ void fx_tiny_row(uint32_t (&w)[FxTiny<L>::NW], const uint32_t (&d)[17]) {
   constexpr int off = J * L, k0 = off >> 2, sh = off & 3;
#pragma unroll
   for (int i = 0; i < FxTiny<L>::NW; ++i) w[i] = sh == 0 ? d[k0 + i] : __builtin_amdgcn_alignbyte(d[k0 + i + 1], d[k0 + i], (uint32_t)sh);
}
// OR of the row's L bytes
template <int L>
__device__ __forceinline__ uint32_t fx_tiny_or(const uint32_t (&w)[FxTiny<L>::NW]) {
   uint32_t na = 0;
#pragma unroll
   for (int i = 0; i < FxTiny<L>::NW; ++i) na |= (4 * i + 4 <= L) ? w[i] : (w[i] & (0xFFFFFFFFu >> (8 * (4 * i + 4 - L))));
   return na;
}
// the lane's RPL verdict bytes (consecutive rows): one store where RPL is a power of two and all rows exist, bytes otherwise
template <int L>
__device__ __forceinline__ void fx_tiny_emit(uint8_t* __restrict__ flags, const int64_t row_first, const int64_t n, const uint32_t (&out)[(FxTiny<L>::RPL + 3) / 4]) {
   constexpr int RPL = FxTiny<L>::RPL;
   if constexpr (RPL == 2 || RPL == 4 || RPL == 8 || RPL == 16) {
      // (one wide store only where it is aligned: result set i of a many-pattern call starts at flags + i * n -- wave-uniform)
      if (row_first + RPL <= n && (reinterpret_cast<uintptr_t>(flags) & (uintptr_t)(RPL - 1)) == 0) {
         uint8_t* dst = flags + row_first;
         if constexpr (RPL == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)out[0];
         else if constexpr (RPL == 4) *reinterpret_cast<uint32_t*>(dst) = out[0];
         else if constexpr (RPL == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(out[0], out[1]);
         else *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
         return;
      }
   }
#pragma unroll
   for (int j = 0; j < RPL; ++j)
      if (row_first + j < n) flags[row_first + j] = (uint8_t)(out[j / 4] >> (8 * (j & 3)));
}
// rows that go to the row-level fix-up: one atomic per wave and row slot that has any
__device__ __forceinline__ void fx_tiny_list(const bool listed, const int64_t row, const uint32_t lane, uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ worklist) {
   const uint64_t em = __builtin_amdgcn_ballot_w64(listed);
   if (em != 0) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&n_deferred[1], (uint32_t)__builtin_popcountll(em));
      base = __builtin_amdgcn_readfirstlane(base);
      if (listed) worklist[base + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = (uint32_t)row;
   }
}

template <int L, int SCH, int J>
struct FxTinyMatchRows {
   template <class F, class Gate>
   static __device__ __forceinline__ void run(const uint32_t (&d)[17], const F* __restrict__ tabA, const FastParams& fp, const uint32_t fm0, const uint32_t fm1, const uint32_t fm2, const uint32_t fm3, const bool gated,
                                              const Gate& gate_of, const int64_t row_first, const int64_t n, const uint32_t lane, uint32_t* n_deferred,
                                              uint32_t* worklist, uint32_t (&out)[(FxTiny<L>::RPL + 3) / 4]) {
      if constexpr (J < FxTiny<L>::RPL) {
         uint32_t w[FxTiny<L>::NW];
         fx_tiny_row<L, J>(w, d);
         const uint32_t na = fx_tiny_or<L>(w);
         F f[L];
#pragma unroll
         for (int i = 0; i < L; ++i) f[i] = tabA[(w[i >> 2] >> (8 * (i & 3))) & 0xFFu];
         uint32_t st = fp.A_init;   // = M_start
#pragma unroll
         for (int i = 0; i < L; ++i) st = fxstep(f[i], st, nullptr);
         uint32_t fin;
         if (SCH == 2) {
            const uint32_t fw = (st & 8u) ? ((st & 4u) ? fm3 : fm2) : ((st & 4u) ? fm1 : fm0);
            fin = (fw >> ((st & 3u) * 8u)) & 3u;
         } else fin = __builtin_amdgcn_perm(fm1, fm0, st) & 1u;
         uint32_t flag = (st != 0u && fin == 1u) ? 1u : 0u;
         if (gated) {   // (wave-uniform: the program has a literal / prefix / suffix gate)
            const uint32_t gate = gate_of(J * L);
            flag = gate == 2u ? 1u : (gate == 0u ? 0u : flag);
         }
         const int64_t row = row_first + J;
         const bool listed = row < n && (na & 0x80808080u) != 0u;   // a byte >= 0x80: UTF-8 decode by the general procedure
         fx_tiny_list(listed, row, lane, n_deferred, worklist);
         if (listed) flag = FX_NEEDS_GENERAL;
         out[J / 4] |= flag << (8 * (J & 3));
         FxTinyMatchRows<L, SCH, J + 1>::run(d, tabA, fp, fm0, fm1, fm2, fm3, gated, gate_of, row_first, n, lane, n_deferred, worklist, out);
      }
   }
};

template <int L, int SCH>
__global__ __launch_bounds__(256) void fx_match_tiny(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog, FastParams fp,
                                                      uint8_t* __restrict__ flags, uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ clear_next,
                                                      uint32_t* __restrict__ worklist) {
   static_assert(SCH == 0 || SCH == 2, "class-level v_perm or nibble tables");
   using T = FxTiny<L>;
   using F = typename FxF<SCH>::type;
   if (blockIdx.x == 0 && threadIdx.x == 0) {   // (a first pass of the multi-pass pipelines: it zeroes the next call's counter group)
      clear_next[0] = 0u;
      clear_next[1] = 0u;
      clear_next[2] = 0u;
      clear_next[3] = 0u;
   }
   __shared__ F tabA_s[256];
   __shared__ __attribute__((aligned(16))) uint4 tiles[4 * 64 * 4];
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   {
      const uint2 e = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16A : h->off_fastA))[threadIdx.x];
      reinterpret_cast<uint2*>(tabA_s)[threadIdx.x] = e;
   }
   __syncthreads();
   const F* tabA = tabA_s;
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint4* tile = tiles + wave * 256;
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   const int64_t total = n * (int64_t)L;
   const int64_t n_tiles = (total + 64 * T::LS - 1) / (64 * T::LS);
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   const bool gated = (h->len_prefix | h->len_suffix | h->len_all) != 0u;
   uint32_t fm0 = 0, fm1 = 0, fm2 = 0, fm3 = 0;   // FINAL verdict of a state: byte q of {fm1, fm0} (v_perm) / of fm0..fm3 (nibble tables)
   if (SCH == 2) {
      fm0 = h->w16_finalM[0];
      fm1 = h->w16_finalM[1];
      fm2 = h->w16_finalM[2];
      fm3 = h->w16_finalM[3];
   } else {
      fm0 = h->fast_finalM[0];
      fm1 = h->fast_finalM[1];
   }
   uint4 stage[4];
   fx_tiny_load<L>(stage, rows, total, wave_global, lane);
   for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
      fx_tiny_store<L>(stage, tile, lane, rows, total, t);
      fx_tiny_load<L>(stage, rows, total, t + wave_stride, lane);   // the ONE reload site of the staging registers
      uint32_t d[17];
      fx_tiny_span<L>(d, tile, lane);
      const int64_t row_first = ((t << 6) + lane) * T::RPL;
      uint32_t out[(T::RPL + 3) / 4] = {0};
      auto gate_of = [&](const int off) -> uint32_t {
         auto rowb = [&](uint32_t k) -> uint32_t { return tb[(tile_cell(lane, ((uint32_t)off + k) >> 4) << 4) + (((uint32_t)off + k) & 15u)]; };
         return fxrow::match_gate(h, prog, rowb, (uint32_t)L);
      };
      FxTinyMatchRows<L, SCH, 0>::run(d, tabA, fp, fm0, fm1, fm2, fm3, gated, gate_of, row_first, n, lane, n_deferred, worklist, out);
      fx_tiny_emit<L>(flags, row_first, n, out);
   }
}

// fx_search_tiny: the `.in.` VERDICT (flags only: what the reference's operator returns, forgex.F90:74-160) over the same tiny rows.  Per row
// the reverse automaton walks the L bytes from the row's last byte (start state: after the trailing NUL) -- a hit anywhere is a start
// inside the text, which always yields a span (api_internal_m.F90:140-148), so the verdict is TRUE -- then the leading NUL: a start THERE is
// the leftmost one, and the verdict is the forward walk's (max_match > 2: an accept after at least one more symbol than the NUL), walked
// here with the anchored tables whenever a lane of the wave has such a start.  Rows with a byte >= 0x80 and rows that end in the overlap
// state of a bordered prefix literal (FXP_F_OVERLAP_SINK) go to the row-level fix-up.
template <int L, int SCH, int J>
struct FxTinySearchRows {
   template <class F>
   static __device__ __forceinline__ void run(const uint32_t (&d)[17], const F* __restrict__ tabR, const F* __restrict__ tabA, const FastParams& fp, const F fz,
                                              const int64_t row_first, const int64_t n, const uint32_t lane, uint32_t* n_deferred, uint32_t* worklist,
                                              uint32_t (&out)[(FxTiny<L>::RPL + 3) / 4]) {
      if constexpr (J < FxTiny<L>::RPL) {
         uint32_t w[FxTiny<L>::NW];
         fx_tiny_row<L, J>(w, d);
         const uint32_t na = fx_tiny_or<L>(w);
         F f[L];
#pragma unroll
         for (int i = 0; i < L; ++i) f[i] = tabR[(w[i >> 2] >> (8 * (i & 3))) & 0xFFu];
         uint32_t st = fp.R_start, mx = 0;
#pragma unroll
         for (int i = L - 1; i >= 0; --i) {
            st = fxstep(f[i], st, nullptr);
            mx = max(mx, st);
         }
         const bool hit = mx >= fp.hit_min;                 // a start inside the text
         const uint32_t sn = fxstep(fz, st, nullptr);
         const bool s_nul = sn >= fp.hit_min;               // a start at the leading NUL
         bool verdict = hit;
         if (__builtin_amdgcn_ballot_w64(s_nul) != 0) {   // (wave-uniform; `^`-anchored patterns) forward from the leading NUL
            uint32_t cur = s_nul ? fp.A_init : 0u;
            const F fza = tabA[0];
            cur = fxstep(fza, cur, nullptr);
            bool acc = false;
#pragma unroll
            for (int i = 0; i < L; ++i) {
               const F fa = tabA[(w[i >> 2] >> (8 * (i & 3))) & 0xFFu];
               cur = fxstep(fa, cur, nullptr);
               acc = acc || cur >= fp.acc_min;
            }
            cur = fxstep(fza, cur, nullptr);   // the trailing NUL
            acc = acc || cur >= fp.acc_min;
            verdict = s_nul ? acc : hit;
         }
         const int64_t row = row_first + J;
         const bool listed = row < n && ((na & 0x80808080u) != 0u || (fp.inv_on != 0u && sn == fp.inv));
         fx_tiny_list(listed, row, lane, n_deferred, worklist);
         const uint32_t flag = listed ? (uint32_t)FX_NEEDS_GENERAL : (verdict ? 1u : 0u);
         out[J / 4] |= flag << (8 * (J & 3));
         FxTinySearchRows<L, SCH, J + 1>::run(d, tabR, tabA, fp, fz, row_first, n, lane, n_deferred, worklist, out);
      }
   }
};

template <int L, int SCH>
__global__ __launch_bounds__(256) void fx_search_tiny(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog, FastParams fp,
                                                       uint8_t* __restrict__ flags, uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ clear_next,
                                                       uint32_t* __restrict__ worklist) {
   static_assert(SCH == 0 || SCH == 2, "class-level v_perm or nibble tables");
   using T = FxTiny<L>;
   using F = typename FxF<SCH>::type;
   if (blockIdx.x == 0 && threadIdx.x == 0) {
      clear_next[0] = 0u;
      clear_next[1] = 0u;
      clear_next[2] = 0u;
      clear_next[3] = 0u;
   }
   __shared__ F tabR_s[256];
   __shared__ F tabA_s[256];
   __shared__ __attribute__((aligned(16))) uint4 tiles[4 * 64 * 4];
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(prog);
   {
      const uint2 e = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16R : h->off_fastR))[threadIdx.x];
      reinterpret_cast<uint2*>(tabR_s)[threadIdx.x] = e;
      const uint2 ea = reinterpret_cast<const uint2*>(prog + (SCH == 2 ? h->off_w16A : h->off_fastA))[threadIdx.x];
      reinterpret_cast<uint2*>(tabA_s)[threadIdx.x] = ea;
   }
   __syncthreads();
   const F* tabR = tabR_s;
   const F* tabA = tabA_s;
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint4* tile = tiles + wave * 256;
   const int64_t total = n * (int64_t)L;
   const int64_t n_tiles = (total + 64 * T::LS - 1) / (64 * T::LS);
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   uint4 stage[4];
   fx_tiny_load<L>(stage, rows, total, wave_global, lane);
   const F fz = tabR[0];   // the leading NUL
   for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
      fx_tiny_store<L>(stage, tile, lane, rows, total, t);
      fx_tiny_load<L>(stage, rows, total, t + wave_stride, lane);
      uint32_t d[17];
      fx_tiny_span<L>(d, tile, lane);
      const int64_t row_first = ((t << 6) + lane) * T::RPL;
      uint32_t out[(T::RPL + 3) / 4] = {0};
      FxTinySearchRows<L, SCH, 0>::run(d, tabR, tabA, fp, fz, row_first, n, lane, n_deferred, worklist, out);
      fx_tiny_emit<L>(flags, row_first, n, out);
   }
}

template <int L, int SCH>
hipError_t launch_tiny_search(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, uint32_t* ctr, uint32_t* worklist, hipStream_t st) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ctr) ^ 16u);
   const int64_t n_tiles = (n * (int64_t)L + 64 * FxTiny<L>::LS - 1) / (64 * FxTiny<L>::LS);
   int64_t blocks = (n_tiles + 3) / 4;
   if (blocks > 256 * 8) blocks = 256 * 8;
   hipLaunchKernelGGL((fx_search_tiny<L, SCH>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, ctr, clear_next, worklist);
   return hipGetLastError();
}

template <int L, int SCH>
hipError_t launch_tiny(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, uint32_t* ctr, uint32_t* worklist, hipStream_t st) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ctr) ^ 16u);   // the other parity's group of four words
   const int64_t n_tiles = (n * (int64_t)L + 64 * FxTiny<L>::LS - 1) / (64 * FxTiny<L>::LS);
   int64_t blocks = (n_tiles + 3) / 4;
   if (blocks > 256 * 8) blocks = 256 * 8;
   hipLaunchKernelGGL((fx_match_tiny<L, SCH>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, ctr, clear_next, worklist);
   return hipGetLastError();
}
