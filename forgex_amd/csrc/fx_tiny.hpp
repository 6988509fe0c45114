// fx_match_tiny: `.match.` over TINY rows -- 4, 8, 16 or 32 bytes (round 4; BASELINE config 1's shape: `\d{3}-\d{4}` over 8-byte rows).
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

template <int L, int SCH>
__global__ __launch_bounds__(256) void fx_match_tiny(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog, FastParams fp,
                                                      uint8_t* __restrict__ flags, uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ clear_next,
                                                      uint32_t* __restrict__ worklist) {
   static_assert(L == 4 || L == 8 || L == 16 || L == 32, "tiny rows: a divisor of the 64-byte lane span");
   static_assert(SCH == 0 || SCH == 2, "class-level v_perm or nibble tables");
   constexpr int RPL = 64 / L;   // rows per lane
   constexpr bool WIDE = SCH == 2;
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
      const uint2 e = reinterpret_cast<const uint2*>(prog + (WIDE ? h->off_w16A : h->off_fastA))[threadIdx.x];
      reinterpret_cast<uint2*>(tabA_s)[threadIdx.x] = e;
   }
   __syncthreads();
   const F* tabA = tabA_s;
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint4* tile = tiles + wave * 256;
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   const int64_t total = n * (int64_t)L;                 // bytes of the batch (a multiple of 4: no dword straddles the extent)
   const int64_t n_tiles = (total + 4095) >> 12;          // 4 KB per wave and trip = 64 lanes x 64 bytes
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   const bool gated = (h->len_prefix | h->len_suffix | h->len_all) != 0u;   // `.match.` has a literal / prefix / suffix gate
   uint4 stage[4];
   auto load = [&](const int64_t t) {
      const int64_t left = total - (t << 12);
      const uint32_t valid = left <= 0 ? 0u : (left >= 4096 ? 4096u : (uint32_t)left);
      const uint64_t base = reinterpret_cast<uint64_t>(rows) + ((uint64_t)t << 12);
      const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                            __builtin_amdgcn_readfirstlane(valid), 0x00020000);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
         const fx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u + (uint32_t)q * 1024u, 0, FX_LOAD_AUX);
         stage[q] = make_uint4(v.x, v.y, v.z, v.w);
      }
   };
   load(wave_global);
   uint32_t fm0 = 0, fm1 = 0, fm2 = 0, fm3 = 0;   // FINAL verdict of a state: byte q of {fm1, fm0} (v_perm) / of fm0..fm3 (nibble tables)
   if (WIDE) {
      fm0 = h->w16_finalM[0];
      fm1 = h->w16_finalM[1];
      fm2 = h->w16_finalM[2];
      fm3 = h->w16_finalM[3];
   } else {
      fm0 = h->fast_finalM[0];
      fm1 = h->fast_finalM[1];
   }
   for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
      store_tile<4>(stage, tile, lane);
      load(t + wave_stride);   // the ONE reload site of the staging registers
      const int64_t row_first = ((t << 6) + lane) * RPL;   // this lane's first row
      uint32_t out[RPL <= 4 ? 1 : RPL / 4] = {0};          // RPL verdict bytes
#pragma unroll
      for (int j = 0; j < RPL; ++j) {
         const uint32_t off = (uint32_t)(j * L);            // byte offset of row j in the lane's 64-byte span (compile-time)
         uint32_t w[L / 4];
         if constexpr (L == 4) {
            w[0] = *reinterpret_cast<const uint32_t*>(tb + (tile_cell(lane, off >> 4) << 4) + (off & 15u));
         } else if constexpr (L == 8) {
            const uint2 v = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, off >> 4) << 4) + (off & 15u));
            w[0] = v.x;
            w[1] = v.y;
         } else {
#pragma unroll
            for (int c = 0; c < L / 16; ++c) {
               const uint4 v = tile[tile_cell(lane, (off >> 4) + (uint32_t)c)];
               w[4 * c] = v.x;
               w[4 * c + 1] = v.y;
               w[4 * c + 2] = v.z;
               w[4 * c + 3] = v.w;
            }
         }
         uint32_t na = 0;
#pragma unroll
         for (int i = 0; i < L / 4; ++i) na |= w[i];
         uint32_t st = fp.A_init;   // = M_start
         if constexpr (L == 4) {
            F f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = tabA[(w[0] >> (8 * i)) & 0xFFu];
#pragma unroll
            for (int i = 0; i < 4; ++i) st = fxstep(f[i], st, nullptr);
         } else {
#pragma unroll
            for (int g = 0; g < L / 8; ++g) {
               F f[8];
               lookup8(f, w[2 * g], w[2 * g + 1], tabA);
               chain8_fwd(f, st, nullptr);
            }
         }
         uint32_t fin;
         if (WIDE) {
            const uint32_t fw = (st & 8u) ? ((st & 4u) ? fm3 : fm2) : ((st & 4u) ? fm1 : fm0);
            fin = (fw >> ((st & 3u) * 8u)) & 3u;
         } else fin = __builtin_amdgcn_perm(fm1, fm0, st) & 1u;
         uint32_t flag = (st != 0u && fin == 1u) ? 1u : 0u;
         const int64_t row = row_first + j;
         const bool row_ok = row < n;
         if (gated) {   // (wave-uniform: the program has a gate)
            auto rowb = [&](uint32_t k) -> uint32_t { return tb[(tile_cell(lane, (off + k) >> 4) << 4) + ((off + k) & 15u)]; };
            const uint32_t gate = fxrow::match_gate(h, prog, rowb, (uint32_t)L);
            flag = gate == 2u ? 1u : (gate == 0u ? 0u : flag);
         }
         // a byte >= 0x80: the row goes to the row-level fix-up (UTF-8 decode by the general procedure)
         const bool listed = row_ok && (na & 0x80808080u) != 0u;
         const uint64_t em = __builtin_amdgcn_ballot_w64(listed);
         if (em != 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&n_deferred[1], (uint32_t)__builtin_popcountll(em));
            base = __builtin_amdgcn_readfirstlane(base);
            if (listed) worklist[base + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = (uint32_t)row;
         }
         if (listed) flag = FX_NEEDS_GENERAL;
         out[j / 4] |= flag << (8 * (j & 3));
      }
      // the lane's RPL verdict bytes are consecutive in flags[]: one store when all of its rows exist
      if (row_first + RPL <= n) {
         uint8_t* dst = flags + row_first;
         if constexpr (RPL == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)out[0];
         else if constexpr (RPL == 4) *reinterpret_cast<uint32_t*>(dst) = out[0];
         else if constexpr (RPL == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(out[0], out[1]);
         else *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
      } else {
#pragma unroll
         for (int j = 0; j < RPL; ++j)
            if (row_first + j < n) flags[row_first + j] = (uint8_t)(out[j / 4] >> (8 * (j & 3)));
      }
   }
}

// fx_search_tiny: the `.in.` VERDICT (flags only: what the reference's operator returns, forgex.F90:74-160) over the same tiny rows.  Per row
// the reverse automaton walks the L bytes from the row's last byte (start state: after the trailing NUL) -- a hit anywhere is a start
// inside the text, which always yields a span (api_internal_m.F90:140-148), so the verdict is TRUE -- then the leading NUL: a start THERE is
// the leftmost one, and the verdict is the forward walk's (max_match > 2: an accept after at least one more symbol than the NUL), walked
// here with the anchored tables whenever a lane of the wave has such a start.  Rows with a byte >= 0x80 and rows that end in the overlap
// state of a bordered prefix literal (FXP_F_OVERLAP_SINK) go to the row-level fix-up.
template <int L, int SCH>
__global__ __launch_bounds__(256) void fx_search_tiny(const uint8_t* __restrict__ rows, int64_t n, const uint8_t* __restrict__ prog, FastParams fp,
                                                       uint8_t* __restrict__ flags, uint32_t* __restrict__ n_deferred, uint32_t* __restrict__ clear_next,
                                                       uint32_t* __restrict__ worklist) {
   static_assert(L == 4 || L == 8 || L == 16 || L == 32, "tiny rows: a divisor of the 64-byte lane span");
   static_assert(SCH == 0 || SCH == 2, "class-level v_perm or nibble tables");
   constexpr int RPL = 64 / L;
   constexpr bool WIDE = SCH == 2;
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
      const uint2 e = reinterpret_cast<const uint2*>(prog + (WIDE ? h->off_w16R : h->off_fastR))[threadIdx.x];
      reinterpret_cast<uint2*>(tabR_s)[threadIdx.x] = e;
      const uint2 ea = reinterpret_cast<const uint2*>(prog + (WIDE ? h->off_w16A : h->off_fastA))[threadIdx.x];
      reinterpret_cast<uint2*>(tabA_s)[threadIdx.x] = ea;
   }
   __syncthreads();
   const F* tabR = tabR_s;
   const F* tabA = tabA_s;
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint4* tile = tiles + wave * 256;
   const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
   const int64_t total = n * (int64_t)L;
   const int64_t n_tiles = (total + 4095) >> 12;
   const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave, wave_stride = (int64_t)gridDim.x * 4;
   uint4 stage[4];
   auto load = [&](const int64_t t) {
      const int64_t left = total - (t << 12);
      const uint32_t valid = left <= 0 ? 0u : (left >= 4096 ? 4096u : (uint32_t)left);
      const uint64_t base = reinterpret_cast<uint64_t>(rows) + ((uint64_t)t << 12);
      const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0,
                                                                            __builtin_amdgcn_readfirstlane(valid), 0x00020000);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
         const fx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u + (uint32_t)q * 1024u, 0, FX_LOAD_AUX);
         stage[q] = make_uint4(v.x, v.y, v.z, v.w);
      }
   };
   load(wave_global);
   const F fz = tabR[0];   // the leading NUL
   for (int64_t t = wave_global; t < n_tiles; t += wave_stride) {
      store_tile<4>(stage, tile, lane);
      load(t + wave_stride);
      const int64_t row_first = ((t << 6) + lane) * RPL;
      uint32_t out[RPL <= 4 ? 1 : RPL / 4] = {0};
#pragma unroll
      for (int j = 0; j < RPL; ++j) {
         const uint32_t off = (uint32_t)(j * L);
         uint32_t w[L / 4];
         if constexpr (L == 4) {
            w[0] = *reinterpret_cast<const uint32_t*>(tb + (tile_cell(lane, off >> 4) << 4) + (off & 15u));
         } else if constexpr (L == 8) {
            const uint2 v = *reinterpret_cast<const uint2*>(tb + (tile_cell(lane, off >> 4) << 4) + (off & 15u));
            w[0] = v.x;
            w[1] = v.y;
         } else {
#pragma unroll
            for (int c = 0; c < L / 16; ++c) {
               const uint4 v = tile[tile_cell(lane, (off >> 4) + (uint32_t)c)];
               w[4 * c] = v.x;
               w[4 * c + 1] = v.y;
               w[4 * c + 2] = v.z;
               w[4 * c + 3] = v.w;
            }
         }
         uint32_t na = 0;
#pragma unroll
         for (int i = 0; i < L / 4; ++i) na |= w[i];
         uint32_t st = fp.R_start, mx = 0;
         if constexpr (L == 4) {
            F f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = tabR[(w[0] >> (8 * i)) & 0xFFu];
#pragma unroll
            for (int i = 3; i >= 0; --i) {
               st = fxstep(f[i], st, nullptr);
               mx = max(mx, st);
            }
         } else {
#pragma unroll
            for (int g = L / 8 - 1; g >= 0; --g) {
               F f[8];
               lookup8(f, w[2 * g], w[2 * g + 1], tabR);
               mx = max(mx, chain8_back(f, st, nullptr));
            }
         }
         const bool hit = mx >= fp.hit_min;                 // a start inside the text
         const uint32_t sn = fxstep(fz, st, nullptr);
         const bool s_nul = sn >= fp.hit_min;               // a start at the leading NUL
         const int64_t row = row_first + j;
         const bool row_ok = row < n;
         bool verdict = hit;
         if (__builtin_amdgcn_ballot_w64(s_nul) != 0) {   // (wave-uniform; `^`-anchored patterns) forward from the leading NUL
            uint32_t cur = s_nul ? fp.A_init : 0u;
            const F fza = tabA[0];
            cur = fxstep(fza, cur, nullptr);
            bool acc = false;
#pragma unroll
            for (int i = 0; i < L; ++i) {
               const F f = tabA[(w[i / 4] >> (8 * (i & 3))) & 0xFFu];
               cur = fxstep(f, cur, nullptr);
               acc = acc || cur >= fp.acc_min;
            }
            cur = fxstep(fza, cur, nullptr);   // the trailing NUL
            acc = acc || cur >= fp.acc_min;
            verdict = s_nul ? acc : hit;
         }
         const bool listed = row_ok && ((na & 0x80808080u) != 0u || (fp.inv_on != 0u && sn == fp.inv));
         const uint64_t em = __builtin_amdgcn_ballot_w64(listed);
         if (em != 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&n_deferred[1], (uint32_t)__builtin_popcountll(em));
            base = __builtin_amdgcn_readfirstlane(base);
            if (listed) worklist[base + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = (uint32_t)row;
         }
         const uint32_t flag = listed ? (uint32_t)FX_NEEDS_GENERAL : (verdict ? 1u : 0u);
         out[j / 4] |= flag << (8 * (j & 3));
      }
      if (row_first + RPL <= n) {
         uint8_t* dst = flags + row_first;
         if constexpr (RPL == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)out[0];
         else if constexpr (RPL == 4) *reinterpret_cast<uint32_t*>(dst) = out[0];
         else if constexpr (RPL == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(out[0], out[1]);
         else *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
      } else {
#pragma unroll
         for (int j = 0; j < RPL; ++j)
            if (row_first + j < n) flags[row_first + j] = (uint8_t)(out[j / 4] >> (8 * (j & 3)));
      }
   }
}

template <int L, int SCH>
hipError_t launch_tiny_search(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, uint32_t* ctr, uint32_t* worklist, hipStream_t st) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ctr) ^ 16u);
   const int64_t n_tiles = (n * (int64_t)L + 4095) >> 12;
   int64_t blocks = (n_tiles + 3) / 4;
   if (blocks > 256 * 8) blocks = 256 * 8;
   hipLaunchKernelGGL((fx_search_tiny<L, SCH>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, ctr, clear_next, worklist);
   return hipGetLastError();
}

template <int L, int SCH>
hipError_t launch_tiny(const uint8_t* rows, int64_t n, const uint8_t* d_blob, FastParams fp, uint8_t* flags, uint32_t* ctr, uint32_t* worklist, hipStream_t st) {
   uint32_t* clear_next = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ctr) ^ 16u);   // the other parity's group of four words
   const int64_t n_tiles = (n * (int64_t)L + 4095) >> 12;
   int64_t blocks = (n_tiles + 3) / 4;
   if (blocks > 256 * 8) blocks = 256 * 8;
   hipLaunchKernelGGL((fx_match_tiny<L, SCH>), dim3((unsigned)blocks), dim3(256), 0, st, rows, n, d_blob, fp, flags, ctr, clear_next, worklist);
   return hipGetLastError();
}
