// libforgex_amd.so: HIP kernels (gfx950 / CDNA4) + the C ABI of include/forgex_amd.h.
//
// Kernel families (DESIGN.md §0 has the dispatch table, §4 the kernels):
//   fx_search_one<CH>    (fx_one.hpp) a search or a `.match.` over rows of up to 256 bytes in ONE launch; fx_search_multi (fx_multi.hpp):
//                        m patterns in one pass over rows of up to 128 bytes.  Same tile staging and table schemes as:
//   fx_search_fast<CH>   (fx_tile.hpp, with fx_match_fast) the half-row kernel of the headline shape, long rows, literal search.  One wavefront owns a tile of 64 consecutive rows
//                        (64 x 16*CH bytes, contiguous in HBM): 16-byte/lane coalesced buffer loads -> swizzled ds_write_b128
//                        -> transposed ds_read_b128 so that lane r holds row r.  The per-byte state advance is
//                        ONE ds_read_b64 of the fused byte table F[byte] (8 next-state bytes, independent of the
//                        state, so lookups pipeline) + ONE v_perm_b32 (state selects its byte).  Right-to-left
//                        pass with the reverse DFA finds the leftmost match start, a short left-to-right pass with
//                        the anchored DFA finds the longest end.  No MFMA: table lookup, not a contraction.
//   fx_general           one lane = one row through fxrow::run_row (row_engine.hpp): every mode, UTF-8 decode
//                        on device, candidate-list driver, literal search, `.match.`; also the fix-up pass for
//                        rows the fast kernel flags as non-ASCII.
#include <atomic>
#include <functional>

#include "../../include/forgex_amd_bench.h"
#include "fx_multi.hpp"
#include "fx_tiny.hpp"
#include "fx_span.hpp"

#ifndef FX_SINGLE_TU   // the launcher instantiations live in fx_tile_inst.hip (one object per chunk count)
#define FX_X(CH, M, S)                                                  \
   extern template hipError_t launch_fast<CH, M, S> FX_TILE_SIG_FAST;   \
   extern template hipError_t launch_match<CH, M, S> FX_TILE_SIG_MATCH;
FX_TILE_ALL(FX_X)
#undef FX_X
#define FX_X(CH, S, B, G) extern template hipError_t launch_one<CH, S, B, G> FX_ONE_SIG;
FX_ONE_ALL(FX_X)
#undef FX_X
#define FX_Z(CH, G)                                                            \
   extern template hipError_t launch_one_marked<CH, 0, G> FX_ONE_MARKED_SIG;   \
   extern template hipError_t launch_one_marked<CH, 1, G> FX_ONE_MARKED_SIG;   \
   extern template hipError_t launch_one_marked<CH, 2, G> FX_ONE_MARKED_SIG;
FX_Z(16, false) FX_Z(8, false) FX_Z(4, false) FX_Z(2, false) FX_Z(1, false) FX_Z(8, true) FX_Z(4, true) FX_Z(2, true) FX_Z(1, true)
#undef FX_Z
extern template hipError_t launch_one_marked<16, 3, false> FX_ONE_MARKED_SIG;
extern template hipError_t launch_one_marked<8, 3, false> FX_ONE_MARKED_SIG;
extern template hipError_t launch_one_marked<4, 3, false> FX_ONE_MARKED_SIG;
extern template hipError_t launch_one_marked<2, 3, false> FX_ONE_MARKED_SIG;
extern template hipError_t launch_one_marked<1, 3, false> FX_ONE_MARKED_SIG;
extern template hipError_t launch_span<128, 0> FX_SPAN_SIG;
extern template hipError_t launch_span<64, 0> FX_SPAN_SIG;
extern template hipError_t launch_span<32, 0> FX_SPAN_SIG;
extern template hipError_t launch_span<16, 0> FX_SPAN_SIG;
extern template hipError_t launch_span<128, 2> FX_SPAN_SIG;
extern template hipError_t launch_span<64, 2> FX_SPAN_SIG;
extern template hipError_t launch_span<32, 2> FX_SPAN_SIG;
extern template hipError_t launch_span<16, 2> FX_SPAN_SIG;
extern template hipError_t launch_multi<1> FX_MULTI_SIG;
extern template hipError_t launch_multi<2> FX_MULTI_SIG;
extern template hipError_t launch_multi<3> FX_MULTI_SIG;
extern template hipError_t launch_multi<4> FX_MULTI_SIG;
extern template hipError_t launch_multi<6> FX_MULTI_SIG;
extern template hipError_t launch_multi<8> FX_MULTI_SIG;
#endif

// ---- test / experiment hooks: the FXAMD_* environment variables, read once (FxEnv, fx_tile.hpp) ----
#ifndef FX_SPAN_LENS_DEFAULT
#define FX_SPAN_LENS_DEFAULT 111  // row lengths the span kernel takes by default (bit mask: 128, 64, 32, 16; 32 = ragged rows; 64 = nibble tables)
#endif
static FxEnv g_env;
static std::once_flag g_env_once;
static void env_load() {
   auto on = [](const char* name) { return std::getenv(name) != nullptr; };
   auto num = [](const char* name) {
      const char* e = std::getenv(name);
      return e ? std::atoi(e) : 0;
   };
   FxEnv e{};
   e.no_half = on("FXAMD_NO_HALF");
   e.force_general = on("FXAMD_FORCE_GENERAL");
   e.no_w16 = on("FXAMD_NO_W16");
   e.no_byte_dfa = on("FXAMD_NO_BYTE_DFA");
   e.no_a8 = on("FXAMD_NO_A8");
   e.no_spec = on("FXAMD_NO_SPEC");
   e.no_tiny = on("FXAMD_NO_TINY");
   e.no_span = on("FXAMD_NO_SPAN");
   e.no_pack_first = on("FXAMD_NO_PACK_FIRST");
   e.span_lens = std::getenv("FXAMD_SPAN_LENS") ? std::atoi(std::getenv("FXAMD_SPAN_LENS")) : FX_SPAN_LENS_DEFAULT;
   e.no_adapt = on("FXAMD_NO_ADAPT");
   e.multipass = on("FXAMD_MULTIPASS");
   e.no_cache = on("FXAMD_NO_CACHE");
   e.no_multi = on("FXAMD_NO_MULTI");
   e.multi_no_bytes = on("FXAMD_MULTI_NO_BYTES");
   e.multi_inq = on("FXAMD_MULTI_INQ");
   e.multi_serial = on("FXAMD_MULTI_SERIAL");
   e.host_register = on("FXAMD_HOST_REGISTER");
   e.no_latch = on("FXAMD_NO_LATCH");   // the half-row first pass and the span kernel on the plain format of R (test / A-B hook; FXP_F_R_LATCH, program.h)
   e.multi_w16 = on("FXAMD_MULTI_W16");   // automata of 9..16 states (nibble tables) join the shared many-pattern pass (built and tested in round 6, OFF by default: measured no faster, see fxamd_match_multi_device)
   {
      const char* v = std::getenv("FXAMD_SLICE_ROWS");
      int64_t r = v ? std::atoll(v) & ~int64_t(63) : int64_t(1) << 30;
      e.slice_rows = r < 64 ? int64_t(64) : r;
   }
   e.one_grid = num("FXAMD_ONE_GRID");
   e.one_round_mb = num("FXAMD_ONE_ROUND_MB");
   e.one_blocks = num("FXAMD_ONE_BLOCKS");
   e.half_rounds = num("FXAMD_HALF_ROUNDS");
   {
      const char* v = std::getenv("FXAMD_HALF_SCH");
      // bit s: table scheme s (0 v_perm, 1 chain, 2 nibble) stages half rows; bit 3: long rows on the chain tables in 128-byte segments
      // bit 4: 128-byte rows on the chain tables in 64-byte halves
      e.half_sch = v && *v ? std::atoi(v) : 31;   // (test / experiment hook)
   }
   g_env = e;
}
const FxEnv& fx_env() {
   std::call_once(g_env_once, env_load);
   return g_env;
}

// =========================================================================================================
// general kernel: one lane = one row, fxrow::run_row
// =========================================================================================================
struct GlobalRow {
   const uint8_t* p;
   __device__ __forceinline__ uint32_t operator[](int j) const { return p[j]; }
};
struct TileRow {
   const uint8_t* tb;
   uint32_t lane;
   __device__ __forceinline__ uint32_t operator[](int j) const {
      uint32_t k = (uint32_t)j >> 4;
      return tb[(tile_cell(lane, k) << 4) + ((uint32_t)j & 15u)];
   }
};

// Copy the program blob into LDS (prog_lds_bytes > 0) so that every table read of the row procedure is an LDS read.
__device__ __forceinline__ const uint8_t* stage_program(const uint8_t* __restrict__ prog, uint8_t* lds, uint32_t prog_lds_bytes) {
   if (prog_lds_bytes == 0) return prog;
   const uint4* src = reinterpret_cast<const uint4*>(prog);
   uint4* dst = reinterpret_cast<uint4*>(lds);
   for (uint32_t i = threadIdx.x; i < prog_lds_bytes / 16; i += blockDim.x) dst[i] = src[i];
   __syncthreads();
   return lds;
}

// fixup != 0: only rows whose flag is FX_NEEDS_GENERAL are processed
__global__ __launch_bounds__(256) void fx_general(const uint8_t* __restrict__ rows, int64_t n, int32_t L, const uint8_t* __restrict__ prog,
                                                   uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                   int fixup, uint32_t prog_lds_bytes) {
   extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
   const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (fixup) {   // nothing to redo in this block: leave before touching the tables
      const bool mine = row < n && flags[row] == FX_NEEDS_GENERAL;
      if (!__syncthreads_or(mine ? 1 : 0)) return;
   }
   const uint8_t* pbase = stage_program(prog, reinterpret_cast<uint8_t*>(dyn_lds), prog_lds_bytes);
   if (row >= n) return;
   if (fixup && flags[row] != FX_NEEDS_GENERAL) return;
   fxrow::ProgView pv(pbase);
   GlobalRow r{rows + row * (int64_t)L};
   fxrow::Result res;
   fxrow::DfaSim sim(pv);
   fxrow::run_row(pv, sim, r, L, res);
   flags[row] = (uint8_t)res.flag;
   if (from) from[row] = res.from;
   if (to) to[row] = res.to;
}

// Exception rows of a byte-level pass, gathered in a worklist: one lane = one listed row through the general procedure, tables
// staged in LDS (programs whose class-level tables cannot decode, or whose prefilter needs the candidate-list driver).
__global__ __launch_bounds__(256) void fx_fixup_list(const uint8_t* __restrict__ rows, int32_t L, const uint8_t* __restrict__ prog,
                                                      uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                      const uint32_t* __restrict__ worklist, const uint32_t* __restrict__ count_p,
                                                      uint32_t prog_lds_bytes) {
   extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
   const uint32_t count = *count_p;
   if ((uint64_t)blockIdx.x * blockDim.x >= count) return;   // (block-uniform: nothing listed for this block)
   const uint8_t* pbase = stage_program(prog, reinterpret_cast<uint8_t*>(dyn_lds), prog_lds_bytes);
   fxrow::ProgView pv(pbase);
   for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
      const int64_t row = worklist[i];
      GlobalRow r{rows + row * (int64_t)L};
      fxrow::Result res;
      fxrow::DfaSim sim(pv);
      fxrow::run_row(pv, sim, r, L, res);
      flags[row] = (uint8_t)res.flag;
      if (from) from[row] = res.from;
      if (to) to[row] = res.to;
   }
}

// LDS-staged variant for 16-byte-multiple rows: one wave per block, dynamic LDS = 64*L bytes (L <= 1024)
__global__ __launch_bounds__(64) void fx_general_tiled(const uint8_t* __restrict__ rows, int64_t n, int32_t L, const uint8_t* __restrict__ prog,
                                                        uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                        uint32_t prog_lds_bytes) {
   extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
   uint4* tiles = dyn_lds + prog_lds_bytes / 16;
   const uint8_t* pbase = stage_program(prog, reinterpret_cast<uint8_t*>(dyn_lds), prog_lds_bytes);
   const uint32_t lane = threadIdx.x;
   const int CH = L >> 4;
   const int64_t row0 = (int64_t)blockIdx.x << 6;
   const uint4* src = reinterpret_cast<const uint4*>(rows + row0 * (int64_t)L);
   const int64_t rows_left = n - row0;
   const uint32_t valid_pieces = rows_left >= 64 ? 64u * CH : (uint32_t)rows_left * CH;
   for (int q = 0; q < CH; ++q) {
      uint32_t p = q * 64 + lane;
      uint4 v = p < valid_pieces ? src[p] : make_uint4(0, 0, 0, 0);
      tiles[tile_cell(p / CH, p % CH)] = v;
   }
   __syncthreads();
   const int64_t row = row0 + lane;
   if (row >= n) return;
   fxrow::ProgView pv(pbase);
   TileRow r{reinterpret_cast<const uint8_t*>(tiles), lane};
   fxrow::Result res;
   fxrow::DfaSim sim(pv);
   fxrow::run_row(pv, sim, r, L, res);
   flags[row] = (uint8_t)res.flag;
   if (from) from[row] = res.from;
   if (to) to[row] = res.to;
}

// fx_fixup_list with the listed rows gathered into LDS first (16-byte-multiple rows, L <= 1024): one wave per block takes 64 list
// entries at a time.  A listed row costs a serial walk of its bytes; from LDS that walk is several times shorter than from global
// memory, and a short list is all latency.
__global__ __launch_bounds__(64) void fx_fixup_list_tiled(const uint8_t* __restrict__ rows, int32_t L, const uint8_t* __restrict__ prog,
                                                           uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to,
                                                           const uint32_t* __restrict__ worklist, const uint32_t* __restrict__ count_p,
                                                           uint32_t prog_lds_bytes) {
   extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
   const uint32_t count = *count_p;
   if ((uint64_t)blockIdx.x * 64u >= count) return;
   uint4* tiles = dyn_lds + prog_lds_bytes / 16;
   const uint8_t* pbase = stage_program(prog, reinterpret_cast<uint8_t*>(dyn_lds), prog_lds_bytes);
   fxrow::ProgView pv(pbase);
   const uint32_t lane = threadIdx.x;
   const uint32_t CH = (uint32_t)L >> 4;
   for (uint64_t i0 = (uint64_t)blockIdx.x * 64u; i0 < count; i0 += (uint64_t)gridDim.x * 64u) {
      const uint32_t listed = count - i0 >= 64u ? 64u : (uint32_t)(count - i0);
      const uint32_t my_row = lane < listed ? worklist[i0 + lane] : 0u;
      for (uint32_t q = 0; q < CH; ++q) {
         const uint32_t p = q * 64u + lane, R = p / CH, k = p - R * CH;
         const uint32_t src_row = __shfl(my_row, (int)R);
         uint4 v = make_uint4(0, 0, 0, 0);
         if (R < listed) v = *reinterpret_cast<const uint4*>(rows + (int64_t)src_row * L + (k << 4));
         tiles[tile_cell(R, k)] = v;
      }
      __syncthreads();
      if (lane < listed) {
         TileRow r{reinterpret_cast<const uint8_t*>(tiles), lane};
         fxrow::Result res;
         fxrow::DfaSim sim(pv);
         fxrow::run_row(pv, sim, r, L, res);
         flags[my_row] = (uint8_t)res.flag;
         if (from) from[my_row] = res.from;
         if (to) to[my_row] = res.to;
      }
      __syncthreads();
   }
}

// FXP_F_NFA_SIM programs (DFA too large to build): one lane = one row, NFA state sets as bitsets in `scratch`
// (2 * nfa_words words per row of this launch).  Slow by construction; exists so that every valid pattern runs.
__global__ __launch_bounds__(64) void fx_nfa(const uint8_t* __restrict__ rows, int64_t row_begin, int64_t n, int32_t L,
                                              const uint8_t* __restrict__ prog, uint8_t* __restrict__ flags, int32_t* __restrict__ from,
                                              int32_t* __restrict__ to, uint32_t* __restrict__ scratch) {
   const int64_t local = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   const int64_t row = row_begin + local;
   if (row >= n) return;
   fxrow::ProgView pv(prog);
   GlobalRow r{rows + row * (int64_t)L};
   fxrow::NfaSim sim(pv, scratch + local * 2 * (int64_t)pv.h().nfa_words);
   fxrow::Result res;
   fxrow::run_row(pv, sim, r, L, res);
   flags[row] = (uint8_t)res.flag;
   if (from) from[row] = res.from;
   if (to) to[row] = res.to;
}

// number of nonzero flag bytes (fxamd_batch_count): 16 flags per thread and step, one atomic per wave
__global__ void fx_count_flags(const uint8_t* __restrict__ flags, int64_t n, unsigned long long* __restrict__ out) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
   uint32_t c = 0;
   for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
      if (i + 16 <= n && (reinterpret_cast<uintptr_t>(flags + i) & 15u) == 0) {
         const uint4 v = *reinterpret_cast<const uint4*>(flags + i);
         auto nz = [](uint32_t w) { return (uint32_t)__builtin_popcount(((w | (w >> 1) | (w >> 2) | (w >> 3) | (w >> 4) | (w >> 5) | (w >> 6) | (w >> 7)) & 0x01010101u)); };
         c += nz(v.x) + nz(v.y) + nz(v.z) + nz(v.w);
      } else {
         for (int64_t j = i; j < n && j < i + 16; ++j) c += flags[j] != 0 ? 1u : 0u;
      }
   }
   for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
   if ((threadIdx.x & 63u) == 0 && c != 0) atomicAdd(out, (unsigned long long)c);
}

__global__ void fx_fill(uint8_t* flags, int32_t* from, int32_t* to, int64_t n) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   flags[i] = 0;
   if (from) from[i] = 0;
   if (to) to[i] = 0;
}

// packed results <-> flags u8 + from / to int32 (paths without in-kernel packing, and the gathering root's unpack).  One thread
// per 64 rows for the bit words, one per row for the spans.
__global__ void fx_pack(const uint8_t* __restrict__ flags, const int32_t* __restrict__ from, const int32_t* __restrict__ to, int64_t n,
                        uint64_t* __restrict__ bits, uint8_t* __restrict__ pf, uint8_t* __restrict__ pt, uint32_t span_bytes) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   const uint64_t m = __builtin_amdgcn_ballot_w64(i < n && flags[i] != 0);
   if ((threadIdx.x & 63u) == 0 && i < n) bits[i >> 6] = m;
   if (i >= n || span_bytes == 0u) return;
   if (span_bytes == 1u) {
      pf[i] = (uint8_t)from[i];
      pt[i] = (uint8_t)to[i];
   } else if (span_bytes == 2u) {
      reinterpret_cast<uint16_t*>(pf)[i] = (uint16_t)from[i];
      reinterpret_cast<uint16_t*>(pt)[i] = (uint16_t)to[i];
   } else {
      reinterpret_cast<int32_t*>(pf)[i] = from[i];
      reinterpret_cast<int32_t*>(pt)[i] = to[i];
   }
}
__global__ void fx_unpack(const uint64_t* __restrict__ bits, const uint8_t* __restrict__ pf, const uint8_t* __restrict__ pt, int64_t n,
                          uint8_t* __restrict__ flags, int32_t* __restrict__ from, int32_t* __restrict__ to, uint32_t span_bytes) {
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   flags[i] = (uint8_t)((bits[i >> 6] >> (i & 63)) & 1ull);
   if (span_bytes == 0u || !from) return;
   if (span_bytes == 1u) {
      from[i] = pf[i];
      to[i] = pt[i];
   } else if (span_bytes == 2u) {
      from[i] = reinterpret_cast<const uint16_t*>(pf)[i];
      to[i] = reinterpret_cast<const uint16_t*>(pt)[i];
   } else {
      from[i] = reinterpret_cast<const int32_t*>(pf)[i];
      to[i] = reinterpret_cast<const int32_t*>(pt)[i];
   }
}

// =========================================================================================================
// C ABI
// =========================================================================================================
// Device-side state of a handle.  The uploaded image is kept per device; the per-call scratch (counter words, worklist, NFA
// bitsets) is kept per (device, stream): calls on one handle that use DIFFERENT streams -- from one host thread or several --
// own different scratch and may overlap on the device; calls on the same stream are ordered by the stream.  Enqueueing itself is
// serialised by the handle's mutex.
struct DevBlob {
   int device = -1;
   uint8_t* d_blob = nullptr;
};
struct DevScratch {
   int device = -1;
   hipStream_t stream = nullptr;
   uint32_t* d_counter = nullptr;   // two groups of four words used by alternate calls: [0] a first pass deferred tiles, [1] rows in the
                                    // worklist; each first pass zeroes the other group for the call after it
   uint32_t parity = 0;             // which group the next call uses
   uint32_t* d_worklist = nullptr;  // row indices the tile kernels left to a later pass; grown on demand
   int64_t worklist_rows = 0;
   uint32_t* d_nfa_scratch = nullptr;   // bitset scratch of the NFA-simulation kernel (FXP_F_NFA_SIM programs)
   size_t nfa_scratch_rows = 0;
   uint8_t* d_unpacked = nullptr;       // flags + from + to of a packed call that a path without in-kernel packing serves (9 bytes per row)
   int64_t unpacked_rows = 0;
   uint64_t last_use = 0;
};
// host-buffer entry (fxamd_match_batch_host): two chunk slots per device, each with its own stream, device buffers and pinned result
// staging, so that the H2D copy of one chunk overlaps the kernels and the D2H copy of the other
struct HostSlot {
   hipStream_t stream = nullptr;
   hipEvent_t done = nullptr;
   uint8_t* d_rows = nullptr;
   uint8_t* d_flags = nullptr;
   int32_t *d_from = nullptr, *d_to = nullptr;
   uint8_t* h_flags = nullptr;   // pinned
   int32_t *h_from = nullptr, *h_to = nullptr;
   size_t row_bytes = 0;
   int64_t rows_cap = 0;
   bool spans = false;
   int64_t pending_row0 = -1, pending_n = 0;   // results in flight: user rows [row0, row0 + n)
};
struct HostPipe {
   int device = -1;
   HostSlot slot[2];
};
// The pipes are shared by all handles of the process (a scalar `pattern .in. text` of the Fortran module compiles a fresh handle per
// call: per-handle streams, events and pinned buffers would cost more than the match): a caller takes a free pipe of its device, or
// makes one, and hands it back.  They live until the process ends.
static std::mutex g_pipe_mu;
static std::vector<HostPipe*> g_free_pipes;
static HostPipe* acquire_pipe(int dev) {
   std::lock_guard<std::mutex> g(g_pipe_mu);
   for (size_t i = 0; i < g_free_pipes.size(); ++i)
      if (g_free_pipes[i]->device == dev) {
         HostPipe* hp = g_free_pipes[i];
         g_free_pipes.erase(g_free_pipes.begin() + (long)i);
         return hp;
      }
   HostPipe* hp = new (std::nothrow) HostPipe();
   if (hp) hp->device = dev;
   return hp;
}
static void release_pipe(HostPipe* hp) {
   std::lock_guard<std::mutex> g(g_pipe_mu);
   g_free_pipes.push_back(hp);
}
struct fxamd_program {
   int refs = 1;                 // (guarded by g_cache_mu) handles given out + the compile cache's own reference
   std::string cache_key;        // non-empty: this program sits in the compile cache
   fxc::Program prog;
   std::mutex mu;        // guards everything below and serialises enqueueing on this handle
   std::vector<DevBlob> blobs;
   std::vector<DevScratch> scratch;
   uint64_t use_clock = 0;
   int last_path = 0;
   std::atomic<size_t> held{0};   // bytes of trimmable scratch this program keeps (refreshed under `mu`, read by the cache's accounting without it)
};
static constexpr size_t FX_MAX_SCRATCH_SETS = 16;

static thread_local int g_last_hip_error = 0;
static int hip_fail(hipError_t e) {
   g_last_hip_error = (int)e;
   return FXAMD_E_HIP;
}
#define FX_HIP(call)                                  \
   do {                                               \
      hipError_t _e = (call);                         \
      if (_e != hipSuccess) return hip_fail(_e);      \
   } while (0)

// ---- compile cache -------------------------------------------------------------------------------------------------------
// The reference's operators are elemental and recompile the pattern per element; a Fortran loop of scalar `pattern .in. text`
// calls does the same through fxamd_compile.  Compiled programs (with their uploaded tables and scratch) are therefore shared:
// fxamd_compile hands out the cached handle of an identical (op, pattern) and fxamd_program_free only drops a reference.  Handles
// are thread-safe (p->mu), so sharing one between callers is what the header already promises.  FXAMD_NO_CACHE=1 turns it off.
static std::mutex g_cache_mu;
static std::vector<fxamd_program*> g_cache;   // most recently used last
static constexpr size_t FX_CACHE_MAX = 64;
static void destroy_program(fxamd_program* p);
static void trim_scratch(fxamd_program* p);
// Scratch a cached program may keep between calls (worklists, unpacked staging, NFA bitsets).  Trimming means hipFree, and every
// hipFree synchronises the whole device: a Fortran loop of `pattern .in. strs(:)` does compile -> match -> free per call, so the
// buffers stay with the cached program unless they exceed this budget (eviction from the cache frees everything anyway).
static constexpr size_t FX_TRIM_BUDGET = size_t(512) << 20;
// ... and what ALL cached programs may keep together: beyond it the least recently used idle ones are trimmed (64 cached programs of a
// service that matches large batches under many patterns would otherwise sit on 64 x 512 MB next to the caller's own allocator)
static constexpr size_t FX_TRIM_BUDGET_ALL = size_t(1) << 30;
static size_t trim_cache(size_t keep_bytes);
static void release_program(fxamd_program* p) {
   bool dead = false, idle = false;
   {
      std::lock_guard<std::mutex> g(g_cache_mu);
      dead = --p->refs == 0;
      idle = p->refs == 1 && !p->cache_key.empty();   // only the cache holds it now
      if (idle) ++p->refs;   // pinned while it is trimmed: a concurrent fxamd_compile may evict it from the cache meanwhile
   }
   if (dead) {
      destroy_program(p);
      return;
   }
   if (!idle) return;
   trim_scratch(p);   // (no-op below FX_TRIM_BUDGET)
   {
      std::lock_guard<std::mutex> g(g_cache_mu);
      dead = --p->refs == 0;   // the cache dropped it while it was pinned
   }
   if (dead) destroy_program(p);
   (void)trim_cache(FX_TRIM_BUDGET_ALL);   // (no-op while the cached programs together stay below the budget)
}

// (callers hold p->mu)
static int blob_for_device(fxamd_program* p, int dev, uint8_t** out) {
   for (DevBlob& b : p->blobs)
      if (b.device == dev) {
         *out = b.d_blob;
         return FXAMD_OK;
      }
   DevBlob nb;
   nb.device = dev;
   FX_HIP(hipMalloc((void**)&nb.d_blob, p->prog.blob.size()));
   const hipError_t e = hipMemcpy(nb.d_blob, p->prog.blob.data(), p->prog.blob.size(), hipMemcpyHostToDevice);
   if (e != hipSuccess) {
      (void)hipFree(nb.d_blob);
      return hip_fail(e);
   }
   p->blobs.push_back(nb);
   *out = nb.d_blob;
   return FXAMD_OK;
}
static void free_scratch(DevScratch& s) {
   if (s.d_counter) (void)hipFree(s.d_counter);
   if (s.d_worklist) (void)hipFree(s.d_worklist);
   if (s.d_nfa_scratch) (void)hipFree(s.d_nfa_scratch);
   if (s.d_unpacked) (void)hipFree(s.d_unpacked);
   s = DevScratch();
}
static int scratch_for(fxamd_program* p, int dev, hipStream_t st, DevScratch** out) {
   DevScratch* found = nullptr;
   size_t on_dev = 0;
   for (DevScratch& s : p->scratch) {
      if (s.device != dev) continue;
      ++on_dev;
      if (s.stream == st) found = &s;
   }
   if (!found) {
      if (on_dev >= FX_MAX_SCRATCH_SETS) {
         // a caller that keeps coming with new streams: recycle the least recently used set of this device once the device is idle
         FX_HIP(hipDeviceSynchronize());
         for (DevScratch& s : p->scratch)
            if (s.device == dev && (!found || s.last_use < found->last_use)) found = &s;
         found->stream = st;
      } else {
         p->scratch.emplace_back();
         found = &p->scratch.back();
         found->device = dev;
         found->stream = st;
      }
   }
   if (!found->d_counter) {
      FX_HIP(hipMalloc((void**)&found->d_counter, 64));   // two groups of four words + the persistent word of FX_ADAPT_CALLS (+ spare)
      // zeroed IN STREAM ORDER: a memset on the null stream is not ordered against kernels on a non-blocking stream
      FX_HIP(hipMemsetAsync(found->d_counter, 0, 64, st));
   }
   found->last_use = ++p->use_clock;
   *out = found;
   return FXAMD_OK;
}
static int grow_worklist(DevScratch* s, int64_t rows) {
   if (s->worklist_rows >= rows) return FXAMD_OK;
   // (the old list may still be read by kernels of an earlier call on this stream: hipFree waits for the device)
   if (s->d_worklist) (void)hipFree(s->d_worklist);
   s->d_worklist = nullptr;
   s->worklist_rows = 0;
   FX_HIP(hipMalloc((void**)&s->d_worklist, (size_t)rows * 4));
   s->worklist_rows = rows;
   return FXAMD_OK;
}

// chunk count the tile kernels are instantiated for that covers row_len (0 = none)
static int tile_chunks(int64_t row_len) {
   static const int inst[] = {1, 2, 4, 8, 12, 16};   // (round 6: 3 and 6 retired -- rows of 48 / 96 bytes run as ragged rows of the 4- / 8-chunk kernels; 12 stays: config 4)
   for (int c : inst)
      if (row_len <= 16 * c) return c;
   return 0;
}
// the one-launch kernel's instantiation for rows of up to 256 bytes: whole chunks as above; ragged rows (any other length) take the
// power of two at or above their chunk count -- CH lanes share a row in the loader, and the work follows the row length, not CH
static int one_chunks(int64_t row_len) {
   const int c = tile_chunks(row_len);
   if (c == 0 || row_len == 16 * c) return c;
   int p = 1;
   while (16 * p < row_len) p *= 2;
   return p;
}
// long rows (any length): walked in 256-byte segments by the CH = 16 instantiations
static bool long_row(int64_t row_len) { return row_len > 256 && row_len <= 65536; }
// (the multi-pass kernels take ragged rows on the power-of-two instantiations too: round 6 -- launch_fast / launch_match, fx_tile.hpp)
static int chunks_of(int64_t row_len) { return long_row(row_len) ? 16 : one_chunks(row_len); }
// ... and the larger of the two instantiations a row length may meet (LDS budgets of table schemes that must hold for both pipelines)
static int chunks_max(int64_t row_len) { return long_row(row_len) ? 16 : std::max(tile_chunks(row_len), one_chunks(row_len)); }
// ... and the tile columns that instantiation allocates per row (the LDS budgets of fast_scheme / bytes_ok are coupled to the allocation: ADVICE r05)
static size_t tile_cols_max(int64_t row_len) { return long_row(row_len) ? (size_t)fx_tile_cols<16, true, true>() : (size_t)chunks_max(row_len) + 1; }
// what a pass needs to know beyond the tables: deferral policy of a first pass, gate word of a marked-tile pass
struct PassOpts {
   uint32_t defer_tiles = 0, gate_word = 0;
   uint32_t* worklist = nullptr;   // BYTES passes append exception rows, the worklist decode pass (MODE 4) reads them
   int64_t grid_tiles = 0;         // MODE 4: upper bound of the worklist's tiles (the count itself lives on the device)
   bool half = false;              // first pass over 256-byte rows with the 8-state tables: stage HALF rows (CH = 8 segment walker)
   uint32_t out_mode = 0;          // the half-row first pass writes PACKED results (marks in `worklist`'s memory)
};
// 256-byte rows: the multi-pass pipeline whose first pass stages HALF rows (8 KB of LDS per wave: four waves per SIMD instead of two).
// Round 2: the 8-state v_perm tables.  Round 4: the chain tables (one dependent LDS read per byte: latency-bound, so twice the waves is
// twice the rate -- 17-state pattern with spans 1.19 -> 0.72 ms, an e-mail pattern 0.97 -> 0.49 ms on config-3 rows) and, with spans, the
// nibble tables (0.52 -> 0.48 ms; flags only 0.49 -> 0.465 ms); profiles/r04_half_chain_ab.txt.
// (128-byte rows with 64-byte halves were tried in round 3 and are NOT dispatched: a 64-byte piece is half of a 128-byte line, every line
//  is fetched twice -- config 5's shard 0.73-0.76 ms against 0.377 ms on the one-launch kernel, gpurun call r03_c14; round 4, with the
//  default cache policy on the loads so that the second half meets its line in L2: 0.45 ms against 0.366 ms, gpurun call r04_c26)
static bool half_rows(const FxpHeader& h, int scheme, int64_t row_len) {
   // (FXAMD_NO_HALF: test / experiment hook -- these rows on the one-launch kernel; FXAMD_HALF_SCH: bit s = table scheme s takes half rows)
   if (fx_env().no_half || (h.flags & FXP_F_PREFIX_CHECK)) return false;
   // 128-byte rows on the chain tables: 64-byte halves (bit 4 of the hook).  The chain scheme's dependent LDS read per byte is latency-bound:
   // four waves per SIMD on a 4 KB tile against the one-launch kernel's three (and class-level tables on pure-ASCII tiles instead of
   // byte-level ones everywhere): the 17-state pattern over config 5's shard 0.743 -> 0.496 ms (gpurun call r04_c47).  With the v_perm
   // tables the split lines cost more than the waves gain (config 5: 0.45 against 0.366 ms, see below).
   if (row_len == 128 && scheme == 1 && (fx_env().half_sch & 16) != 0 && !(h.flags & FXP_F_NEEDS_NONASCII)) return true;
   // (the nibble tables on 64-byte halves lose: `\d{3}-\d{4}` over config 5's shard 0.327 -> 0.366 ms, gpurun call r04_c48)
   if (row_len != 256 || scheme < 0 || scheme > 2 || ((fx_env().half_sch >> scheme) & 1) == 0) return false;
   // A program whose every match needs a byte >= 0x80 is given text that holds such bytes: the half-row first pass would load every tile
   // only to defer it to the follow-up (config 4's pattern and text in 256-byte rows: 0.1135 ms against 0.0838 ms on the one-launch
   // kernel, which answers the pure-ASCII tiles of such programs with an OR of their words anyway; gpurun call r04_c38)
   if (h.flags & FXP_F_NEEDS_NONASCII) return false;
   return true;
}
// ... and whether that first pass stages half rows or whole ones (flags only, v_perm tables: whole rows, on the memory path; flags only,
// nibble tables: half rows at three waves per SIMD, 0.487-0.493 -> 0.464-0.466 ms on config-3 rows, gpurun call r04_c44)
static bool half_staging(int scheme, bool spans) { return spans || scheme != 0; }
// `.match.` over 256-byte rows on the chain tables: the multi-pass pipeline with a half-row first pass too (fx_match_fast<8,...,LONG>)
// (128-byte rows: `.match.` of a 23-state pattern over config 5's shard 0.4518 -> 0.4443 ms on 64-byte halves -- within a box's drift: not dispatched)
static bool match_half_rows(const FxpHeader& h, int scheme, int64_t row_len) { return scheme == 1 && row_len == 256 && half_rows(h, scheme, row_len); }

template <int MODE, int SCH>
static hipError_t launch_match_any(const FxpHeader& h, const uint8_t* d_blob, const uint8_t* d_rows, int64_t n, int64_t row_len,
                                   uint8_t* d_flags, uint32_t* n_deferred, hipStream_t st, PassOpts po = PassOpts()) {
   constexpr bool BYTES = MODE == 2 || MODE == 3;
   const uint32_t class_map_bytes = (1024u + h.n_pages * 64u) * 2u;
   constexpr bool CHAIN = SCH == 1, WIDE = SCH == 2;
   const uint32_t chain_bytes = CHAIN ? ((512u + (BYTES ? h.byte_TA_bytes : h.chain_TA_bytes) + 15u) & ~15u) : 0u;
   FastParams fp{0, BYTES ? h.byte_A_init : (CHAIN ? h.chain_A_init : h.fast_A_init * 0x01010101u), 0, 0, BYTES ? h.byte_inv_A : 0u, 0u, po.defer_tiles, po.gate_word, 0, 0, 0};
   if (WIDE) {   // encoded state bytes, replicated like the 8-state scheme's
      fp.A_init = BYTES ? h.bw16_A_init : h.w16_A_init;
      fp.inv = BYTES ? h.bw16_inv_A : 0u;
   }
   const bool long8 = CHAIN && (MODE == 0 || MODE == 2) && long_row(row_len) && (fx_env().half_sch & 8) != 0;   // (as launch_fast_any's)
   switch (((po.half && SCH == 1 && MODE == 0) || long8) ? 8 : chunks_of(row_len)) {
      case 1: return launch_match<1, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 2: return launch_match<2, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 4: return launch_match<4, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 8: return launch_match<8, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 12: return launch_match<12, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      default: return launch_match<16, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
   }
}

static bool row_len_ok(const FxpHeader& h, const uint8_t* d_rows, int64_t row_len) {
   // (any base address: a tile is addressed through a buffer resource whose base is the tile's first byte, so a batch that does not
   //  start at a 16-byte multiple only turns the tile loads into unaligned ones -- it used to fall to the general kernel, 20x slower)
   (void)d_rows;
   if (long_row(row_len)) return true;
   if (row_len < 2 || row_len > 256) return false;   // (any length in between: rows that are not whole chunks are padded in LDS;
                                                      //  a one-byte row can be the single blank of api_internal_m.F90:68-74)
   if (row_len == 16 * tile_chunks(row_len)) return true;       // whole chunks: fully coalesced tile loads
   return (h.flags & FXP_F_RAGGED_OK) != 0;                        // padded in LDS with the inert symbol 255
}
// class-level table scheme for these rows: -1 = tile kernel not applicable, 0 = v_perm (<= 8 states), 2 = wide v_perm (<= 16),
// 1 = chain (tables must fit the CU's LDS next to the tiles)   [the numbers are the kernels' SCH template argument]
static int fast_scheme(const FxpHeader& h, const uint8_t* d_rows, int64_t row_len) {
   if (fx_env().force_general) return -1;   // test hook: the general kernel (one lane per row) for everything
   if ((h.mode != FXP_MODE_SEARCH_ENGINE && h.mode != FXP_MODE_MATCH_ENGINE && h.mode != FXP_MODE_SEARCH_LITERAL) || !row_len_ok(h, d_rows, row_len)) return -1;
   // FXP_F_PREFIX_CHECK (round 6): the per-row check of the start lives in the one-launch kernel's scan (fx_scan_tile, GEN instantiations): rows of up to 256 bytes
   // there; longer rows and the multi-pass test hook keep the general kernel
   if ((h.flags & FXP_F_PREFIX_CHECK) && (long_row(row_len) || fx_env().multipass)) return -1;
   if (h.flags & FXP_F_FAST_OK) return 0;
   if ((h.flags & FXP_F_W16_OK) && !fx_env().no_w16) return 2;
   if (h.flags & FXP_F_CHAIN_OK) {
      // (tile columns: chunks + the end-of-row column; the long-row walkers with spans have one more -- fx_tile_cols<16, true, true>, the window capture)
      const size_t need = (size_t)4 * 64 * 16 * tile_cols_max(row_len) + 512 + h.chain_TR_bytes + h.chain_TA_bytes + 16 + (1024u + h.n_pages * 64u) * 2u;
      if (need <= 150 * 1024) return 1;
   }
   return -1;
}
// scheme of the byte-level tables: 2 = wide v_perm when both automata have <= 16 states, else 1 = chain
static int bytes_scheme(const FxpHeader& h) { return ((h.flags & FXP_F_BYTE_W16) && !fx_env().no_w16) ? 2 : 1; }
// byte-level tables usable for these rows: whole chunks only (no inert pad byte exists: every byte value means something)
// (`ragged_too`: the one-launch kernel keeps ragged rows left-aligned with the NUL / KILL symbols behind the text -- no pad symbol, so
//  the byte-level tables run on them too; the multi-pass kernels pad with the inert symbol 255, which byte-level tables do not have)
static bool bytes_ok(const FxpHeader& h, const uint8_t* d_rows, int64_t row_len, bool ragged_too = false) {
   if (fx_env().no_byte_dfa) return false;   // test hook: exercise the decode pass instead
   if (!(h.flags & FXP_F_BYTE_DFA) || !row_len_ok(h, d_rows, row_len) || (!long_row(row_len) && !ragged_too && row_len != 16 * tile_chunks(row_len))) return false;
   return (size_t)4 * 64 * 16 * tile_cols_max(row_len) + 512 + h.byte_TR_bytes + h.byte_TA_bytes + 16 <= 150 * 1024;
}

// the latched format of R (FXP_F_R_LATCH): "latched" is state >= 4, a base state s is a hit state iff s >= fast_hitR_min
static void latch_params(FastParams& fp, const FxpHeader& h) {
   fp.latch = 1u;
   fp.hit_min = 4u * 0x01010101u;
   fp.hit_base = h.fast_hitR_min * 0x01010101u;
}
template <int MODE, int SCH>
static hipError_t launch_fast_any(const FxpHeader& h, const uint8_t* d_blob, const uint8_t* d_rows, int64_t n, int64_t row_len,
                                  uint8_t* d_flags, int32_t* d_from, int32_t* d_to, uint32_t* n_deferred, hipStream_t st, PassOpts po = PassOpts()) {
   constexpr bool BYTES = MODE == 2 || MODE == 3;
   const uint32_t class_map_bytes = (1024u + h.n_pages * 64u) * 2u;
   constexpr bool CHAIN = SCH == 1, WIDE = SCH == 2;
   const uint32_t chain_bytes = CHAIN ? ((512u + (BYTES ? h.byte_TR_bytes + h.byte_TA_bytes : h.chain_TR_bytes + h.chain_TA_bytes) + 15u) & ~15u) : 0u;
   FastParams fp{h.fast_R_start * 0x01010101u, h.fast_A_init * 0x01010101u, h.fast_hitR_min * 0x01010101u, h.fast_accA_min * 0x01010101u,
                 0u, 0u, po.defer_tiles, po.gate_word, h.mode == FXP_MODE_SEARCH_LITERAL ? h.len_all : 0u, 0u, po.out_mode};
   if (WIDE) {   // encoded state bytes, replicated like the 8-state scheme's
      fp.R_start = BYTES ? h.bw16_R_start : h.w16_R_start;
      fp.A_init = BYTES ? h.bw16_A_init : h.w16_A_init;
      fp.hit_min = BYTES ? h.bw16_hit_min : h.w16_hit_min;
      fp.acc_min = BYTES ? h.bw16_acc_min : h.w16_acc_min;
      fp.inv = BYTES ? h.bw16_inv_R : 0u;
   } else if (BYTES) {
      fp.R_start = h.byte_R_start;
      fp.A_init = h.byte_A_init;
      fp.hit_min = h.byte_hit_min;
      fp.acc_min = h.byte_acc_min;
      fp.inv = h.byte_inv_R;
   } else if (CHAIN) {   // states are row byte offsets, compared as plain integers
      fp.R_start = h.chain_R_start;
      fp.A_init = h.chain_A_init;
      fp.hit_min = h.chain_hit_min;
      fp.acc_min = h.chain_acc_min;
   }
   if (MODE == 0 && (h.flags & FXP_F_OVERLAP_SINK)) {   // bordered prefix literal: rows whose backward pass ends in R's overlap state
      fp.inv_on = 1u;
      fp.inv = WIDE ? h.R_inv : (CHAIN ? h.R_inv * h.chain_row_bytes : h.R_inv * 0x01010101u);
   }
   // the half-row first pass of 256-byte rows with spans: the LATCHED format of R where the program has it (<= 4 states in R: FXP_F_R_LATCH, round 6)
   if (MODE == 0 && SCH == 0 && po.half && row_len == 256 && d_from != nullptr && d_to != nullptr && (h.flags & FXP_F_R_LATCH) && !fx_env().no_latch)
      latch_params(fp, h);
   // (rows longer than 256 bytes on the chain tables: 128-byte segments at four waves per SIMD -- fx_search_fast NOHALF; bit 3 of the hook)
   const bool long8 = CHAIN && (MODE == 0 || MODE == 2) && long_row(row_len) && (fx_env().half_sch & 8) != 0;
   switch ((po.half || long8) ? (row_len == 128 ? 4 : 8) : chunks_of(row_len)) {
      case 1: return launch_fast<1, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, d_from, d_to, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 2: return launch_fast<2, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, d_from, d_to, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 4: return launch_fast<4, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, d_from, d_to, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 8: return launch_fast<8, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, d_from, d_to, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      case 12: return launch_fast<12, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, d_from, d_to, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
      default: return launch_fast<16, MODE, SCH>(d_rows, n, d_blob, fp, d_flags, d_from, d_to, n_deferred, class_map_bytes, chain_bytes, (uint32_t)row_len, st, po.worklist, po.grid_tiles);
   }
}

// ---- fx_search_one: the whole search in one launch (rows of up to 256 bytes, class-level tables that can decode UTF-8) ----------
// FastParams of one table family: class-level tables of scheme `sch`, or (bytes) the byte-level tables in the chain / wide format
static FastParams params_of(const FxpHeader& h, int sch, bool bytes) {
   FastParams fp{h.fast_R_start * 0x01010101u, h.fast_A_init * 0x01010101u, h.fast_hitR_min * 0x01010101u, h.fast_accA_min * 0x01010101u,
                 0u, 0u, 0u, 0u, (!bytes && h.mode == FXP_MODE_SEARCH_LITERAL) ? h.len_all : 0u, 0u, 0u};
   if (sch == 3) {   // byte-level tables, FXP_F_BYTE_A8: nibble format backwards, 8-state v_perm format (replicated state bytes) forwards
      fp = params_of(h, 2, true);
      fp.A_init = h.b8_A_init * 0x01010101u;
      fp.acc_min = h.b8_acc_min * 0x01010101u;
      fp.spec |= ((h.flags & FXP_F_SPEC_FWD) && !fx_env().no_spec) ? 1u : 0u;   // (FXAMD_NO_SPEC: test / experiment hook)
      return fp;
   }
   if (sch == 2) {   // encoded state bytes, replicated like the 8-state scheme's
      fp.R_start = bytes ? h.bw16_R_start : h.w16_R_start;
      fp.A_init = bytes ? h.bw16_A_init : h.w16_A_init;
      fp.hit_min = bytes ? h.bw16_hit_min : h.w16_hit_min;
      fp.acc_min = bytes ? h.bw16_acc_min : h.w16_acc_min;
      fp.inv = bytes ? h.bw16_inv_R : 0u;
   } else if (bytes) {
      fp.R_start = h.byte_R_start;
      fp.A_init = h.byte_A_init;
      fp.hit_min = h.byte_hit_min;
      fp.acc_min = h.byte_acc_min;
      fp.inv = h.byte_inv_R;
   } else if (sch == 1) {   // states are row byte offsets, compared as plain integers
      fp.R_start = h.chain_R_start;
      fp.A_init = h.chain_A_init;
      fp.hit_min = h.chain_hit_min;
      fp.acc_min = h.chain_acc_min;
   }
   if (!bytes && (h.flags & FXP_F_OVERLAP_SINK)) {   // bordered prefix literal: rows whose backward pass ends in R's overlap state
      fp.inv_on = 1u;
      fp.inv = sch == 2 ? h.R_inv : (sch == 1 ? h.R_inv * h.chain_row_bytes : h.R_inv * 0x01010101u);
   }
   if (h.mode == FXP_MODE_MATCH_ENGINE) fp.inv = bytes ? (sch == 2 ? h.bw16_inv_A : h.byte_inv_A) : 0u;   // `.match.`: A only (A_init holds M_start)
   if (bytes && (h.flags & FXP_F_NEEDS_NONASCII) && !fx_env().no_spec) fp.spec |= 2u;   // pure-ASCII rows hold no match (fx_search_one's shortcuts)
   return fp;
}
template <int SCH, int BSCH, bool GEN>
static hipError_t launch_one_ch(const FxpHeader& h, const uint8_t* d_blob, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags,
                                int32_t* d_from, int32_t* d_to, hipStream_t st, uint32_t out_mode) {
   const bool is_match = h.mode == FXP_MODE_MATCH_ENGINE;
   const FastParams fp = params_of(h, SCH, false), fpb = BSCH != 0 ? params_of(h, BSCH, true) : FastParams{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
   const uint32_t class_map_bytes = (1024u + h.n_pages * 64u) * 2u;
   const uint32_t table_bytes = (SCH == 1 ? ((512u + h.chain_TR_bytes + h.chain_TA_bytes + 15u) & ~15u) : 0u) +
                                (BSCH == 1 ? ((512u + h.byte_TR_bytes + h.byte_TA_bytes + 15u) & ~15u) : 0u);
   const uint32_t Lr = (uint32_t)row_len;
   switch (one_chunks(row_len)) {
      case 1: return launch_one<1, SCH, BSCH, GEN>(d_rows, n, d_blob, fp, fpb, d_flags, d_from, d_to, class_map_bytes, table_bytes, Lr, st, out_mode, is_match);
      case 2: return launch_one<2, SCH, BSCH, GEN>(d_rows, n, d_blob, fp, fpb, d_flags, d_from, d_to, class_map_bytes, table_bytes, Lr, st, out_mode, is_match);
      case 4: return launch_one<4, SCH, BSCH, GEN>(d_rows, n, d_blob, fp, fpb, d_flags, d_from, d_to, class_map_bytes, table_bytes, Lr, st, out_mode, is_match);
      case 8: return launch_one<8, SCH, BSCH, GEN>(d_rows, n, d_blob, fp, fpb, d_flags, d_from, d_to, class_map_bytes, table_bytes, Lr, st, out_mode, is_match);
      case 12: return launch_one<12, SCH, BSCH, GEN>(d_rows, n, d_blob, fp, fpb, d_flags, d_from, d_to, class_map_bytes, table_bytes, Lr, st, out_mode, is_match);
      default: return launch_one<16, SCH, BSCH, GEN>(d_rows, n, d_blob, fp, fpb, d_flags, d_from, d_to, class_map_bytes, table_bytes, Lr, st, out_mode, is_match);
   }
}
// bsch: 0 = no byte-level tables for these rows, 1 chain, 2 wide
static hipError_t launch_one_any(int sch, int bsch, bool gen, const FxpHeader& h, const uint8_t* d_blob, const uint8_t* d_rows, int64_t n, int64_t row_len,
                                 uint8_t* d_flags, int32_t* d_from, int32_t* d_to, hipStream_t st, uint32_t out_mode) {
#define FX_ONE_CASE(S, B)                                                                                                        \
   if (sch == S && bsch == B)                                                                                                     \
      return gen ? launch_one_ch<S, B, true>(h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, st, out_mode)                 \
                 : launch_one_ch<S, B, false>(h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, st, out_mode);
   FX_ONE_CASE(0, 0) FX_ONE_CASE(1, 0) FX_ONE_CASE(2, 0) FX_ONE_CASE(0, 1) FX_ONE_CASE(0, 2) FX_ONE_CASE(0, 3) FX_ONE_CASE(1, 1) FX_ONE_CASE(1, 2) FX_ONE_CASE(2, 1) FX_ONE_CASE(2, 2)
#undef FX_ONE_CASE
   return hipErrorInvalidValue;
}
// Which byte-level format rides along in the one-launch kernel: wide v_perm when the tables exist in it and two blocks per CU still
// fit next to it (4 tiles + class-level tables + 8 KB), else the chain format (a few hundred bytes to a few KB), 0 = none.
static int one_bytes_scheme(const FxpHeader& h, const uint8_t* d_rows, int64_t row_len, int sch) {
   if (!bytes_ok(h, d_rows, row_len, true)) return 0;
   const size_t tiles_b = (size_t)4 * 64 * 16 * (one_chunks(row_len) + 1);
   const size_t cls_b = sch == 0 ? 4096 : (sch == 2 ? 4096 : 512 + h.chain_TR_bytes + h.chain_TA_bytes + 16);
   if (bytes_scheme(h) == 2 && tiles_b + cls_b + 4096 + 2048 <= 80 * 1024)   // (3: with the forward automaton in the v_perm format, FXP_F_BYTE_A8)
      return (sch == 0 && (h.flags & FXP_F_BYTE_A8) && h.mode == FXP_MODE_SEARCH_ENGINE && !fx_env().no_a8) ? 3 : 2;
   if (tiles_b + cls_b + 512 + h.byte_TR_bytes + h.byte_TA_bytes + 16 + 2048 <= 150 * 1024) return 1;
   return 0;
}

// runtime scheme -> instantiation (byte-level modes have no 8-state variant)
template <int MODE>
static hipError_t fast_by(int sch, const FxpHeader& h, const uint8_t* d_blob, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags,
                          int32_t* d_from, int32_t* d_to, uint32_t* ctr, hipStream_t st, PassOpts po) {
   if constexpr (MODE != 2 && MODE != 3)
      if (sch == 0) return launch_fast_any<MODE, 0>(h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, po);
   if (sch == 2) return launch_fast_any<MODE, 2>(h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, po);
   return launch_fast_any<MODE, 1>(h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, po);
}
template <int MODE>
static hipError_t match_by(int sch, const FxpHeader& h, const uint8_t* d_blob, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags,
                           uint32_t* ctr, hipStream_t st, PassOpts po) {
   if constexpr (MODE != 2 && MODE != 3)
      if (sch == 0) return launch_match_any<MODE, 0>(h, d_blob, d_rows, n, row_len, d_flags, ctr, st, po);
   if (sch == 2) return launch_match_any<MODE, 2>(h, d_blob, d_rows, n, row_len, d_flags, ctr, st, po);
   return launch_match_any<MODE, 1>(h, d_blob, d_rows, n, row_len, d_flags, ctr, st, po);
}
static bool scheme_decodes_utf8(const FxpHeader& h, int sch) {   // the class-level tables hold the 128+class / SKIP rows
   return (h.flags & (sch == 0 ? FXP_F_FAST_UTF8 : (sch == 2 ? FXP_F_W16_UTF8 : FXP_F_CHAIN_UTF8))) != 0;
}

// The span kernel (fx_span.hpp; round 5): searches with spans over rows of 128 / 64 / 32 / 16 bytes on the 8-state tables -- first pass + ONE
// gated follow-up (the one-launch kernel's MARKED instantiation over the tiles the first pass marked: bytes >= 0x80, rows in the overlap
// state of a bordered prefix literal).  Measured against the one-launch kernel in one allocation (gpurun call r05_c4): config 5's shard
// 0.372 -> 0.342 ms; `[a-z]+\d+` over config-5 bytes viewed as rows of 64 / 32 / 16 B (one row in 4 / 8 / 16 matches) 0.454 -> 0.362,
// 0.722 -> 0.410, 0.970 -> 0.516 ms.  NOT for candidate-list driver programs on rows of up to 64 bytes: their matches are sparse (a
// literal prefix), and there the one-launch kernel's match compaction -- 64 gathered rows per finish pass ACROSS tiles -- beats one
// finish pass per 8 KB tile (BASELINE config 2, one row in ten matching: 18.2 us against 19.5 us + 1.6 us of follow-up launch).
// FXAMD_NO_SPAN=1: the one-launch kernel (test / A-B hook); FXAMD_SPAN_LENS: bit mask of the row lengths it takes (1: 128, 2: 64,
// 4: 32, 8: 16 -- the LDS bytes a row gets: its length rounded up to those; +16: candidate-list driver programs at every length; +32: ragged
// rows, any length 2..127 that is not one of the four; +64: first pass of the multi-pass pipeline on the nibble tables; experiment hook).
static int span_cell(int64_t row_len) { return row_len <= 16 ? 16 : (row_len <= 32 ? 32 : (row_len <= 64 ? 64 : 128)); }   // bytes of LDS a row gets (fx_span.hpp: RL)
static bool span_kind(const FxpHeader& h, int scheme, int64_t row_len, bool spans) {
   if (!spans || scheme != 0 || h.mode != FXP_MODE_SEARCH_ENGINE || (h.flags & (FXP_F_RAW_BYTES | FXP_F_NEEDS_NONASCII | FXP_F_PREFIX_CHECK)) || fx_env().multipass || fx_env().no_span)
      return false;
   if (row_len < 2 || row_len > 128) return false;
   const int lens = fx_env().span_lens;
   if (row_len <= 64 && (h.flags & FXP_F_PREFILTER) && !(lens & 16)) return false;
   if (row_len != span_cell(row_len) && !(lens & 32)) return false;   // ragged rows (any other length: bit 5)
   const int rl = span_cell(row_len);
   return (rl == 128 && (lens & 1)) || (rl == 64 && (lens & 2)) || (rl == 32 && (lens & 4)) || (rl == 16 && (lens & 8));
}

// ---- the pipeline of one batch call, enqueued on `st` with the scratch set `sc` (p->mu held) ----------------------------------
// out_mode != 0: PACKED results (d_flags = bit words, d_from / d_to = narrow arrays of out_mode bytes per row).  Only the one-launch
// kernel writes them itself; for every other path the function returns FX_NOT_PACKED before anything is enqueued and the caller
// runs the unpacked pipeline into scratch and packs it with one more kernel.
static constexpr int FX_NOT_PACKED = 1;
// first_pass: FX_FP_OWN = the usual call; FX_FP_PREPARE = a shared first pass (fx_search_multi) is about to run for this pattern:
// flip the counter group, make the worklist, report what the shared kernel needs in *shared -- nothing is enqueued;
// FX_FP_DONE = the shared first pass has been enqueued: only this pattern's follow-up passes.
enum { FX_FP_OWN = 0, FX_FP_PREPARE = 1, FX_FP_DONE = 2 };
struct SharedFirstPass {
   uint32_t* ctr = nullptr;
   uint32_t* worklist = nullptr;
   uint32_t defer_tiles = 0;
   bool bytes_in_shared = false;   // the shared pass scans tiles with bytes >= 0x80 with this pattern's byte-level tables: no pass over deferred tiles
   bool exc_in_shared = false;     // ... and finishes the exception rows of those scans itself (class-level tables that decode): no follow-up at all
};
static int enqueue_batch_body(fxamd_program* p, const uint8_t* d_blob, DevScratch* sc, const uint8_t* d_rows, int64_t n, int64_t row_len,
                              uint8_t* d_flags, int32_t* d_from, int32_t* d_to, hipStream_t st, uint32_t out_mode, int first_pass, SharedFirstPass* shared,
                              const bool capturing_now);
// (a kernel node, not a memset node: a 16-byte hipMemsetAsync captured into a graph aborted the process at the first replay on this ROCm -- gpurun call r06_c9)
__global__ __launch_bounds__(64) void fx_zero_words(uint32_t* __restrict__ p, uint32_t n) {
   if (threadIdx.x < n) p[threadIdx.x] = 0u;
}
// A stream that is being captured into a hipGraph (round 6): the multi-pass pipelines keep their "tiles were deferred / rows are listed" words in two
// groups that alternate between calls on the HOST, each first pass zeroing the other group for the call after it -- a replayed graph alternates nothing.
// Under capture the call therefore zeroes ITS group with a one-block kernel node before its first pass and both groups behind its last pass: every replay starts
// and ends with clean words, whatever ran on the stream in between, and eager calls after it find the group they expect to be zero.  Pipelines of
// several launches (rows longer than 256 bytes, which have no one-launch kernel) are thereby replay-safe; rows of up to 256 bytes still take the
// one-launch kernel under capture (one node instead of two or three).
static int enqueue_batch(fxamd_program* p, const uint8_t* d_blob, DevScratch* sc, const uint8_t* d_rows, int64_t n, int64_t row_len,
                         uint8_t* d_flags, int32_t* d_from, int32_t* d_to, hipStream_t st, uint32_t out_mode = 0u, int first_pass = FX_FP_OWN,
                         SharedFirstPass* shared = nullptr) {
   bool cap_now = false;
   {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
      else cap_now = cap != hipStreamCaptureStatusNone;
   }
   const int rc = enqueue_batch_body(p, d_blob, sc, d_rows, n, row_len, d_flags, d_from, d_to, st, out_mode, first_pass, shared, cap_now);
   if (cap_now && rc == FXAMD_OK && first_pass != FX_FP_PREPARE && sc->d_counter) {
      hipLaunchKernelGGL(fx_zero_words, dim3(1), dim3(64), 0, st, sc->d_counter, 8u);
      FX_HIP(hipGetLastError());
   }
   return rc;
}
static int enqueue_batch_body(fxamd_program* p, const uint8_t* d_blob, DevScratch* sc, const uint8_t* d_rows, int64_t n, int64_t row_len,
                              uint8_t* d_flags, int32_t* d_from, int32_t* d_to, hipStream_t st, uint32_t out_mode, int first_pass, SharedFirstPass* shared,
                              const bool capturing_now) {
   const FxpHeader& h = p->prog.hdr();
   if (out_mode != 0u) {
      // who writes packed results itself: the one-launch kernel; round 5: the half-row first pass of 256-byte rows (8-state tables, spans) and
      // the span kernel, each with its follow-up (the first pass leaves a byte per deferred tile where the unpacked form marks the rows' flags)
      const int sc0 = (h.flags & FXP_F_NFA_SIM) ? -1 : fast_scheme(h, d_rows, row_len);
      const bool spans_p = d_from != nullptr && d_to != nullptr;
      const bool half0 = sc0 == 0 && h.mode == FXP_MODE_SEARCH_ENGINE && spans_p && row_len == 256 && half_rows(h, sc0, row_len) && scheme_decodes_utf8(h, sc0) &&
                         first_pass == FX_FP_OWN && !fx_env().no_pack_first;
      const bool one = sc0 >= 0 && (h.mode == FXP_MODE_SEARCH_ENGINE || h.mode == FXP_MODE_MATCH_ENGINE) && !(h.flags & FXP_F_RAW_BYTES) && !long_row(row_len) &&
                       (h.mode == FXP_MODE_MATCH_ENGINE ? (!match_half_rows(h, sc0, row_len) && scheme_decodes_utf8(h, sc0)) : (!half_rows(h, sc0, row_len) || half0)) &&
                       !fx_env().multipass;   // (`.match.` programs that cannot decode: the multi-pass pipeline, see gen_match below)
      if (!one) return FX_NOT_PACKED;
      // (FXAMD_NO_PACK_FIRST=1, test / A-B hook: 256-byte rows unpacked + fx_pack, the span kernel's rows packed by the one-launch kernel -- as before round 5)
   }
   const unsigned gblocks = (unsigned)((n + 255) / 256);
   const bool aligned16 = (reinterpret_cast<uintptr_t>(d_rows) & 15u) == 0 && (row_len & 15) == 0 && row_len > 0;
   if (h.flags & FXP_F_NFA_SIM) {
      // bounded scratch: rows are processed in chunks that share one scratch area (allocated on first use)
      const size_t per_row = (size_t)2 * h.nfa_words * 4;
      size_t chunk = (size_t(256) << 20) / per_row;
      if (chunk > 65536) chunk = 65536;
      if (chunk < 64) chunk = 64;
      chunk &= ~size_t(63);
      if (sc->nfa_scratch_rows < chunk) {
         if (sc->d_nfa_scratch) (void)hipFree(sc->d_nfa_scratch);
         sc->d_nfa_scratch = nullptr;
         sc->nfa_scratch_rows = 0;
         FX_HIP(hipMalloc((void**)&sc->d_nfa_scratch, chunk * per_row));
         sc->nfa_scratch_rows = chunk;
      }
      for (int64_t b0 = 0; b0 < n; b0 += (int64_t)chunk) {
         const int64_t cnt = n - b0 < (int64_t)chunk ? n - b0 : (int64_t)chunk;
         hipLaunchKernelGGL(fx_nfa, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st, d_rows, b0, b0 + cnt, (int32_t)row_len, d_blob, d_flags,
                            d_from, d_to, sc->d_nfa_scratch);
         FX_HIP(hipGetLastError());
      }
      p->last_path = 4;
      return FXAMD_OK;
   }
   const uint32_t prog_lds = h.total_bytes <= 32768u ? h.total_bytes : 0u;   // tables in LDS when they fit comfortably
   if (h.mode == FXP_MODE_MATCH_ENGINE) {   // `.match.` has no span: from/to stay untouched
      d_from = nullptr;
      d_to = nullptr;
   }
   const int scheme = fast_scheme(h, d_rows, row_len);
   if (scheme >= 0) {
      const bool is_match = h.mode == FXP_MODE_MATCH_ENGINE;
      // The counter groups alternate between calls, and it is a call's FIRST-PASS kernel that zeroes the other group for the call
      // after it: so the group flips only when such a kernel runs -- not for the one-launch kernel, which uses no counters (a
      // handle that alternates between the two pipelines would otherwise meet the stale counts of its last multi-pass call).
      // `.match.` over tiny rows (2 to 32 bytes) on the class-level v_perm / nibble tables: a lane takes a span of 64 / L whole
      // rows (fx_tiny.hpp); a first pass of the multi-pass kind: rows with bytes >= 0x80 are listed for the row-level fix-up
      // ... and the `.in.` VERDICT (no spans asked for) over the same rows: fx_search_tiny
      // (a stream that is being captured into a hipGraph keeps the one-launch kernel wherever another pipeline would hold host-side state between
      //  launches -- the counter groups alternate on the host.  Asked ONCE per call, and only when a rule below can need it: ADVICE r05.)
      auto capturing = [&]() -> bool { return capturing_now; };   // (asked once per call, by enqueue_batch)
      bool tiny = first_pass == FX_FP_OWN && (is_match || (h.mode == FXP_MODE_SEARCH_ENGINE && d_from == nullptr && !(h.flags & FXP_F_RAW_BYTES))) && out_mode == 0u &&
                  row_len >= 2 && row_len <= 32 && (scheme == 0 || scheme == 2) && !fx_env().multipass && !fx_env().no_tiny && !(h.flags & FXP_F_PREFIX_CHECK);
      if (tiny && capturing()) tiny = false;
      // Where round 4 moved rows off the one-launch kernel -- 256-byte rows on the chain / nibble tables (half rows) -- a stream that is
      // being captured into a hipGraph keeps the one-launch kernel: it holds no host-side state between launches (the counter groups of
      // the multi-pass pipelines alternate on the host).
      // (Rows of 129..255 bytes on the chain tables walked like long rows, in 128-byte segments, were tried too: the 17-state pattern over
      //  200-byte rows 1.283 -> 1.349 ms -- a 128-byte and a 72-byte segment pay two segments' fixed work; gpurun call r04_c34.)
      bool half = is_match ? match_half_rows(h, scheme, row_len) : half_rows(h, scheme, row_len);
      if (half && (scheme != 0 || is_match) && capturing()) half = false;
      const bool as_long = long_row(row_len);
      // Rows of 2..128 bytes with spans on the 8-state tables (round 5, fx_span.hpp; BASELINE config 5): a lane owns a 128-byte span of
      // 128 / RL whole rows -- the half-row kernel's memory path and four waves per SIMD -- first pass + ONE gated follow-up over the tiles it
      // marked (last_path 18; the counter groups alternate on the host, so a stream under hipGraph capture keeps the one-launch kernel); for
      // candidate-list driver programs the follow-up is the GEN instantiation (the general row procedure for the rows the tables cannot answer;
      // still 18).  FXAMD_NO_SPAN=1: the one-launch kernel (test / A-B hook).
      bool span = first_pass == FX_FP_OWN && (out_mode == 0u || (out_mode == 1u && !fx_env().no_pack_first)) && !half && !tiny &&
                  span_kind(h, scheme, row_len, d_from != nullptr && d_to != nullptr);
      if (span && capturing()) span = false;   // (a stream under hipGraph capture keeps the one-launch kernel: the counter groups alternate on the host)
      // ... and as the FIRST PASS of the multi-pass pipeline for automata of 9..16 states (round 5): aligned rows of 128 / 64 / 32 / 16 bytes on
      // the nibble tables; the tiles it marks go to the passes that pipeline has (byte-level tables over marked tiles / the decode pass), so it
      // needs one of the two.  last_path 20.  Measured (gpurun calls r05_c14 / c15, FXAMD_NO_SPAN=1 as the other arm): 16-byte rows 0.862 -> 0.575 ms,
      // 128-byte rows 0.3225 -> 0.3177 ms.  (The chain tables at 128-byte rows: 0.504 against 0.494 ms on round 4's 64-byte halves -- not built.)
      bool span_first = false;
      if (first_pass == FX_FP_OWN && out_mode == 0u && !is_match && !tiny && !span && h.mode == FXP_MODE_SEARCH_ENGINE && d_from != nullptr && d_to != nullptr &&
          !(h.flags & (FXP_F_RAW_BYTES | FXP_F_NEEDS_NONASCII | FXP_F_PREFIX_CHECK)) && !fx_env().multipass && !fx_env().no_span && (fx_env().span_lens & 64) &&
          scheme == 2 && row_len >= 2 && row_len <= 128 &&
          // (round 6: ragged rows too -- character(20), (80), (100) with a 9..16-state pattern used to stay on the one-launch kernel's RAGGED instantiations; the
          //  tiles the first pass marks then need the decode pass: byte-level tables do not run on the multi-pass kernels' padded ragged rows)
          ((row_len == span_cell(row_len) && (scheme_decodes_utf8(h, scheme) || bytes_ok(h, d_rows, row_len))) ||
           (row_len != span_cell(row_len) && (fx_env().span_lens & 32) && scheme_decodes_utf8(h, scheme)))) {
         span_first = !capturing();
      }
      // (`.match.` programs whose class-level tables cannot decode UTF-8 -- more than 126 symbol classes -- take the multi-pass pipeline: first pass + the general row
      //  procedure over a worklist; their one-launch instantiations, GEN x MATCH, were 117 kernels nobody's tables asked for: retired in round 6)
      const bool gen_match = is_match && !scheme_decodes_utf8(h, scheme);
      const bool one_launch = first_pass == FX_FP_OWN && !tiny && !(h.flags & FXP_F_RAW_BYTES) && !as_long && !half && !span && !span_first && !fx_env().multipass && !gen_match;
      if (first_pass != FX_FP_DONE && !one_launch) sc->parity ^= 1u;
      uint32_t* ctr = sc->d_counter + 4u * sc->parity;   // this call's words: [0] tiles deferred, [1] exception rows left
      if (first_pass == FX_FP_DONE && shared && shared->ctr) ctr = shared->ctr;   // (what PREPARE chose and the shared kernel used)
      if (capturing_now && !one_launch && first_pass == FX_FP_OWN) {   // (a replay cannot rely on the call before it: see enqueue_batch)
         hipLaunchKernelGGL(fx_zero_words, dim3(1), dim3(64), 0, st, ctr, 4u);
         FX_HIP(hipGetLastError());
      }
      // (long rows have no in-LDS decode pass: their non-ASCII / exception rows go to the row-level fix-up)
      const bool utf8_tables = scheme_decodes_utf8(h, scheme) && !as_long;
      const bool bytes = !(h.flags & FXP_F_RAW_BYTES) && bytes_ok(h, d_rows, row_len);
      const int bsch = bytes_scheme(h);
      const int big = scheme == 0 ? 0 : 4;   // last_path: 1 / 3 with the 8-state tables, 5 / 6 with the wide v_perm or chain tables
      PassOpts first, marked, listp;
      first.defer_tiles = (utf8_tables || bytes) ? 1u : 0u;
      // 256-byte rows on the 8-state tables keep the multi-pass pipeline: its first pass stages HALF rows when spans are asked for
      // (8 KB of LDS per wave: three waves per SIMD), which the one-launch kernel -- a full row per lane in LDS -- cannot
      const bool keep_multipass = !is_match && half;
      // (flags only: whole rows sit on the memory path with the v_perm and nibble tables; the chain tables' dependent LDS read per byte is
      //  latency-bound and gains from the four waves per SIMD of the half-row tile there too)
      first.half = (keep_multipass && half_staging(scheme, d_from != nullptr)) || (is_match && half);
      if (one_launch) {
         // ONE launch: every tile finished by the wave that staged it (class-level tables on pure-ASCII tiles, byte-level tables or
         // the in-LDS decode on the others, exception rows through per-wave queues -- decoded in LDS, or, for programs whose tables
         // cannot decode, through the general row procedure); last_path 9 / 10 / 11 (12 / 13 / 14: general procedure for the queued rows)
         const int ob = one_bytes_scheme(h, d_rows, row_len, scheme);
         const bool gen = !utf8_tables;
         FX_HIP(launch_one_any(scheme, ob, gen, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, st, out_mode));
         p->last_path = (ob == 0 ? 9 : (scheme == 0 ? 10 : 11)) + (gen ? 3 : 0);
         return FXAMD_OK;
      }
      if (span) {
         FastParams fps = params_of(h, 0, false);
         fps.defer_tiles = (fx_env().no_adapt || !utf8_tables) ? 1u : 3u;   // (bit 1: FX_ADAPT_CALLS -- batches that are mostly UTF-8 skip the first pass's loads)
         fps.out_mode = out_mode;
         if ((h.flags & FXP_F_R_LATCH) && !fx_env().no_latch) latch_params(fps, h);   // (R of <= 4 states: its latched format, round 6)
         uint8_t* marks = nullptr;
         if (out_mode != 0u) {   // packed results: a byte per 64-row tile says "left to the follow-up" (in the worklist's memory: this pipeline lists no rows)
            const int rcw = grow_worklist(sc, (n >> 8) + 64);
            if (rcw != FXAMD_OK) return rcw;
            marks = reinterpret_cast<uint8_t*>(sc->d_worklist);
         }
         const bool gen = !utf8_tables;
         int ob = one_bytes_scheme(h, d_rows, row_len, scheme);
         if (gen && ob == 3) ob = 2;   // (the v_perm forward automaton rides with the speculative pass: programs that decode; the nibble tables exist whenever it does)
         const FastParams fpc = params_of(h, 0, false), fpb = ob != 0 ? params_of(h, ob, true) : FastParams{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
         const uint32_t cmb = (1024u + h.n_pages * 64u) * 2u;
         const uint32_t tb = ob == 1 ? ((512u + h.byte_TR_bytes + h.byte_TA_bytes + 15u) & ~15u) : 0u;
#define FX_MARKED_G(CH, G)                                                                                                            \
   {                                                                                                                                  \
      if (ob == 2) FX_HIP((launch_one_marked<CH, 2, G>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, (uint32_t)row_len)));       \
      else if (ob == 1) FX_HIP((launch_one_marked<CH, 1, G>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, (uint32_t)row_len)));  \
      else FX_HIP((launch_one_marked<CH, 0, G>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, (uint32_t)row_len)));               \
   }
#define FX_SPAN_CASE(RL, CH)                                                                                                          \
   case RL:                                                                                                                           \
      FX_HIP((launch_span<RL, 0>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, st, marks, (uint32_t)row_len)));                 \
      if (gen) FX_MARKED_G(CH, true)                                                                                                  \
      else if (ob == 3) FX_HIP((launch_one_marked<CH, 3, false>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, (uint32_t)row_len))); \
      else FX_MARKED_G(CH, false)                                                                                                     \
      break;
         switch (span_cell(row_len)) {
            FX_SPAN_CASE(128, 8)
            FX_SPAN_CASE(64, 4)
            FX_SPAN_CASE(32, 2)
            FX_SPAN_CASE(16, 1)
            default: return FXAMD_E_ARG;
         }
#undef FX_SPAN_CASE
#undef FX_MARKED_G
         p->last_path = 18;
         return FXAMD_OK;
      }
      // worklist of the rows the tile kernels cannot answer: structurally invalid or non-canonical UTF-8 for the byte-level tables;
      // every row with a byte >= 0x80, and overlap rows, when there are no decode tables at all
      const bool marked_followup = first_pass == FX_FP_OWN && keep_multipass && scheme == 0 && row_len == 256 && utf8_tables && !fx_env().multipass;
      if ((bytes || !utf8_tables || first_pass != FX_FP_OWN) && !marked_followup) {
         const int rc = grow_worklist(sc, n);
         if (rc != FXAMD_OK) return rc;
         first.worklist = marked.worklist = listp.worklist = sc->d_worklist;
         listp.gate_word = 1;
         listp.grid_tiles = (n + 63) >> 6;
      }
      if (first_pass == FX_FP_PREPARE) {
         shared->ctr = ctr;
         shared->worklist = sc->d_worklist;
         shared->defer_tiles = first.defer_tiles;
         return FXAMD_OK;
      }
      // exception rows of a byte-level pass: the decode pass over the gathered worklist when the class-level tables can decode,
      // else the row-level fix-up through the general engine
      auto list_fixup = [&]() -> int {   // the general engine over the worklist
         if (aligned16 && (size_t)64 * row_len + prog_lds <= 65536u) {
            int64_t tb = (n + 63) / 64;
            if (tb > 16384) tb = 16384;
            hipLaunchKernelGGL(fx_fixup_list_tiled, dim3((unsigned)tb), dim3(64), (size_t)64 * row_len + prog_lds, st, d_rows, (int32_t)row_len, d_blob,
                               d_flags, d_from, d_to, sc->d_worklist, ctr + 1, prog_lds);
         } else {
            int64_t lblocks = (n + 255) / 256;
            if (lblocks > 4096) lblocks = 4096;
            hipLaunchKernelGGL(fx_fixup_list, dim3((unsigned)lblocks), dim3(256), prog_lds, st, d_rows, (int32_t)row_len, d_blob, d_flags, d_from, d_to,
                               sc->d_worklist, ctr + 1, prog_lds);
         }
         FX_HIP(hipGetLastError());
         return FXAMD_OK;
      };
      auto exceptions = [&]() -> int {
         if (!utf8_tables) return list_fixup();
         if (is_match) FX_HIP(match_by<4>(scheme, h, d_blob, d_rows, n, row_len, d_flags, ctr, st, listp));
         else FX_HIP(fast_by<4>(scheme, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, listp));
         return FXAMD_OK;
      };
      if (tiny) {
         const int rc = grow_worklist(sc, n);
         if (rc != FXAMD_OK) return rc;
         const FastParams fpt = params_of(h, scheme, false);
#define FX_TINY(LL)                                                                                                                    \
   FX_HIP(is_match ? (scheme == 0 ? (launch_tiny<LL, 0>(d_rows, n, d_blob, fpt, d_flags, ctr, sc->d_worklist, st))                     \
                                  : (launch_tiny<LL, 2>(d_rows, n, d_blob, fpt, d_flags, ctr, sc->d_worklist, st)))                    \
                   : (scheme == 0 ? (launch_tiny_search<LL, 0>(d_rows, n, d_blob, fpt, d_flags, ctr, sc->d_worklist, st))              \
                                  : (launch_tiny_search<LL, 2>(d_rows, n, d_blob, fpt, d_flags, ctr, sc->d_worklist, st))))
         switch (row_len) {
#define FX_TINY_CASE(LL) \
   case LL: FX_TINY(LL); break;
            FX_TINY_CASE(2) FX_TINY_CASE(3) FX_TINY_CASE(4) FX_TINY_CASE(5) FX_TINY_CASE(6) FX_TINY_CASE(7) FX_TINY_CASE(8) FX_TINY_CASE(9) FX_TINY_CASE(10)
            FX_TINY_CASE(11) FX_TINY_CASE(12) FX_TINY_CASE(13) FX_TINY_CASE(14) FX_TINY_CASE(15) FX_TINY_CASE(16) FX_TINY_CASE(17) FX_TINY_CASE(18)
            FX_TINY_CASE(19) FX_TINY_CASE(20) FX_TINY_CASE(21) FX_TINY_CASE(22) FX_TINY_CASE(23) FX_TINY_CASE(24) FX_TINY_CASE(25) FX_TINY_CASE(26)
            FX_TINY_CASE(27) FX_TINY_CASE(28) FX_TINY_CASE(29) FX_TINY_CASE(30) FX_TINY_CASE(31) FX_TINY_CASE(32)
#undef FX_TINY_CASE
            default: return FXAMD_E_ARG;
         }
#undef FX_TINY
         p->last_path = 17;
         return list_fixup();   // (gated on the list's count: empty on pure-ASCII batches)
      }
      if (h.flags & FXP_F_RAW_BYTES) {   // literal search over raw bytes: nothing is deferred
         if (first_pass == FX_FP_OWN) FX_HIP(fast_by<0>(scheme, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, first));
         p->last_path = 1 + big;
         return FXAMD_OK;
      }
      if (bytes && scheme != 0 && first_pass == FX_FP_OWN && !first.half && !span_first) {
         // no 8-state class-level tables to be faster with on ASCII: the byte-level tables take every tile, UTF-8 or not, in one pass
         if (is_match) FX_HIP(match_by<2>(bsch, h, d_blob, d_rows, n, row_len, d_flags, ctr, st, first));
         else FX_HIP(fast_by<2>(bsch, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, first));
         p->last_path = 7;
         return exceptions();
      }
      // 256-byte rows on the 8-state tables whose tables can decode: the first pass (half-row staging when spans are asked for) and
      // ONE gated follow-up -- the one-launch kernel over the tiles that pass marked (byte-level tables or in-LDS decode, exception
      // queues inside) -- instead of a pass over marked tiles plus a pass over a worklist
      if (marked_followup) {
         if (!fx_env().no_adapt) first.defer_tiles |= 2u;   // (FX_ADAPT_CALLS: batches that are mostly UTF-8 skip the first pass's loads)
         uint8_t* marks = nullptr;
         if (out_mode != 0u) {   // packed results from the half-row first pass (round 5): a byte per tile says "left to the follow-up"
            const int rcw = grow_worklist(sc, (n >> 8) + 64);
            if (rcw != FXAMD_OK) return rcw;
            marks = reinterpret_cast<uint8_t*>(sc->d_worklist);
            first.worklist = sc->d_worklist;
            first.out_mode = out_mode;
         }
         FX_HIP(fast_by<0>(scheme, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, first));
         const int ob = one_bytes_scheme(h, d_rows, row_len, scheme);
         const FastParams fpc = params_of(h, 0, false), fpb = ob != 0 ? params_of(h, ob, true) : FastParams{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
         const uint32_t cmb = (1024u + h.n_pages * 64u) * 2u;
         const uint32_t tb = ob == 1 ? ((512u + h.byte_TR_bytes + h.byte_TA_bytes + 15u) & ~15u) : 0u;
#define FX_MARKED(CH)                                                                                                              \
   {                                                                                                                                  \
      if (ob == 3) FX_HIP((launch_one_marked<CH, 3, false>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, 16u * CH)));          \
      else if (ob == 2) FX_HIP((launch_one_marked<CH, 2, false>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, 16u * CH)));     \
      else if (ob == 1) FX_HIP((launch_one_marked<CH, 1, false>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, 16u * CH)));     \
      else FX_HIP((launch_one_marked<CH, 0, false>(d_rows, n, d_blob, fpc, fpb, d_flags, d_from, d_to, cmb, tb, st, ctr, out_mode, marks, 16u * CH)));                  \
   }
         // (what the gated launch costs a step on a pure-ASCII batch, measured with the launch compiled out: config 3 0.4519 / 0.4618 / 0.4626 ms with it,
         //  0.4546 / 0.4608 / 0.4620 ms without -- nothing; at 1.25 M rows, an eighth of the batch: 66.4 against 63.5-64.7 us; gpurun call r05_c28.  Under
         //  rocprofv3 the same launch shows as 4.7 us + a 10 us longer first pass: the profiler serialises the dispatches.)
         FX_MARKED(16)
#undef FX_MARKED
         p->last_path = 16;
         return FXAMD_OK;
      }
      // first pass with the class-level tables: pure-ASCII tiles are finished here
      if (first_pass == FX_FP_DONE) {
         // (done by the shared kernel)
      } else if (is_match) FX_HIP(match_by<0>(scheme, h, d_blob, d_rows, n, row_len, d_flags, ctr, st, first));
      else if (span_first) {
         FastParams fps = params_of(h, scheme, false);
         fps.defer_tiles = 1u;
         const int rl = span_cell(row_len);   // (the LDS bytes a row gets: its length rounded up to 16 / 32 / 64 / 128)
         if (rl == 128) FX_HIP((launch_span<128, 2>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, st, nullptr, (uint32_t)row_len)));
         else if (rl == 64) FX_HIP((launch_span<64, 2>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, st, nullptr, (uint32_t)row_len)));
         else if (rl == 32) FX_HIP((launch_span<32, 2>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, st, nullptr, (uint32_t)row_len)));
         else FX_HIP((launch_span<16, 2>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, st, nullptr, (uint32_t)row_len)));
      } else FX_HIP(fast_by<0>(scheme, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, first));
      if (bytes) {
         // deferred tiles (bytes >= 0x80): byte-level tables on the raw bytes; structurally invalid rows go on to the decode pass
         // (a shared first pass that scanned those tiles itself has deferred none: only its exception rows are left)
         if (first_pass == FX_FP_DONE && shared && shared->bytes_in_shared) {
            if (shared->exc_in_shared) {   // nothing was deferred, nothing was listed
               p->last_path = 8;
               return FXAMD_OK;
            }
         } else if (is_match) FX_HIP(match_by<3>(bsch, h, d_blob, d_rows, n, row_len, d_flags, ctr, st, marked));
         else FX_HIP(fast_by<3>(bsch, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, marked));
         p->last_path = span_first ? 20 : 8;
         return exceptions();
      }
      if (utf8_tables) {
         // deferred tiles: the decode pass rewrites UTF-8 to symbol ids in LDS and scans only those tiles
         if (is_match) FX_HIP(match_by<1>(scheme, h, d_blob, d_rows, n, row_len, d_flags, ctr, st, marked));
         else FX_HIP(fast_by<1>(scheme, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, st, marked));
         p->last_path = span_first ? 20 : 1 + big;
         return FXAMD_OK;
      }
      // rows holding bytes >= 0x80 (and overlap rows) were listed one by one: row-level fix-up over that list
      p->last_path = 3 + big - (big ? 1 : 0);
      return list_fixup();
   }
   if (aligned16 && (size_t)64 * row_len + prog_lds <= 65536u) {   // (64 KB: the dynamic LDS a launch gets without opting in to more)
      const unsigned tblocks = (unsigned)((n + 63) / 64);
      hipLaunchKernelGGL(fx_general_tiled, dim3(tblocks), dim3(64), (size_t)64 * row_len + prog_lds, st, d_rows, n, (int32_t)row_len, d_blob,
                         d_flags, d_from, d_to, prog_lds);
   } else {
      hipLaunchKernelGGL(fx_general, dim3(gblocks), dim3(256), prog_lds, st, d_rows, n, (int32_t)row_len, d_blob, d_flags, d_from, d_to, 0, prog_lds);
   }
   FX_HIP(hipGetLastError());
   p->last_path = 2;
   return FXAMD_OK;
}

// (p->mu held) what trim_scratch could free
static size_t held_scratch(const fxamd_program* p) {
   size_t held = 0;
   for (const DevScratch& s : p->scratch)
      held += (s.worklist_rows * 4 > (int64_t(1) << 20) ? (size_t)s.worklist_rows * 4 : 0) +   // (a worklist of up to 1 MB is kept: see trim_scratch)
              (size_t)s.unpacked_rows * 9 + s.nfa_scratch_rows * (size_t)2 * p->prog.hdr().nfa_words * 4;
   return held;
}
static void trim_scratch(fxamd_program* p, size_t budget) {
   std::lock_guard<std::mutex> g(p->mu);
   const size_t held = held_scratch(p);
   p->held.store(held);
   if (held <= budget) return;
   for (DevScratch& s : p->scratch) {
      if (s.d_worklist && s.worklist_rows * 4 > (int64_t(1) << 20)) {
         (void)hipFree(s.d_worklist);
         s.d_worklist = nullptr;
         s.worklist_rows = 0;
      }
      if (s.d_unpacked) {
         (void)hipFree(s.d_unpacked);
         s.d_unpacked = nullptr;
         s.unpacked_rows = 0;
      }
      if (s.d_nfa_scratch) {
         (void)hipFree(s.d_nfa_scratch);
         s.d_nfa_scratch = nullptr;
         s.nfa_scratch_rows = 0;
      }
   }
   p->held.store(held_scratch(p));
}
static void trim_scratch(fxamd_program* p) { trim_scratch(p, FX_TRIM_BUDGET); }
// Trim idle cached programs, least recently used first, until the cache as a whole keeps at most `keep_bytes`; returns the bytes freed.
// Victims are pinned under the cache lock and trimmed outside it (a trim takes the program's own lock and calls hipFree).
static size_t trim_cache(size_t keep_bytes) {
   std::vector<fxamd_program*> victims;
   {
      std::lock_guard<std::mutex> g(g_cache_mu);
      size_t total = 0;
      for (fxamd_program* q : g_cache) total += q->held.load();
      if (total <= keep_bytes) return 0;
      for (fxamd_program* q : g_cache) {   // (most recently used last)
         if (total <= keep_bytes) break;
         const size_t h = q->held.load();
         if (q->refs != 1 || h == 0) continue;   // in use by a caller, or nothing to free
         ++q->refs;
         victims.push_back(q);
         total -= h;
      }
   }
   size_t freed = 0;
   for (fxamd_program* q : victims) {
      const size_t before = q->held.load();
      trim_scratch(q, 0);
      freed += before - std::min(before, q->held.load());
      bool dead = false;
      {
         std::lock_guard<std::mutex> g(g_cache_mu);
         dead = --q->refs == 0;
      }
      if (dead) destroy_program(q);
   }
   return freed;
}
static void destroy_program(fxamd_program* p) {
   for (DevBlob& b : p->blobs)
      if (b.d_blob) (void)hipFree(b.d_blob);
   for (DevScratch& s : p->scratch) free_scratch(s);
   delete p;
}

#pragma GCC visibility push(default)   // the library is built with -fvisibility=hidden: only the C ABI is exported
extern "C" {

int fxamd_last_hip_error(void) { return g_last_hip_error; }
void fxamd_reload_env(void) {   // tests only: not synchronised with calls in flight on other threads
   (void)fx_env();
   env_load();
}
/* Free the device scratch (work lists, unpacked staging, NFA bitsets) of every idle program of the compile cache; returns the bytes
 * freed.  The library trims by itself once the cached programs together keep more than 1 GB; a host that shares the GPU with another
 * allocator calls this when it wants the memory back now. */
int64_t fxamd_cache_trim(void) { return (int64_t)trim_cache(0); }
int fxamd_device_count(void) {
   int c = 0;
   if (hipGetDeviceCount(&c) != hipSuccess) return 0;
   return c;
}

int fxamd_compile(const char* pattern, int64_t pattern_len, int op, fxamd_program** out, int32_t* status) {
   if (!out || pattern_len < 0 || (!pattern && pattern_len > 0) || (op != FXAMD_OP_SEARCH && op != FXAMD_OP_MATCH)) return FXAMD_E_ARG;
   fxamd_program* p = nullptr;
   bool inserted = false, from_cache = false;   // p sits in g_cache / was handed out by it: then it is not ours to delete
   std::vector<fxamd_program*> evicted;
   try {   // nothing may cross the C boundary: the library never aborts the process
      evicted.reserve(2);   // (one insertion evicts at most one entry: the push_back below cannot throw)
      const bool cached = pattern_len <= 4096 && !fx_env().no_cache;
      std::string key;
      if (cached) {
         key.assign(1, (char)('0' + op));
         key.append(pattern ? pattern : "", (size_t)pattern_len);
         std::lock_guard<std::mutex> g(g_cache_mu);
         for (size_t i = g_cache.size(); i-- > 0;)
            if (g_cache[i]->cache_key == key) {
               p = g_cache[i];
               g_cache.erase(g_cache.begin() + (long)i);
               g_cache.push_back(p);
               ++p->refs;
               from_cache = true;
               break;
            }
      }
      if (!p) {
         p = new fxamd_program();
         p->prog = fxc::compile(std::string(pattern ? pattern : "", (size_t)pattern_len), op);
         if (cached && p->prog.blob.size() <= (size_t(1) << 20)) {
            std::lock_guard<std::mutex> g(g_cache_mu);
            g_cache.reserve(g_cache.size() + 1);   // (may throw: nothing has been changed yet)
            p->cache_key = key;                    // (may throw: p is not in the cache yet, `inserted` still false)
            g_cache.push_back(p);                  // (cannot throw after the reserve)
            ++p->refs;
            inserted = true;
            while (g_cache.size() > FX_CACHE_MAX) {
               fxamd_program* old = g_cache.front();
               g_cache.erase(g_cache.begin());
               old->cache_key.clear();
               if (--old->refs == 0) evicted.push_back(old);
            }
         }
      }
   } catch (const std::bad_alloc&) {
      if (p && !inserted && !from_cache) delete p;
      for (fxamd_program* e : evicted) destroy_program(e);
      return FXAMD_E_NOMEM;
   } catch (...) {
      if (p && !inserted && !from_cache) delete p;
      for (fxamd_program* e : evicted) destroy_program(e);
      return FXAMD_E_ARG;
   }
   for (fxamd_program* e : evicted) destroy_program(e);
   if (status) *status = p->prog.status;
   *out = p;
   return FXAMD_OK;
}

int fxamd_compile_nfa(int32_t n_states, int32_t entry, int32_t exit_state, int64_t n_transitions, const int32_t* src, const int32_t* dst,
                      const int64_t* seg_begin, const int32_t* seg_min, const int32_t* seg_max, const char* lit_all, int64_t len_all,
                      const char* lit_prefix, int64_t len_prefix, const char* lit_suffix, int64_t len_suffix, int op, fxamd_program** out,
                      int32_t* status) {
   if (!out || n_states < 2 || entry < 1 || entry > n_states || exit_state < 1 || exit_state > n_states || n_transitions < 0) return FXAMD_E_ARG;
   if (n_transitions > 0 && (!src || !dst || !seg_begin || !seg_min || !seg_max)) return FXAMD_E_ARG;
   if (op != FXAMD_OP_SEARCH && op != FXAMD_OP_MATCH) return FXAMD_E_ARG;
   if (len_all < 0 || len_prefix < 0 || len_suffix < 0) return FXAMD_E_ARG;
   if (n_transitions > 0 && seg_begin[0] < 0) return FXAMD_E_ARG;
   fxamd_program* p = nullptr;
   try {
      fxfe::Nfa nfa;
      nfa.nfa_top = n_states;
      nfa.entry = entry;
      nfa.exit = exit_state;
      nfa.nodes.resize((size_t)n_states + 1);
      for (int64_t t = 0; t < n_transitions; ++t) {
         if (src[t] < 1 || src[t] > n_states || dst[t] < 1 || dst[t] > n_states || seg_begin[t + 1] < seg_begin[t]) return FXAMD_E_ARG;
         fxfe::NfaTransition tr;
         tr.dst = dst[t];
         for (int64_t k = seg_begin[t]; k < seg_begin[t + 1]; ++k) {
            // a segment is a code-point range min <= max, or one of the reference's markers with min == max (SEG_EPSILON (-1,-1),
            // the unused SEG_INIT slots above the code space)
            if (seg_min[k] > seg_max[k]) return FXAMD_E_ARG;
            tr.c.emplace_back(seg_min[k], seg_max[k]);
         }
         tr.c_top = (int)tr.c.size();
         nfa.nodes[(size_t)src[t]].forward.push_back(tr);
      }
      fxfe::Literals lit;
      lit.all.assign(lit_all ? lit_all : "", (size_t)(lit_all ? len_all : 0));
      lit.prefix.assign(lit_prefix ? lit_prefix : "", (size_t)(lit_prefix ? len_prefix : 0));
      lit.suffix.assign(lit_suffix ? lit_suffix : "", (size_t)(lit_suffix ? len_suffix : 0));
      p = new fxamd_program();
      if (op == FXAMD_OP_SEARCH && !fxfe::f_eq(lit.all, ""))
         p->prog = fxc::make_search_literal(lit.all);   // whole-pattern literal: raw-byte INDEX path (forgex.F90:111-130)
      else
         p->prog = fxc::compile_from_nfa(nfa, lit, op);
   } catch (const std::bad_alloc&) {
      delete p;
      return FXAMD_E_NOMEM;
   } catch (...) {
      delete p;
      return FXAMD_E_ARG;
   }
   if (status) *status = p->prog.status;
   *out = p;
   return FXAMD_OK;
}

void fxamd_program_free(fxamd_program* p) {
   if (!p) return;
   release_program(p);
}
int32_t fxamd_program_status(const fxamd_program* p) { return p ? p->prog.status : FXAMD_E_ARG; }
int64_t fxamd_program_blob_size(const fxamd_program* p) { return p ? (int64_t)p->prog.blob.size() : FXAMD_E_ARG; }
int fxamd_program_blob(const fxamd_program* p, void* buf, int64_t capacity) {
   if (!p || !buf || capacity < (int64_t)p->prog.blob.size()) return FXAMD_E_ARG;
   std::memcpy(buf, p->prog.blob.data(), p->prog.blob.size());
   return FXAMD_OK;
}
int fxamd_program_from_blob(const void* blob, int64_t size, fxamd_program** out) {
   if (!blob || !out || size < (int64_t)sizeof(FxpHeader)) return FXAMD_E_ARG;
   // images from outside are checked table by table (extent inside the image, index entries inside their tables) and by checksum
   if (fxc::validate_blob((const uint8_t*)blob, (size_t)size) != 0) return FXAMD_E_BLOB;
   fxamd_program* p = nullptr;
   try {
      p = new fxamd_program();
      p->prog.blob.assign((const uint8_t*)blob, (const uint8_t*)blob + size);
   } catch (...) {
      delete p;
      return FXAMD_E_NOMEM;
   }
   p->prog.status = (int)p->prog.hdr().status;
   *out = p;
   return FXAMD_OK;
}
int fxamd_program_info(const fxamd_program* p, int32_t* info) {
   if (!p || !info) return FXAMD_E_ARG;
   const FxpHeader& h = p->prog.hdr();
   info[0] = (int32_t)h.mode;
   info[1] = (int32_t)h.flags;
   info[2] = (int32_t)h.nA;
   info[3] = (int32_t)h.nR;
   info[4] = (int32_t)h.n_classes;
   info[5] = p->prog.status;
   info[6] = (int32_t)h.total_bytes;
   info[7] = (int32_t)h.n_bounds;
   return FXAMD_OK;
}
const char* fxamd_strerror(int32_t status) { return fxfe::status_message(status); }
int64_t fxamd_strerror_copy(int32_t status, char* buf, int64_t capacity) {
   if (!buf || capacity <= 0) return 0;
   const char* m = fxfe::status_message(status);
   int64_t n = (int64_t)std::strlen(m);
   if (n > capacity) n = capacity;
   std::memcpy(buf, m, (size_t)n);
   return n;
}

int fxamd_program_upload(fxamd_program* p) {
   if (!p) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(p->mu);
   int dev = -1;
   FX_HIP(hipGetDevice(&dev));
   uint8_t* d_blob = nullptr;
   return blob_for_device(p, dev, &d_blob);
}

int fxamd_program_reserve(fxamd_program* p, int64_t max_rows, void* hip_stream) {
   if (!p || max_rows < 0) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(p->mu);
   int dev = -1;
   FX_HIP(hipGetDevice(&dev));
   uint8_t* d_blob = nullptr;
   int rc = blob_for_device(p, dev, &d_blob);
   if (rc != FXAMD_OK) return rc;
   DevScratch* sc = nullptr;
   rc = scratch_for(p, dev, (hipStream_t)hip_stream, &sc);
   if (rc != FXAMD_OK) return rc;
   return grow_worklist(sc, max_rows);
}

int fxamd_last_path(const fxamd_program* p) {
   if (!p) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(const_cast<fxamd_program*>(p)->mu);
   return p->last_path;
}

int fxamd_launch_fast_only(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags, int32_t* d_from,
                           int32_t* d_to, void* hip_stream) {
   if (!p || n <= 0 || !d_rows || !d_flags || (d_from == nullptr) != (d_to == nullptr)) return FXAMD_E_ARG;
   if (p->prog.status != 0) return FXAMD_E_ARG;
   const FxpHeader& h = p->prog.hdr();
   const int scheme = fast_scheme(h, d_rows, row_len);
   if (scheme < 0 || h.mode == FXP_MODE_MATCH_ENGINE) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(p->mu);
   int dev = -1;
   FX_HIP(hipGetDevice(&dev));
   uint8_t* d_blob = nullptr;
   int rc = blob_for_device(p, dev, &d_blob);
   if (rc != FXAMD_OK) return rc;
   DevScratch* sc = nullptr;
   rc = scratch_for(p, dev, (hipStream_t)hip_stream, &sc);
   if (rc != FXAMD_OK) return rc;
   sc->parity ^= 1u;
   uint32_t* ctr = sc->d_counter + 4u * sc->parity;
   const bool bytes = !(h.flags & FXP_F_RAW_BYTES) && bytes_ok(h, d_rows, row_len);
   PassOpts po;
   po.defer_tiles = ((scheme_decodes_utf8(h, scheme) && !long_row(row_len)) || bytes) ? 1u : 0u;
   po.half = half_rows(h, scheme, row_len) && half_staging(scheme, d_from != nullptr);
   if (!po.half && n > 0 && span_kind(h, scheme, row_len, d_from != nullptr)) {   // the span kernel's first pass
      FastParams fps = params_of(h, 0, false);
      fps.defer_tiles = 1u;
      if ((h.flags & FXP_F_R_LATCH) && !fx_env().no_latch) latch_params(fps, h);   // (the instantiation the product's call launches)
      switch (span_cell(row_len)) {
         case 128: FX_HIP((launch_span<128, 0>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, (hipStream_t)hip_stream, nullptr, (uint32_t)row_len))); break;
         case 64: FX_HIP((launch_span<64, 0>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, (hipStream_t)hip_stream, nullptr, (uint32_t)row_len))); break;
         case 32: FX_HIP((launch_span<32, 0>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, (hipStream_t)hip_stream, nullptr, (uint32_t)row_len))); break;
         default: FX_HIP((launch_span<16, 0>(d_rows, n, d_blob, fps, d_flags, d_from, d_to, ctr, (hipStream_t)hip_stream, nullptr, (uint32_t)row_len))); break;
      }
      return FXAMD_OK;
   }
   if (scheme != 0 && bytes && !po.half && sc->worklist_rows >= n) {   // the dominant pass of these programs is the byte-level one over all tiles
      po.worklist = sc->d_worklist;
      FX_HIP(fast_by<2>(bytes_scheme(h), h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, (hipStream_t)hip_stream, po));
   } else {
      FX_HIP(fast_by<0>(scheme, h, d_blob, d_rows, n, row_len, d_flags, d_from, d_to, ctr, (hipStream_t)hip_stream, po));
   }
   return FXAMD_OK;
}

// Rows per enqueue: worklists, exception queues and gate counters hold 32-bit row numbers, so a batch of more rows than this (288 GB
// of HBM hold 2^34 sixteen-byte rows) is enqueued slice by slice on the same stream.  A multiple of 64: slices are whole tiles and whole
// words of the packed flags.  (FXAMD_SLICE_ROWS: test hook.)
static int64_t slice_rows() { return fx_env().slice_rows; }

int fxamd_match_batch_device(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags, int32_t* d_from,
                             int32_t* d_to, void* hip_stream) {
   if (!p || n < 0 || row_len < 0 || row_len > 0x3FFFFFFF || !d_flags || (n > 0 && row_len > 0 && !d_rows)) return FXAMD_E_ARG;
   if ((d_from == nullptr) != (d_to == nullptr)) return FXAMD_E_ARG;
   if (p->prog.status >= 100) return FXAMD_E_UNSUPPORTED;
   if (n == 0) return FXAMD_OK;
   if (n > slice_rows()) {
      for (int64_t o = 0; o < n; o += slice_rows()) {
         const int64_t m = std::min(slice_rows(), n - o);
         const int rc = fxamd_match_batch_device(p, d_rows ? d_rows + o * row_len : nullptr, m, row_len, d_flags + o, d_from ? d_from + o : nullptr,
                                                 d_to ? d_to + o : nullptr, hip_stream);
         if (rc != FXAMD_OK) return rc;
      }
      return FXAMD_OK;
   }
   hipStream_t st = (hipStream_t)hip_stream;
   const FxpHeader& h = p->prog.hdr();
   std::lock_guard<std::mutex> g(p->mu);
   if (h.mode == FXP_MODE_INVALID) {
      hipLaunchKernelGGL(fx_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_flags, d_from, d_to, n);
      FX_HIP(hipGetLastError());
      p->last_path = 0;
      return FXAMD_OK;
   }
   int dev = -1;
   FX_HIP(hipGetDevice(&dev));
   uint8_t* d_blob = nullptr;
   int rc = blob_for_device(p, dev, &d_blob);
   if (rc != FXAMD_OK) return rc;
   DevScratch* sc = nullptr;
   rc = scratch_for(p, dev, st, &sc);
   if (rc != FXAMD_OK) return rc;
   rc = enqueue_batch(p, d_blob, sc, d_rows, n, row_len, d_flags, d_from, d_to, st);
   p->held.store(held_scratch(p));
   return rc;
}

int fxamd_packed_layout(int64_t n, int64_t row_len, int with_spans, int64_t* off_from, int64_t* off_to, int64_t* total_bytes, int32_t* span_bytes) {
   if (n < 0 || row_len < 0) return FXAMD_E_ARG;
   const int32_t w = !with_spans ? 0 : (row_len <= 255 ? 1 : (row_len <= 65535 ? 2 : 4));
   const int64_t nb = (((n + 7) / 8) + 15) & ~int64_t(15);   // (>= 8 bytes per 64 rows: the kernels store whole 64-bit words)
   const int64_t ns = (n * w + 15) & ~int64_t(15);
   if (off_from) *off_from = nb;
   if (off_to) *off_to = nb + ns;
   if (total_bytes) *total_bytes = nb + 2 * ns;
   if (span_bytes) *span_bytes = w;
   return FXAMD_OK;
}

// one slice of a packed call: `d_words` = the slice's flag words, pf / pt = its narrow span arrays (w bytes per row, 0 = none)
static int packed_slice(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, int32_t w, uint8_t* d_words, uint8_t* pf, uint8_t* pt,
                        hipStream_t st);

int fxamd_match_batch_device_packed(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, int with_spans, uint8_t* d_packed,
                                    void* hip_stream) {
   if (!p || n < 0 || row_len < 0 || row_len > 0x3FFFFFFF || !d_packed || (n > 0 && row_len > 0 && !d_rows)) return FXAMD_E_ARG;
   if ((reinterpret_cast<uintptr_t>(d_packed) & 15u) != 0) return FXAMD_E_ARG;
   if (p->prog.status >= 100) return FXAMD_E_UNSUPPORTED;
   if (n == 0) return FXAMD_OK;
   hipStream_t st = (hipStream_t)hip_stream;
   const FxpHeader& h = p->prog.hdr();
   if (h.mode == FXP_MODE_MATCH_ENGINE) with_spans = 0;   // `.match.` has no span
   int64_t off_f = 0, off_t = 0, total = 0;
   int32_t w = 0;
   (void)fxamd_packed_layout(n, row_len, with_spans, &off_f, &off_t, &total, &w);
   if (h.mode == FXP_MODE_INVALID) {   // every row: no match, spans 0
      std::lock_guard<std::mutex> g(p->mu);
      FX_HIP(hipMemsetAsync(d_packed, 0, (size_t)total, st));
      p->last_path = 0;
      return FXAMD_OK;
   }
   for (int64_t o = 0; o < n; o += slice_rows()) {
      const int64_t m = std::min(slice_rows(), n - o);
      const int rc = packed_slice(p, d_rows ? d_rows + o * row_len : nullptr, m, row_len, w, d_packed + o / 8, d_packed + off_f + o * w, d_packed + off_t + o * w, st);
      if (rc != FXAMD_OK) return rc;
   }
   return FXAMD_OK;
}

static int packed_slice(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, int32_t w, uint8_t* d_packed, uint8_t* pf, uint8_t* pt,
                        hipStream_t st) {
   const FxpHeader& h = p->prog.hdr();
   std::lock_guard<std::mutex> g(p->mu);
   int dev = -1;
   FX_HIP(hipGetDevice(&dev));
   uint8_t* d_blob = nullptr;
   int rc = blob_for_device(p, dev, &d_blob);
   if (rc != FXAMD_OK) return rc;
   DevScratch* sc = nullptr;
   rc = scratch_for(p, dev, st, &sc);
   if (rc != FXAMD_OK) return rc;
   // in-kernel packing: the tile's ballot is the flag word, spans are stored narrow
   rc = enqueue_batch(p, d_blob, sc, d_rows, n, row_len, d_packed, w ? reinterpret_cast<int32_t*>(pf) : nullptr, w ? reinterpret_cast<int32_t*>(pt) : nullptr, st,
                      w ? (uint32_t)w : 1u);
   if (rc != FX_NOT_PACKED) return rc;
   // every other path: unpacked into the handle's scratch, then one packing kernel
   if (sc->unpacked_rows < n) {
      if (sc->d_unpacked) (void)hipFree(sc->d_unpacked);
      sc->d_unpacked = nullptr;
      sc->unpacked_rows = 0;
      FX_HIP(hipMalloc((void**)&sc->d_unpacked, (size_t)n * 9 + 64));
      sc->unpacked_rows = n;
      p->held.store(held_scratch(p));
   }
   int32_t* uf = reinterpret_cast<int32_t*>(sc->d_unpacked);
   int32_t* ut = uf + n;
   uint8_t* ufl = reinterpret_cast<uint8_t*>(ut + n);
   if (w && h.mode != FXP_MODE_MATCH_ENGINE) FX_HIP(hipMemsetAsync(sc->d_unpacked, 0, (size_t)n * 8, st));
   rc = enqueue_batch(p, d_blob, sc, d_rows, n, row_len, ufl, w ? uf : nullptr, w ? ut : nullptr, st, 0u);
   if (rc != FXAMD_OK) return rc;
   hipLaunchKernelGGL(fx_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ufl, uf, ut, n, reinterpret_cast<uint64_t*>(d_packed), pf, pt, (uint32_t)w);
   FX_HIP(hipGetLastError());
   return FXAMD_OK;
}

int fxamd_unpack_results(const uint8_t* d_packed, int64_t n, int64_t row_len, int with_spans, uint8_t* d_flags, int32_t* d_from, int32_t* d_to,
                         void* hip_stream) {
   if (!d_packed || n < 0 || !d_flags || (d_from == nullptr) != (d_to == nullptr)) return FXAMD_E_ARG;
   if (n == 0) return FXAMD_OK;
   int64_t off_f = 0, off_t = 0;
   int32_t w = 0;
   (void)fxamd_packed_layout(n, row_len, with_spans, &off_f, &off_t, nullptr, &w);
   hipLaunchKernelGGL(fx_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, reinterpret_cast<const uint64_t*>(d_packed),
                      d_packed + off_f, d_packed + off_t, n, d_flags, d_from, d_to, (uint32_t)w);
   FX_HIP(hipGetLastError());
   return FXAMD_OK;
}

// Side streams for the per-pattern follow-up passes of a shared first pass: each is a small, latency-bound launch (a gated pass over
// a pattern's exception rows), so the patterns' follow-ups run side by side -- forked from the caller's stream by an event, joined back
// by one event per pattern -- instead of one after the other (6 UTF-8 patterns on config 4's rows: 6 x ~40 us in a row otherwise).
// One set per (device, caller stream) -- ADVICE r03: with one set per device, callers on different streams were serialised on the host
// (one mutex around every fork / join) and coupled on the device (B's follow-ups queued behind A's on the same side stream).  A set's
// mutex is held while its fork / join is enqueued, so only callers that share a caller stream meet on it; the pool keeps at most
// FX_SIDE_SETS sets per process and re-keys the least recently used free one after that (its side streams may still hold the old
// caller's work: the new one's follow-ups queue behind it -- an ordering that is only stricter).
struct SideStreams {
   int device = -1;
   hipStream_t caller = nullptr;
   uint64_t stamp = 0;
   std::mutex mu;   // held while a fork / join is enqueued (the events are the set's)
   hipEvent_t fork = nullptr;
   std::vector<hipStream_t> streams;
   std::vector<hipEvent_t> joined;
};
constexpr size_t FX_SIDE_SETS = 16;
static std::mutex g_side_mu;   // the pool's list only
static std::vector<SideStreams*> g_side;
static uint64_t g_side_clock = 0;
// the set of (dev, caller), locked (`lock` owns its mutex on success), with at least k side streams
static SideStreams* side_streams(int dev, hipStream_t caller, size_t k, std::unique_lock<std::mutex>& lock) {
   SideStreams* ss = nullptr;
   {
      std::lock_guard<std::mutex> g(g_side_mu);
      for (SideStreams* x : g_side)
         if (x->device == dev && x->caller == caller) ss = x;
      if (!ss && g_side.size() >= FX_SIDE_SETS) {   // re-key the least recently used set of this device that nobody is using
         for (SideStreams* x : g_side)
            if (x->device == dev && (!ss || x->stamp < ss->stamp)) ss = x;
         if (ss) {
            std::unique_lock<std::mutex> l(ss->mu, std::try_to_lock);
            if (!l.owns_lock()) return nullptr;   // (in use right now: this call keeps its follow-ups on the caller's stream)
            ss->caller = caller;
            ss->stamp = ++g_side_clock;
            lock = std::move(l);
         }
      }
      if (!ss) {
         ss = new (std::nothrow) SideStreams();
         if (!ss) return nullptr;
         ss->device = dev;
         ss->caller = caller;
         if (hipEventCreateWithFlags(&ss->fork, hipEventDisableTiming) != hipSuccess) {
            delete ss;
            return nullptr;
         }
         g_side.push_back(ss);
      }
      ss->stamp = ++g_side_clock;
   }
   if (!lock.owns_lock()) lock = std::unique_lock<std::mutex>(ss->mu);
   while (ss->streams.size() < k) {
      hipStream_t s = nullptr;
      hipEvent_t e = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
         (void)hipStreamDestroy(s);
         return nullptr;
      }
      ss->streams.push_back(s);
      ss->joined.push_back(e);
   }
   return ss;
}

// m patterns over the same device-resident rows: results pattern-major ([m][n]) -- the elemental operators with an ARRAY of patterns
// (forgex.F90:74 / :163) against one batch.  Patterns on the 8-state tile tables (rows of up to 256 bytes) share ONE pass over the
// rows, a group of up to multi_max_patterns() per launch of fx_search_multi; every other pattern runs its own pipeline.
int fxamd_match_multi_device(fxamd_program* const* progs, int32_t m, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags,
                             int32_t* d_from, int32_t* d_to, void* hip_stream) {
   if (!progs || m < 0 || n < 0 || row_len < 0 || !d_flags || (d_from == nullptr) != (d_to == nullptr)) return FXAMD_E_ARG;
   for (int32_t i = 0; i < m; ++i)
      if (!progs[i]) return FXAMD_E_ARG;
   if (n == 0 || m == 0) return FXAMD_OK;
   hipStream_t st = (hipStream_t)hip_stream;
   // the same handle more than once (identical patterns share one cached program): computed once, its results copied to the other slots
   std::vector<int32_t> first_of((size_t)m);
   for (int32_t i = 0; i < m; ++i) {
      first_of[(size_t)i] = i;
      for (int32_t j = 0; j < i; ++j)
         if (progs[j] == progs[i]) {
            first_of[(size_t)i] = j;
            break;
         }
   }
   std::vector<int32_t> fused;
   if (!long_row(row_len) && !fx_env().no_multi && n <= slice_rows())   // (more rows than one enqueue takes: pattern by pattern, each sliced)
      for (int32_t i = 0; i < m; ++i) {
         const FxpHeader& h = progs[i]->prog.hdr();
         if (first_of[(size_t)i] == i && progs[i]->prog.status == 0 && (h.mode == FXP_MODE_SEARCH_ENGINE || h.mode == FXP_MODE_SEARCH_LITERAL) && !(h.flags & FXP_F_NFA_SIM) &&
             (fast_scheme(h, d_rows, row_len) == 0 || (fast_scheme(h, d_rows, row_len) == 2 && fx_env().multi_w16)) &&
             !(h.flags & FXP_F_PREFIX_CHECK))   // (prefix-check programs: the one-launch kernel only, see fast_scheme)
            fused.push_back(i);
      }
   // Automata of 9..16 states (scheme 2, the nibble tables: the same 4 KB of LDS per pattern) can share the pass too (round 6, FXAMD_MULTI_W16=1) and
   // do NOT by default: measured in one allocation (tools/exp_multi.py, gpurun call r06_m4) the shared pass is no faster with them -- 12.5 M x 128 B,
   // six patterns of which three on the nibble tables: 2.16-2.20 ms shared against 2.24-2.25 ms (their own span-kernel pipelines, last_path 20);
   // three nibble patterns alone 1.08-1.11 against 1.06 ms; 16 M x 64 B: 1.54 against 1.48 ms and 0.85 against 0.71 ms -- a lane that owns a
   // span of whole short rows beats the shared pass's tile of 64 rows by more than the second read of the rows costs.
   // The shared pass pays where a tile is small enough for full occupancy next to m patterns' tables -- rows of up to 128 bytes: 6 patterns
   // over 100 M x 128 B 15.4 ms against 16.9 ms pattern by pattern (tools/exp_multi.py, gpurun call r03_c11).
   // Rows longer than 128 bytes: one pipeline per pattern.  The shared pass LOST there in every variant measured over rounds 2-3 (six
   // patterns over config 3: 3.67 ms shared against 3.21 ms one by one; config 4: 0.75 against 0.69) -- the single-pattern
   // kernels stage half rows / speculate, the shared one stages whole rows once per group and runs m full scans -- so round 4 removed that
   // path instead of keeping a slower kernel behind a hook: fx_search_multi exists for rows of up to 128 bytes only.
   if (row_len > 128) fused.clear();
   const int ch = tile_chunks(row_len);
   // byte-level tables in the shared pass (nibble format; 8 KB of LDS per pattern then): when some fused pattern has them for these rows
   std::vector<int> obs((size_t)m, 0);
   bool any_bytes = false;
   for (int32_t i : fused) {
      // (ragged rows: fx_search_multi still pads them with the inert symbol 255, which byte-level tables do not have -- the
      //  pad-free scheme of round 4 is the one-launch kernel's)
      const int ob = row_len == 16 * tile_chunks(row_len) ? one_bytes_scheme(progs[i]->prog.hdr(), d_rows, row_len, fast_scheme(progs[i]->prog.hdr(), d_rows, row_len)) : 0;
      obs[(size_t)i] = (ob == 2 || ob == 3) && progs[i]->prog.hdr().mode == FXP_MODE_SEARCH_ENGINE && !fx_env().multi_no_bytes ? ob : 0;
      any_bytes = any_bytes || obs[(size_t)i] != 0;
   }
   // Launch groups: patterns WITH byte-level tables first (8 KB of LDS per pattern and the queue area), the others behind them, and the
   // group size, the kernel's table stride and `any_bytes` are per GROUP -- a group of ASCII-only patterns does not pay for the
   // others' byte-level tables with half the patterns per launch
   std::stable_partition(fused.begin(), fused.end(), [&](int32_t i) { return obs[(size_t)i] != 0; });
   if ((int)fused.size() < 2 || ch <= 0 || multi_max_patterns(ch, any_bytes) < 2) fused.clear();
   std::vector<char> done((size_t)m, 0);
   int dev = -1;
   if (!fused.empty()) FX_HIP(hipGetDevice(&dev));
   for (size_t g0 = 0, g1 = 0; g0 < fused.size(); g0 = g1) {
      FxMultiArgs a;
      std::memset(&a, 0, sizeof(a));
      const bool group_bytes = obs[(size_t)fused[g0]] != 0;   // (sorted: a group that starts without them holds none)
      const int gmax = multi_max_patterns(ch, group_bytes);
      g1 = std::min(fused.size(), g0 + (size_t)std::max(gmax, 1));
      if (g1 - g0 < 2 || gmax < 2) {   // a single leftover pattern takes its own (faster) pipeline
         if (g1 - g0 >= 2) g1 = g0 + 1;
         continue;
      }
      // The group's handles stay locked from the first PREPARE to the last follow-up (in address order: two threads fusing
      // overlapping groups cannot deadlock): PREPARE flips a handle's counter group, and nobody else may enqueue on that handle
      // until the shared first pass -- which zeroes the other group for the call after this one -- is in the stream.
      std::vector<fxamd_program*> group;
      for (size_t k = g0; k < g1; ++k) group.push_back(progs[fused[k]]);   // (distinct: `fused` holds first occurrences only)
      std::sort(group.begin(), group.end(), std::less<fxamd_program*>());
      std::vector<std::unique_lock<std::mutex>> locks;
      locks.reserve(group.size());
      for (fxamd_program* p : group) locks.emplace_back(p->mu);
      std::vector<SharedFirstPass> shs(g1 - g0);
      std::vector<DevScratch*> scs(g1 - g0, nullptr);
      std::vector<uint8_t*> blobs(g1 - g0, nullptr);
      size_t prepared = 0;   // patterns whose counter group PREPARE has flipped
      // an error before the shared first pass is enqueued: flip the groups back, so that the next call meets the zeroed group again
      auto undo = [&]() {
         for (size_t k = 0; k < prepared; ++k) scs[k]->parity ^= 1u;
      };
      int rc = FXAMD_OK;
      for (size_t k = g0; k < g1 && rc == FXAMD_OK; ++k) {
         fxamd_program* p = progs[fused[k]];
         rc = blob_for_device(p, dev, &blobs[k - g0]);
         if (rc == FXAMD_OK) rc = scratch_for(p, dev, st, &scs[k - g0]);
         if (rc != FXAMD_OK) break;
         const int64_t slot = fused[k];
         const uint32_t parity_before = scs[k - g0]->parity;
         rc = enqueue_batch(p, blobs[k - g0], scs[k - g0], d_rows, n, row_len, d_flags + slot * n, d_from ? d_from + slot * n : nullptr,
                            d_to ? d_to + slot * n : nullptr, st, 0u, FX_FP_PREPARE, &shs[k - g0]);
         if (rc != FXAMD_OK) {
            scs[k - g0]->parity = parity_before;   // (a PREPARE that failed half-way)
            break;
         }
         ++prepared;
         a.blob[a.m] = blobs[k - g0];
         a.sch[a.m] = (uint32_t)fast_scheme(p->prog.hdr(), d_rows, row_len);   // (0 or 2: see the `fused` filter)
         a.fp[a.m] = params_of(p->prog.hdr(), (int)a.sch[a.m], false);
         a.slot[a.m] = (uint32_t)slot;
         a.ctr[a.m] = shs[k - g0].ctr;
         a.worklist[a.m] = shs[k - g0].worklist;
         a.defer_tiles[a.m] = shs[k - g0].defer_tiles;
         a.bsch[a.m] = (uint32_t)obs[(size_t)slot];
         if (obs[(size_t)slot] != 0) {
            a.fpb[a.m] = params_of(p->prog.hdr(), obs[(size_t)slot], true);
            shs[k - g0].bytes_in_shared = true;
            // (finishing the exception rows inside the shared pass -- one mixed-pattern queue per wave -- is built and tested but OFF: a
            //  drained row decodes through its pattern's class map in GLOBAL memory, there is no LDS left for six of them, and the
            //  pass got slower: 0.898 against 0.750 ms for 6 UTF-8 patterns on config 4's rows; FXAMD_MULTI_INQ=1 turns it on)
            if ((p->prog.hdr().flags & FXP_F_FAST_UTF8) && a.sch[a.m] == 0u && fx_env().multi_inq) {
               a.inq[a.m] = 1u;
               shs[k - g0].exc_in_shared = true;
            }
         }
         a.any_bytes = group_bytes ? 1u : 0u;
         ++a.m;
      }
      if (rc != FXAMD_OK) {
         undo();
         return rc;
      }
      hipError_t e = hipErrorInvalidValue;
      switch (ch) {
         case 1: e = launch_multi<1>(d_rows, n, a, d_flags, d_from, d_to, (uint32_t)row_len, st); break;
         case 2: e = launch_multi<2>(d_rows, n, a, d_flags, d_from, d_to, (uint32_t)row_len, st); break;
         case 4: e = launch_multi<4>(d_rows, n, a, d_flags, d_from, d_to, (uint32_t)row_len, st); break;
         case 8: e = launch_multi<8>(d_rows, n, a, d_flags, d_from, d_to, (uint32_t)row_len, st); break;
         default: e = hipErrorInvalidValue; break;   // (never dispatched: the shared pass takes rows of up to 128 bytes)
      }
      if (e != hipSuccess) {
         undo();
         return hip_fail(e);
      }
      // every pattern's own follow-up passes (tiles it deferred, rows it listed): gated kernels, empty on pure-ASCII batches.  They
      // use the counter words and the worklist PREPARE chose (carried in `shs`, not recomputed).
      // (side by side on side streams when the batch may hold rows for them -- some pattern scans UTF-8 tiles in the shared pass or
      //  defers them; the scratch -- counter words, worklist -- is the one PREPARE chose on the caller's stream)
      std::unique_lock<std::mutex> side_lock;
      SideStreams* ss = nullptr;
      // (a caller's stream that is being CAPTURED into a hipGraph keeps its follow-ups on itself: the shared side streams would join the
      //  capture, and another thread that uses them meanwhile would meet a capture error -- ADVICE r03)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
         (void)hipGetLastError();
         cap = hipStreamCaptureStatusNone;
      }
      if (!fx_env().multi_serial && cap == hipStreamCaptureStatusNone) {
         ss = side_streams(dev, st, g1 - g0, side_lock);
         if (ss && hipEventRecord(ss->fork, st) != hipSuccess) ss = nullptr;
         if (!ss) {
            (void)hipGetLastError();
            if (side_lock.owns_lock()) side_lock.unlock();
         }
      }
      for (size_t k = g0; k < g1; ++k) {
         fxamd_program* p = progs[fused[k]];
         const int64_t slot = fused[k];
         hipStream_t fs = st;
         if (ss) {
            fs = ss->streams[k - g0];
            FX_HIP(hipStreamWaitEvent(fs, ss->fork, 0));
         }
         rc = enqueue_batch(p, blobs[k - g0], scs[k - g0], d_rows, n, row_len, d_flags + slot * n, d_from ? d_from + slot * n : nullptr,
                            d_to ? d_to + slot * n : nullptr, fs, 0u, FX_FP_DONE, &shs[k - g0]);
         if (ss) {   // joined back whatever happened: nothing of this call may stay detached from the caller's stream
            const hipError_t e1 = hipEventRecord(ss->joined[k - g0], fs);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, ss->joined[k - g0], 0) : e1;
            if (e2 != hipSuccess && rc == FXAMD_OK) rc = hip_fail(e2);
         }
         if (rc != FXAMD_OK) return rc;   // (the shared first pass ran: the counter groups are consistent)
         p->last_path = 15;   // first pass shared with other patterns
         done[(size_t)slot] = 1;
      }
   }
   for (int32_t i = 0; i < m; ++i) {
      if (done[(size_t)i]) continue;
      const int64_t src = first_of[(size_t)i];
      if (src != i) {
         FX_HIP(hipMemcpyAsync(d_flags + (int64_t)i * n, d_flags + src * n, (size_t)n, hipMemcpyDeviceToDevice, st));
         if (d_from && progs[i]->prog.hdr().mode != FXP_MODE_MATCH_ENGINE) {   // (`.match.` leaves from/to untouched)
            FX_HIP(hipMemcpyAsync(d_from + (int64_t)i * n, d_from + src * n, (size_t)n * 4u, hipMemcpyDeviceToDevice, st));
            FX_HIP(hipMemcpyAsync(d_to + (int64_t)i * n, d_to + src * n, (size_t)n * 4u, hipMemcpyDeviceToDevice, st));
         }
         continue;
      }
      const int rc = fxamd_match_batch_device(progs[i], d_rows, n, row_len, d_flags + (int64_t)i * n, d_from ? d_from + (int64_t)i * n : nullptr,
                                              d_to ? d_to + (int64_t)i * n : nullptr, hip_stream);
      if (rc != FXAMD_OK) return rc;
   }
   return FXAMD_OK;
}

// Host-buffer entry (what the Fortran module binds): the batch is cut into chunks that flow through two slots -- H2D copy of
// the chunk's rows, the kernels, D2H copy of its results into pinned staging, each slot on its own stream -- so the copy of
// one chunk overlaps the kernels and the result copy of the other.  Device buffers, pinned staging and streams are kept in
// the handle: steady-state calls allocate nothing.
int fxamd_match_batch_host(fxamd_program* p, const uint8_t* h_rows, int64_t n, int64_t row_len, uint8_t* h_flags, int32_t* h_from,
                           int32_t* h_to) {
   if (!p || n < 0 || row_len < 0 || row_len > 0x3FFFFFFF || !h_flags || (n > 0 && row_len > 0 && !h_rows)) return FXAMD_E_ARG;
   if ((h_from == nullptr) != (h_to == nullptr)) return FXAMD_E_ARG;
   if (p->prog.status >= 100) return FXAMD_E_UNSUPPORTED;
   if (n == 0) return FXAMD_OK;
   const FxpHeader& h = p->prog.hdr();
   if (h.mode == FXP_MODE_MATCH_ENGINE) {   // `.match.`: from/to stay untouched (nothing is allocated or copied for them)
      h_from = nullptr;
      h_to = nullptr;
   }
   const bool spans = h_from != nullptr;
   int dev = -1;
   FX_HIP(hipGetDevice(&dev));
   HostPipe* hp = acquire_pipe(dev);
   if (!hp) return FXAMD_E_NOMEM;
   struct PipeReturn {
      HostPipe* hp;
      ~PipeReturn() { release_pipe(hp); }
   } pipe_return{hp};
   // chunk size: about 64 MB of rows (a multiple of 64 rows: whole tiles), at least one row
   const size_t rl = (size_t)(row_len > 0 ? row_len : 1);
   int64_t chunk_rows = (int64_t)((size_t(64) << 20) / rl) & ~int64_t(63);
   if (chunk_rows < 64) chunk_rows = 64;
   if (chunk_rows > n) chunk_rows = n;
   const bool reg = fx_env().host_register && row_len > 0;   // pin the caller's rows in place for the call (experiment)
   bool registered = false;
   if (reg) registered = hipHostRegister(const_cast<uint8_t*>(h_rows), (size_t)n * rl, hipHostRegisterPortable) == hipSuccess;
   int rc = FXAMD_OK;
   auto fail = [&](hipError_t e) { rc = hip_fail(e); };
   auto prepare = [&](HostSlot& s) {
      if (!s.stream && rc == FXAMD_OK) {
         hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
         if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
         if (e != hipSuccess) return fail(e);
      }
      const size_t need = (size_t)chunk_rows * rl;
      if (s.row_bytes < need) {
         if (s.d_rows) (void)hipFree(s.d_rows);
         s.d_rows = nullptr;
         s.row_bytes = 0;
         const hipError_t e = hipMalloc((void**)&s.d_rows, need);
         if (e != hipSuccess) return fail(e);
         s.row_bytes = need;
      }
      if (s.rows_cap < chunk_rows || (spans && !s.spans)) {
         if (s.d_flags) (void)hipFree(s.d_flags);
         if (s.d_from) (void)hipFree(s.d_from);
         if (s.d_to) (void)hipFree(s.d_to);
         if (s.h_flags) (void)hipHostFree(s.h_flags);
         if (s.h_from) (void)hipHostFree(s.h_from);
         if (s.h_to) (void)hipHostFree(s.h_to);
         s.d_flags = s.h_flags = nullptr;
         s.d_from = s.d_to = s.h_from = s.h_to = nullptr;
         s.rows_cap = 0;
         s.spans = false;
         const int64_t cap = chunk_rows > s.rows_cap ? chunk_rows : s.rows_cap;
         hipError_t e = hipMalloc((void**)&s.d_flags, (size_t)cap);
         if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_flags, (size_t)cap, hipHostMallocDefault);
         if (e == hipSuccess && spans) e = hipMalloc((void**)&s.d_from, (size_t)cap * 4);
         if (e == hipSuccess && spans) e = hipMalloc((void**)&s.d_to, (size_t)cap * 4);
         if (e == hipSuccess && spans) e = hipHostMalloc((void**)&s.h_from, (size_t)cap * 4, hipHostMallocDefault);
         if (e == hipSuccess && spans) e = hipHostMalloc((void**)&s.h_to, (size_t)cap * 4, hipHostMallocDefault);
         if (e != hipSuccess) return fail(e);
         s.rows_cap = cap;
         s.spans = spans;
      }
   };
   // results of a slot's last chunk: wait for its D2H copies, then hand them to the caller's arrays
   auto drain = [&](HostSlot& s) {
      if (s.pending_row0 < 0) return;
      const hipError_t e = hipEventSynchronize(s.done);
      if (e != hipSuccess && rc == FXAMD_OK) fail(e);
      if (rc == FXAMD_OK) {
         std::memcpy(h_flags + s.pending_row0, s.h_flags, (size_t)s.pending_n);
         if (spans) {
            std::memcpy(h_from + s.pending_row0, s.h_from, (size_t)s.pending_n * 4);
            std::memcpy(h_to + s.pending_row0, s.h_to, (size_t)s.pending_n * 4);
         }
      }
      s.pending_row0 = -1;
      s.pending_n = 0;
   };
   for (HostSlot& s : hp->slot) {
      s.pending_row0 = -1;
      prepare(s);
   }
   int which = 0;
   for (int64_t r0 = 0; r0 < n && rc == FXAMD_OK; r0 += chunk_rows, which ^= 1) {
      HostSlot& s = hp->slot[which];
      drain(s);   // the slot's buffers are free again once its previous chunk has been handed over
      if (rc != FXAMD_OK) break;
      const int64_t cnt = n - r0 < chunk_rows ? n - r0 : chunk_rows;
      hipError_t e = hipSuccess;
      if (row_len > 0) e = hipMemcpyAsync(s.d_rows, h_rows + (size_t)r0 * rl, (size_t)cnt * rl, hipMemcpyHostToDevice, s.stream);
      if (e != hipSuccess) {
         fail(e);
         break;
      }
      rc = fxamd_match_batch_device(p, s.d_rows, cnt, row_len, s.d_flags, spans ? s.d_from : nullptr, spans ? s.d_to : nullptr, s.stream);
      if (rc != FXAMD_OK) break;
      e = hipMemcpyAsync(s.h_flags, s.d_flags, (size_t)cnt, hipMemcpyDeviceToHost, s.stream);
      if (e == hipSuccess && spans) e = hipMemcpyAsync(s.h_from, s.d_from, (size_t)cnt * 4, hipMemcpyDeviceToHost, s.stream);
      if (e == hipSuccess && spans) e = hipMemcpyAsync(s.h_to, s.d_to, (size_t)cnt * 4, hipMemcpyDeviceToHost, s.stream);
      if (e == hipSuccess) e = hipEventRecord(s.done, s.stream);
      if (e != hipSuccess) {
         fail(e);
         break;
      }
      s.pending_row0 = r0;
      s.pending_n = cnt;
   }
   for (HostSlot& s : hp->slot) {
      if (rc != FXAMD_OK && s.stream) (void)hipStreamSynchronize(s.stream);   // nothing of a failed call stays in flight
      drain(s);
   }
   if (registered) (void)hipHostUnregister(const_cast<uint8_t*>(h_rows));
   return rc;
}

// ---- caller-pinned host buffers --------------------------------------------------------------------------------------------------
// fxamd_match_batch_host copies the caller's rows with hipMemcpyAsync: from PAGEABLE memory the runtime stages them through its own
// pinned bounce buffers (one more CPU copy, and the call is not asynchronous), from pinned memory the DMA engine reads them in place.
// A caller that keeps a large batch in one array pins it once with these (the array stays usable as before).
int fxamd_host_register(void* p, int64_t bytes) {
   if (!p || bytes <= 0) return FXAMD_E_ARG;
   FX_HIP(hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable));   // (portable: pinned for every device of the process, not only the current one)
   return FXAMD_OK;
}
int fxamd_host_unregister(void* p) {
   if (!p) return FXAMD_E_ARG;
   FX_HIP(hipHostUnregister(p));
   return FXAMD_OK;
}

// ---- subroutine forms for Fortran `pure` hosts ---------------------------------------------------------------------------------
// A Fortran PURE FUNCTION may only have INTENT(IN) / VALUE dummies (F2018 C1590), and a compiler may merge or drop calls of a pure
// function whose result it does not need: the module binds these instead -- every output, the return code included, is an INTENT(OUT)
// argument of a pure SUBROUTINE.
void fxamd_f_compile(const char* pattern, int64_t pattern_len, int op, fxamd_program** out, int32_t* status, int32_t* rc) {
   const int r = fxamd_compile(pattern, pattern_len, op, out, status);
   if (rc) *rc = r;
}
void fxamd_f_program_free(fxamd_program* p, int32_t* rc) {
   fxamd_program_free(p);
   if (rc) *rc = FXAMD_OK;
}
void fxamd_f_strerror_copy(int32_t status, char* buf, int64_t capacity, int64_t* n) {
   const int64_t r = fxamd_strerror_copy(status, buf, capacity);
   if (n) *n = r;
}
// ---- device-resident batches for hosts without a device runtime of their own (the Fortran module's type(fx_batch)) ------------------
// The rows are uploaded once (or a caller's device pointer is wrapped) and stay in HBM across calls and patterns; every run leaves its
// results in device buffers the batch owns (m result sets of n flags and, with spans, n from / to values); the host fetches, counts or
// ignores them.  One stream per batch; the entries are serialised per batch.
struct fxamd_batch {
   uint8_t* d_rows = nullptr;
   bool owns_rows = false;
   int64_t n = 0, row_len = 0;
   int dev = 0;
   hipStream_t st = nullptr;
   uint8_t* d_flags = nullptr;
   int32_t *d_from = nullptr, *d_to = nullptr;
   int64_t cap_flags = 0, cap_spans = 0;   // result sets the buffers hold (in rows: sets * n)
   unsigned long long* d_count = nullptr;
   hipEvent_t ev = nullptr;                 // fxamd_batch_after: orders a producer stream before the batch's stream
   int32_t sets = 0;                        // result sets of the last run
   bool spans = false;                      // ... and whether it wrote from / to
   std::mutex mu;
};
static int batch_make(const uint8_t* d_rows, bool owns, int64_t n, int64_t row_len, fxamd_batch** out) {
   fxamd_batch* b = new (std::nothrow) fxamd_batch();
   if (!b) return FXAMD_E_NOMEM;
   b->d_rows = const_cast<uint8_t*>(d_rows);
   b->owns_rows = owns;
   b->n = n;
   b->row_len = row_len;
   // the device the rows live on (a wrapped pointer need not be on the current one): the batch's stream and buffers are made there
   int cur = -1;
   bool ok = hipGetDevice(&cur) == hipSuccess;
   b->dev = cur;
   if (ok && d_rows != nullptr) {
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, d_rows) == hipSuccess) b->dev = at.device;
      else (void)hipGetLastError();
   }
   if (ok && b->dev != cur) ok = hipSetDevice(b->dev) == hipSuccess;
   ok = ok && hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking) == hipSuccess && hipMalloc((void**)&b->d_count, sizeof(unsigned long long)) == hipSuccess &&
        hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) == hipSuccess;
   if (!ok) {
      g_last_hip_error = (int)hipGetLastError();
      if (b->st) (void)hipStreamDestroy(b->st);
      if (b->d_count) (void)hipFree(b->d_count);
      if (cur >= 0 && b->dev != cur) (void)hipSetDevice(cur);
      delete b;
      return FXAMD_E_HIP;
   }
   if (b->dev != cur) (void)hipSetDevice(cur);
   *out = b;
   return FXAMD_OK;
}
int fxamd_batch_upload(const uint8_t* h_rows, int64_t n, int64_t row_len, fxamd_batch** out) {
   if (!out || n < 0 || row_len < 0 || row_len > 0x3FFFFFFF || (n > 0 && row_len > 0 && !h_rows)) return FXAMD_E_ARG;
   *out = nullptr;
   uint8_t* d = nullptr;
   const size_t bytes = (size_t)n * (size_t)row_len;
   FX_HIP(hipMalloc((void**)&d, bytes + 16));   // (+16: never a zero-byte allocation)
   if (bytes != 0) {
      const hipError_t e = hipMemcpy(d, h_rows, bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
         g_last_hip_error = (int)e;
         (void)hipFree(d);
         return FXAMD_E_HIP;
      }
   }
   const int rc = batch_make(d, true, n, row_len, out);
   if (rc != FXAMD_OK) (void)hipFree(d);
   return rc;
}
int fxamd_batch_wrap(const uint8_t* d_rows, int64_t n, int64_t row_len, fxamd_batch** out) {
   if (!out || n < 0 || row_len < 0 || row_len > 0x3FFFFFFF || (n > 0 && row_len > 0 && !d_rows)) return FXAMD_E_ARG;
   *out = nullptr;
   return batch_make(d_rows, false, n, row_len, out);
}
void fxamd_batch_free(fxamd_batch* b) {
   if (!b) return;
   if (b->st) {
      (void)hipStreamSynchronize(b->st);
      (void)hipStreamDestroy(b->st);
   }
   if (b->owns_rows && b->d_rows) (void)hipFree(b->d_rows);
   if (b->d_flags) (void)hipFree(b->d_flags);
   if (b->d_from) (void)hipFree(b->d_from);
   if (b->d_to) (void)hipFree(b->d_to);
   if (b->d_count) (void)hipFree(b->d_count);
   if (b->ev) (void)hipEventDestroy(b->ev);
   delete b;
}
// Everything enqueued so far on `producer_hip_stream` (the kernel that wrote wrapped rows, a consumer still reading the result buffers of
// fxamd_batch_results) happens before whatever the batch enqueues next: the batch runs on a private non-blocking stream that nothing
// else orders against the caller's streams.
int fxamd_batch_after(fxamd_batch* b, void* producer_hip_stream) {
   if (!b) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(b->mu);
   // (on the batch's device: a null producer stream then means the rows' device, whatever the caller's current device is -- ADVICE r05)
   int cur = -1;
   if (hipGetDevice(&cur) != hipSuccess) cur = -1;
   if (cur != b->dev) FX_HIP(hipSetDevice(b->dev));
   hipError_t e = hipEventRecord(b->ev, (hipStream_t)producer_hip_stream);
   if (e == hipSuccess) e = hipStreamWaitEvent(b->st, b->ev, 0);
   if (cur >= 0 && cur != b->dev) (void)hipSetDevice(cur);
   FX_HIP(e);
   return FXAMD_OK;
}
int fxamd_batch_info(const fxamd_batch* b, int64_t* n, int64_t* row_len) {
   if (!b) return FXAMD_E_ARG;
   if (n) *n = b->n;
   if (row_len) *row_len = b->row_len;
   return FXAMD_OK;
}
static int batch_reserve(fxamd_batch* b, int32_t sets, bool spans) {
   const int64_t need = (int64_t)sets * std::max<int64_t>(b->n, 1);
   if (b->cap_flags < need) {
      if (b->d_flags) (void)hipFree(b->d_flags);
      b->d_flags = nullptr;
      b->cap_flags = 0;
      FX_HIP(hipMalloc((void**)&b->d_flags, (size_t)need + 16));
      b->cap_flags = need;
   }
   if (spans && b->cap_spans < need) {
      if (b->d_from) (void)hipFree(b->d_from);
      if (b->d_to) (void)hipFree(b->d_to);
      b->d_from = b->d_to = nullptr;
      b->cap_spans = 0;
      FX_HIP(hipMalloc((void**)&b->d_from, (size_t)need * 4 + 16));
      FX_HIP(hipMalloc((void**)&b->d_to, (size_t)need * 4 + 16));
      b->cap_spans = need;
   }
   return FXAMD_OK;
}
// m patterns over the resident rows (m = 1: fxamd_match_batch_device, else fxamd_match_multi_device); asynchronous on the batch's stream.
// `.match.` programs write no spans.  The results stay on the device: fxamd_batch_fetch / _count / _results.
int fxamd_batch_run(fxamd_program* const* progs, int32_t m, fxamd_batch* b, int with_spans) {
   if (!progs || m < 1 || !b) return FXAMD_E_ARG;
   for (int32_t i = 0; i < m; ++i)
      if (!progs[i]) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(b->mu);
   // (the buffers may be reallocated below: until this run has succeeded the batch advertises no result sets -- ADVICE r04)
   b->sets = 0;
   b->spans = false;
   int cur = -1;
   FX_HIP(hipGetDevice(&cur));
   if (cur != b->dev) FX_HIP(hipSetDevice(b->dev));
   bool spans = with_spans != 0;
   for (int32_t i = 0; i < m; ++i)
      if (progs[i]->prog.hdr().mode == FXP_MODE_MATCH_ENGINE) spans = false;
   int rc = batch_reserve(b, m, spans);
   if (rc == FXAMD_OK) {
      if (m == 1) rc = fxamd_match_batch_device(progs[0], b->d_rows, b->n, b->row_len, b->d_flags, spans ? b->d_from : nullptr, spans ? b->d_to : nullptr, b->st);
      else rc = fxamd_match_multi_device(progs, m, b->d_rows, b->n, b->row_len, b->d_flags, spans ? b->d_from : nullptr, spans ? b->d_to : nullptr, b->st);
   }
   if (rc == FXAMD_OK) {
      b->sets = m;
      b->spans = spans;
   }
   if (cur != b->dev) (void)hipSetDevice(cur);
   return rc;
}
int fxamd_batch_sync(fxamd_batch* b) {
   if (!b) return FXAMD_E_ARG;
   FX_HIP(hipStreamSynchronize(b->st));
   return FXAMD_OK;
}
// results of pattern `which` (0-based) of the last run, copied to host arrays (h_from / h_to may both be NULL); synchronous
int fxamd_batch_fetch(fxamd_batch* b, int32_t which, uint8_t* h_flags, int32_t* h_from, int32_t* h_to) {
   if (!b || which < 0 || (h_from == nullptr) != (h_to == nullptr)) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(b->mu);
   if (which >= b->sets || (h_from && !b->spans)) return FXAMD_E_ARG;
   int cur = -1;
   FX_HIP(hipGetDevice(&cur));
   if (cur != b->dev) FX_HIP(hipSetDevice(b->dev));
   struct Back {
      int cur, dev;
      ~Back() {
         if (cur != dev) (void)hipSetDevice(cur);
      }
   } back{cur, b->dev};
   const size_t o = (size_t)which * (size_t)b->n;
   if (b->n != 0) {
      if (h_flags) FX_HIP(hipMemcpyAsync(h_flags, b->d_flags + o, (size_t)b->n, hipMemcpyDeviceToHost, b->st));
      if (h_from) {
         FX_HIP(hipMemcpyAsync(h_from, b->d_from + o, (size_t)b->n * 4, hipMemcpyDeviceToHost, b->st));
         FX_HIP(hipMemcpyAsync(h_to, b->d_to + o, (size_t)b->n * 4, hipMemcpyDeviceToHost, b->st));
      }
   }
   FX_HIP(hipStreamSynchronize(b->st));
   return FXAMD_OK;
}
// number of matching rows of pattern `which` of the last run (a reduction on the device: 8 bytes cross the bus); synchronous
int fxamd_batch_count(fxamd_batch* b, int32_t which, int64_t* n_matches) {
   if (!b || which < 0 || !n_matches) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(b->mu);
   if (which >= b->sets) return FXAMD_E_ARG;
   int cur = -1;
   FX_HIP(hipGetDevice(&cur));
   if (cur != b->dev) FX_HIP(hipSetDevice(b->dev));
   struct Back {
      int cur, dev;
      ~Back() {
         if (cur != dev) (void)hipSetDevice(cur);
      }
   } back{cur, b->dev};
   unsigned long long c = 0;
   if (b->n != 0) {
      FX_HIP(hipMemsetAsync(b->d_count, 0, sizeof(unsigned long long), b->st));
      int64_t blocks = (b->n + 16 * 256 - 1) / (16 * 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(fx_count_flags, dim3((unsigned)blocks), dim3(256), 0, b->st, b->d_flags + (size_t)which * (size_t)b->n, b->n, b->d_count);
      FX_HIP(hipGetLastError());
      FX_HIP(hipMemcpyAsync(&c, b->d_count, sizeof(c), hipMemcpyDeviceToHost, b->st));
      FX_HIP(hipStreamSynchronize(b->st));
   }
   *n_matches = (int64_t)c;
   return FXAMD_OK;
}
// the device buffers of the last run, for hosts that can use device pointers: sets * n flags (and from / to when it had spans)
int fxamd_batch_results(fxamd_batch* b, const uint8_t** d_flags, const int32_t** d_from, const int32_t** d_to, int32_t* sets, void** hip_stream) {
   if (!b) return FXAMD_E_ARG;
   std::lock_guard<std::mutex> g(b->mu);
   if (d_flags) *d_flags = b->d_flags;
   if (d_from) *d_from = b->spans ? b->d_from : nullptr;
   if (d_to) *d_to = b->spans ? b->d_to : nullptr;
   if (sets) *sets = b->sets;
   if (hip_stream) *hip_stream = (void*)b->st;
   return FXAMD_OK;
}
// subroutine forms for Fortran hosts (see fxamd_f_compile)
void fxamd_f_batch_upload(const uint8_t* h_rows, int64_t n, int64_t row_len, fxamd_batch** out, int32_t* rc) {
   const int r = fxamd_batch_upload(h_rows, n, row_len, out);
   if (rc) *rc = r;
}
void fxamd_f_batch_wrap(const uint8_t* d_rows, int64_t n, int64_t row_len, fxamd_batch** out, int32_t* rc) {
   const int r = fxamd_batch_wrap(d_rows, n, row_len, out);
   if (rc) *rc = r;
}
void fxamd_f_batch_free(fxamd_batch* b, int32_t* rc) {
   fxamd_batch_free(b);
   if (rc) *rc = 0;
}
void fxamd_f_batch_run(fxamd_program* const* progs, int32_t m, fxamd_batch* b, int32_t with_spans, int32_t* rc) {
   const int r = fxamd_batch_run(progs, m, b, with_spans);
   if (rc) *rc = r;
}
void fxamd_f_batch_sync(fxamd_batch* b, int32_t* rc) {
   const int r = fxamd_batch_sync(b);
   if (rc) *rc = r;
}
void fxamd_f_batch_fetch(fxamd_batch* b, int32_t which, uint8_t* h_flags, int32_t* h_from, int32_t* h_to, int32_t* rc) {
   const int r = fxamd_batch_fetch(b, which, h_flags, h_from, h_to);
   if (rc) *rc = r;
}
void fxamd_f_batch_count(fxamd_batch* b, int32_t which, int64_t* n_matches, int32_t* rc) {
   const int r = fxamd_batch_count(b, which, n_matches);
   if (rc) *rc = r;
}

void fxamd_f_match_batch_host(fxamd_program* p, const uint8_t* h_rows, int64_t n, int64_t row_len, uint8_t* h_flags, int32_t* h_from,
                              int32_t* h_to, int32_t* rc) {
   const int r = fxamd_match_batch_host(p, h_rows, n, row_len, h_flags, h_from, h_to);
   if (rc) *rc = r;
}

}   // extern "C"
#pragma GCC visibility pop
