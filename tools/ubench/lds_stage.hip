// Microbenchmark (GPU box): two ways of staging 64-row tiles of 256-byte rows through LDS as 128-byte half rows (8 KB per wave), in the
// shape of the headline kernel's memory path: every lane then reads its OWN row's eight 16-byte cells from LDS
// and folds them to one word per row (no automaton work: this is the memory path alone).
//   regs   coalesced global_load_dwordx4 into 8 x uint4 of VGPRs (the next half tile in flight during the LDS phase), ds_write_b128 into
//          the transposed, swizzled tile -- what fx_search_fast does today
//   dma2   global_load_lds_dwordx4 (gfx950): the load's 64 x 16 bytes go straight to LDS at M0 + lane * 16, no staging VGPRs and no
//          ds_write; slot(r, k) = 8 r + (k xor ((r >> 1) & 7)) keeps a row's half one coalesced 128-byte line and the per-lane
//          ds_read_b128 free of bank conflicts; two 8 KB buffers per wave, the next half tile in flight during the LDS phase
//   dma1   the same with ONE buffer per wave: ask, wait, read -- the overlap comes from the other waves of the SIMD
// regs and dma1 also with WHOLE 256-byte rows per stage (16 KB per wave).  Prints GB/s of input for each at 2 / 3 / 4 blocks (of 4
// waves) per CU (the whole-row and two-buffer variants fit two) and checks the folded words against each other.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

__global__ void k_fill(uint32_t* p, size_t n) {
   for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u + (uint32_t)(i >> 7);
}

__device__ __forceinline__ uint32_t fold(const uint4 v) { return v.x ^ (v.y * 3u) ^ (v.z * 5u) ^ (v.w * 7u); }

// ---- regs: register staging ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cell_t(uint32_t row, uint32_t k) { return k * 64u + (row ^ (k & 7u)); }   // the tile kernels' layout (fx_tile.hpp tile_cell)

template <int CH>   // CH = 8: half rows (two stages per tile, right half first); CH = 16: whole rows
__global__ __launch_bounds__(256) void k_regs(const uint8_t* __restrict__ rows, int64_t n_tiles, uint32_t* __restrict__ out) {
   extern __shared__ __attribute__((aligned(16))) uint4 lds[];
   constexpr uint32_t HALVES = 16 / CH;
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   uint4* tile = lds + wave * (64 * CH);
   const int64_t w0 = (int64_t)blockIdx.x * 4 + wave, ws = (int64_t)gridDim.x * 4;
   auto load = [&](uint4 (&st)[CH], int64_t t, uint32_t h) {
      const uint8_t* base = rows + (t << 6) * 256 + ((HALVES == 2 && h == 0) ? 128 : 0);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
         const uint32_t p = (uint32_t)i * 64u + lane, r = p / CH, k = p % CH;
         st[i] = t < n_tiles ? *reinterpret_cast<const uint4*>(base + r * 256u + k * 16u) : make_uint4(0, 0, 0, 0);
      }
   };
   uint4 st[CH];
   load(st, w0, 0);
   for (int64_t t = w0; t < n_tiles; t += ws) {
      uint32_t acc = 0;
#pragma unroll 1
      for (uint32_t h = 0; h < HALVES; ++h) {
#pragma unroll
         for (int i = 0; i < CH; ++i) {
            const uint32_t p = (uint32_t)i * 64u + lane, r = p / CH, k = p % CH;
            tile[cell_t(r, k)] = st[i];
         }
         if (h + 1 < HALVES) load(st, t, h + 1);
         else load(st, t + ws, 0);
#pragma unroll
         for (int k = CH - 1; k >= 0; --k) acc = acc * 31u + fold(tile[cell_t(lane, (uint32_t)k)]);
      }
      out[(t << 6) + lane] = acc;
   }
}

// ---- dma: direct-to-LDS loads ---------------------------------------------------------------------------------------------------
template <int CH>
__device__ __forceinline__ uint32_t swz(uint32_t r) { return CH == 16 ? (r & 15u) : ((r >> 1) & 7u); }
template <int CH>
__device__ __forceinline__ uint32_t slot(uint32_t r, uint32_t k) { return (uint32_t)CH * r + (k ^ swz<CH>(r)); }

template <bool DOUBLE, int CH>
__global__ __launch_bounds__(256) void k_dma(const uint8_t* __restrict__ rows, int64_t n_tiles, uint32_t* __restrict__ out) {
   extern __shared__ __attribute__((aligned(16))) uint4 lds[];
   const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   static_assert(!DOUBLE || CH == 8, "two buffers: half rows");
   constexpr uint32_t HALVES = 16 / CH;
   uint4* buf = lds + wave * (DOUBLE ? 1024 : 64 * CH);   // two buffers of 512 cells (DOUBLE), or one of 64 CH
   const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)buf;   // byte address of the wave's buffers in LDS
   const int64_t w0 = (int64_t)blockIdx.x * 4 + wave, ws = (int64_t)gridDim.x * 4;
   // instruction i of a half tile fills slots 64 i .. 64 i + 63: slot s = 64 i + lane -> row s >> 3, stored chunk position s & 7
   auto issue = [&](uint32_t b, int64_t t, uint32_t h) {
      const uint8_t* base = rows + (t < n_tiles ? t << 6 : 0) * 256 + ((HALVES == 2 && h == 0) ? 128 : 0);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
         const uint32_t s = (uint32_t)i * 64u + lane, r = s / CH, k = (s % CH) ^ swz<CH>(r);
         const uint8_t* g = base + r * 256u + k * 16u;
         const uint32_t m0 = lds_base + b * 8192u + (uint32_t)i * 1024u;
         asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(g) : "memory", "m0");
      }
   };
   if (!DOUBLE) {
      // one buffer: ask, wait, read -- the overlap comes from the other waves of the SIMD (no staging registers: many fit)
      for (int64_t t = w0; t < n_tiles; t += ws) {
         uint32_t acc = 0;
#pragma unroll 1
         for (uint32_t h = 0; h < HALVES; ++h) {
            issue(0, t, h);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = CH - 1; k >= 0; --k) acc = acc * 31u + fold(buf[slot<CH>(lane, (uint32_t)k)]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         }
         out[(t << 6) + lane] = acc;
      }
      return;
   }
   issue(0, w0, 0);
   uint32_t b = 0;
   for (int64_t t = w0; t < n_tiles; t += ws) {
      uint32_t acc = 0;
#pragma unroll 1
      for (uint32_t h = 0; h < 2; ++h) {
         if (h == 0) issue(b ^ 1u, t, 1);
         else issue(b ^ 1u, t + ws, 0);
         // the eight loads of THIS half tile are done when at most the eight just issued (and this wave's earlier result store) are left
         asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
         const uint4* cur = buf + b * 512u;
#pragma unroll
         for (int k = 7; k >= 0; --k) acc = acc * 31u + fold(cur[slot<8>(lane, (uint32_t)k)]);
         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of this buffer are done before the next round's loads overwrite it
         b ^= 1u;
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // (keeps the store below out of the count the next wait relies on: see main)
      out[(t << 6) + lane] = acc;
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char** argv) {
   const int64_t n_rows = (argc > 1 ? atoll(argv[1]) : 4000000) & ~63ll;
   const int64_t n_tiles = n_rows >> 6;
   uint8_t* d_rows = nullptr;
   uint32_t *d_a = nullptr, *d_b = nullptr;
   CK(hipMalloc((void**)&d_rows, (size_t)n_rows * 256 + 65536));
   CK(hipMalloc((void**)&d_a, (size_t)n_rows * 4));
   CK(hipMalloc((void**)&d_b, (size_t)n_rows * 4));
   hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(d_rows), (size_t)n_rows * 64);
   CK(hipDeviceSynchronize());
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0));
   CK(hipEventCreate(&e1));
   CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
   CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
   CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_regs<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
   uint32_t* d_c = nullptr;
   CK(hipMalloc((void**)&d_c, (size_t)n_rows * 4));
   const char* names[5] = {"regs  half rows ", "dma2  half rows ", "dma1  half rows ", "regs  whole rows", "dma1  whole rows"};
   for (int per_cu : {2, 3, 4}) {
      for (int kind = 0; kind < 5; ++kind) {
         const size_t lds = (kind == 1 || kind >= 3) ? 4 * 16384 : 4 * 8192;
         const unsigned grid = 256u * (unsigned)per_cu;
         float best = 1e30f;
         for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (kind == 0) hipLaunchKernelGGL(k_regs<8>, dim3(grid), dim3(256), lds, 0, d_rows, n_tiles, d_a);
            else if (kind == 1) hipLaunchKernelGGL((k_dma<true, 8>), dim3(grid), dim3(256), lds, 0, d_rows, n_tiles, d_b);
            else if (kind == 2) hipLaunchKernelGGL((k_dma<false, 8>), dim3(grid), dim3(256), lds, 0, d_rows, n_tiles, d_c);
            else if (kind == 3) hipLaunchKernelGGL(k_regs<16>, dim3(grid), dim3(256), lds, 0, d_rows, n_tiles, d_b);
            else hipLaunchKernelGGL((k_dma<false, 16>), dim3(grid), dim3(256), lds, 0, d_rows, n_tiles, d_c);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
         }
         printf("%s  blocks/CU %d  lds/block %zu  %.4f ms  %.0f GB/s of input\n", names[kind], per_cu, lds, best, (double)n_rows * 256.0 / (best * 1e-3) / 1e9);
      }
   }
   // the two kernels fold the same bytes in the same order
   uint32_t* h_a = (uint32_t*)malloc((size_t)n_rows * 4);
   uint32_t* h_b = (uint32_t*)malloc((size_t)n_rows * 4);
   uint32_t* h_c = (uint32_t*)malloc((size_t)n_rows * 4);
   CK(hipMemcpy(h_a, d_a, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
   CK(hipMemcpy(h_b, d_b, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
   CK(hipMemcpy(h_c, d_c, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
   int64_t bad_b = 0, bad_c = 0;
   for (int64_t i = 0; i < n_rows; ++i) {
      bad_b += h_a[i] != h_b[i];
      bad_c += h_a[i] != h_c[i];
   }
   printf("rows %lld  folded words differing from the half-row register kernel: whole-row register kernel %lld, whole-row dma1 %lld\n", (long long)n_rows, (long long)bad_b, (long long)bad_c);
   return (bad_b | bad_c) != 0;
}
