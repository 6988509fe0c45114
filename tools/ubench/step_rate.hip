// Microbenchmark (GPU box): cost of one automaton step per input byte for the three lookup-table formats of the tile kernels, in the
// shape of their inner loop (8 state-independent LDS lookups issued ahead of an 8-step dependent chain), 2 blocks of 4 waves per CU:
//   perm8   ds_read_b64  + v_perm_b32                                (<= 8 states)
//   wide16  ds_read_b128 + 2 v_perm_b32 + v_xor + v_and              (<= 16 states, bytes)
//   nib16   ds_read_b64  + v_lshlrev_b32 + v_lshrrev_b64 + v_and     (<= 16 states, nibbles)
//   perm4   ds_read_b32  + v_perm_b32                                (<= 4 states)
// Input bytes come from a per-lane xorshift stream restricted to `spread` distinct values (LDS bank conflicts depend on it).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_step(uint32_t* out, int iters, uint32_t seed, uint32_t mask) {
   __shared__ uint2 t8[256];
   __shared__ uint4 t16[256];
   __shared__ uint32_t t4[256];
   t4[threadIdx.x] = (threadIdx.x * 0x9E3779B9u + seed) & 0x03030303u;
   t8[threadIdx.x] = make_uint2((threadIdx.x * 0x9E3779B9u + seed) & 0x07070707u, (threadIdx.x * 0x85EBCA6Bu) & 0x07070707u);
   t16[threadIdx.x] = make_uint4(threadIdx.x * 0x9E3779B9u & 0x07070707u, threadIdx.x * 0x85EBCA6Bu & 0x07070707u, 0x80808080u | (threadIdx.x * 77u & 0x07070707u), 0x80808080u | (threadIdx.x * 131u & 0x07070707u));
   __syncthreads();
   uint32_t x = threadIdx.x * 2654435761u + seed + blockIdx.x;
   uint32_t st = (KIND == 2 || KIND == 4 || KIND == 5) ? (x & 15u) : (KIND == 3 ? (x & 3u) * 0x01010101u : (x & 7u) * 0x01010101u);
   for (int i = 0; i < iters; ++i) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      const uint32_t lo = x & mask, hi = (x * 0x01000193u) & mask;
      if (KIND == 0) {
         uint2 f[8];
#pragma unroll
         for (int k = 0; k < 8; ++k) f[k] = t8[((k < 4 ? lo : hi) >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
         for (int k = 0; k < 8; ++k) st = __builtin_amdgcn_perm(f[k].y, f[k].x, st);
      } else if (KIND == 1) {
         uint4 f[8];
#pragma unroll
         for (int k = 0; k < 8; ++k) f[k] = t16[((k < 4 ? lo : hi) >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
         for (int k = 0; k < 8; ++k) st = __builtin_amdgcn_perm(f[k].y, f[k].x, st) & __builtin_amdgcn_perm(f[k].w, f[k].z, st ^ 0x80808080u);
      } else if (KIND == 2) {
         uint2 f[8];
#pragma unroll
         for (int k = 0; k < 8; ++k) f[k] = t8[((k < 4 ? lo : hi) >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
         for (int k = 0; k < 8; ++k) st = (uint32_t)(((((uint64_t)f[k].y) << 32) | f[k].x) >> (st << 2)) & 15u;
      } else if (KIND == 4) {   // nibbles without the 64-bit shift: half select (v_cmp + v_cndmask), v_lshlrev_b32, v_bfe_u32 (the offset uses 5 bits: 4 * (st & 7))
         uint2 f[8];
#pragma unroll
         for (int k = 0; k < 8; ++k) f[k] = t8[((k < 4 ? lo : hi) >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
         for (int k = 0; k < 8; ++k) {
            const uint32_t half = st > 7u ? f[k].y : f[k].x;
            st = __builtin_amdgcn_ubfe(half, st << 2, 4u);
         }
      } else if (KIND == 5) {   // the same with the state kept pre-scaled (4 * id): v_cmp, v_cndmask, v_bfe_u32, v_lshlrev_b32
         uint2 f[8];
#pragma unroll
         for (int k = 0; k < 8; ++k) f[k] = t8[((k < 4 ? lo : hi) >> ((k & 3) * 8)) & 0xFFu];
         uint32_t s4 = st << 2;
#pragma unroll
         for (int k = 0; k < 8; ++k) {
            const uint32_t half = s4 > 31u ? f[k].y : f[k].x;
            s4 = __builtin_amdgcn_ubfe(half, s4, 4u) << 2;
         }
         st = s4 >> 2;
      } else {
         uint32_t f[8];   // <= 4 states: 4 next-state bytes per symbol, ds_read_b32 + v_perm_b32
#pragma unroll
         for (int k = 0; k < 8; ++k) f[k] = t4[((k < 4 ? lo : hi) >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
         for (int k = 0; k < 8; ++k) st = __builtin_amdgcn_perm(f[k], f[k], st);
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = st;
}

int main() {
   uint32_t* d;
   CK(hipMalloc(&d, 256 * 8 * 256 * 4));
   hipEvent_t a, b;
   CK(hipEventCreate(&a));
   CK(hipEventCreate(&b));
   const int iters = 20000;
   const uint32_t masks[3] = {0x0F0F0F0Fu, 0x3F3F3F3Fu, 0xFFFFFFFFu};
   for (int m = 0; m < 3; ++m)
      for (int kind = 0; kind < 6; ++kind)
         for (int bpc = 2; bpc <= 4; bpc += 2) {
            const int blocks = 256 * bpc;
            for (int rep = 0; rep < 2; ++rep) {
               CK(hipEventRecord(a));
               if (kind == 0) hipLaunchKernelGGL(k_step<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, masks[m]);
               if (kind == 1) hipLaunchKernelGGL(k_step<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, masks[m]);
               if (kind == 2) hipLaunchKernelGGL(k_step<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, masks[m]);
               if (kind == 3) hipLaunchKernelGGL(k_step<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, masks[m]);
               if (kind == 4) hipLaunchKernelGGL(k_step<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, masks[m]);
               if (kind == 5) hipLaunchKernelGGL(k_step<5>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, masks[m]);
               CK(hipEventRecord(b));
               CK(hipEventSynchronize(b));
            }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            const double bytes = (double)blocks * 256 * iters * 8;   // input bytes stepped over
            printf("%-6s distinct byte values %3u, %d waves/SIMD: %.3f ms  %.2f T steps/s  (%.2f cycles per wave-step per CU at 2.4 GHz)\n",
                   kind == 0 ? "perm8" : kind == 1 ? "wide16" : kind == 2 ? "nib16" : kind == 3 ? "perm4" : kind == 4 ? "nib32" : "nib32s", (masks[m] & 0xFF) + 1, bpc, ms, bytes / (ms * 1e-3) / 1e12,
                   ms * 1e-3 * 2.4e9 / ((double)iters * 8 * bpc * 4));
         }
   return 0;
}
