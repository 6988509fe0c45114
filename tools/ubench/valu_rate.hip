// Microbenchmark (GPU box): VALU issue rate per SIMD with 1, 2, 4 waves per SIMD -- v_perm_b32 chains as in fx_search_fast,
// and ds_read_b64 table lookups (random 8-byte entries of a 2 KiB table) per CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_perm(uint32_t* out, int iters, uint32_t seed) {
   uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
   const uint32_t x = seed * 0x01020304u, y = seed * 0x04030201u;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
         a0 = __builtin_amdgcn_perm(x, y, a0 & 0x07070707u);
         a1 = __builtin_amdgcn_perm(x, y, a1 & 0x07070707u);
         a2 = __builtin_amdgcn_perm(x, y, a2 & 0x07070707u);
         a3 = __builtin_amdgcn_perm(x, y, a3 & 0x07070707u);
         a4 = __builtin_amdgcn_perm(x, y, a4 & 0x07070707u);
         a5 = __builtin_amdgcn_perm(x, y, a5 & 0x07070707u);
         a6 = __builtin_amdgcn_perm(x, y, a6 & 0x07070707u);
         a7 = __builtin_amdgcn_perm(x, y, a7 & 0x07070707u);
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// dependent chain: one v_perm per step, as the state chain
__global__ __launch_bounds__(256) void k_chain(uint32_t* out, int iters, uint32_t seed) {
   uint32_t a = (threadIdx.x + seed) & 0x07070707u;
   const uint32_t x = seed * 0x01020304u & 0x07070707u, y = seed * 0x04030201u & 0x07070707u;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 64; ++u) a = __builtin_amdgcn_perm(x, y, a);
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, int iters, uint32_t seed) {
   __shared__ uint2 tab[256];
   tab[threadIdx.x] = make_uint2(threadIdx.x * 0x9E3779B9u + seed, threadIdx.x * 0x85EBCA6Bu);
   __syncthreads();
   uint32_t i0 = threadIdx.x * 7 + seed, i1 = i0 + 13, i2 = i0 + 29, i3 = i0 + 31, acc = 0;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
         const uint2 r0 = tab[i0 & 255u], r1 = tab[i1 & 255u], r2 = tab[i2 & 255u], r3 = tab[i3 & 255u];
         acc += r0.y ^ r1.y ^ r2.y ^ r3.y;
         i0 = r0.x; i1 = r1.x; i2 = r2.x; i3 = r3.x;
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
   uint32_t* d;
   CK(hipMalloc(&d, 256 * 64 * 256 * 4));
   hipEvent_t a, b;
   CK(hipEventCreate(&a));
   CK(hipEventCreate(&b));
   int clk = 0;
   CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
   printf("clock attr %d kHz\n", clk);
   for (int kind = 0; kind < 3; ++kind)
      for (int wps = 1; wps <= 8; wps *= 2) {
         const int blocks = 256 * wps;   // 256 threads = 4 waves = one per SIMD; wps blocks per CU
         const int iters = 20000 / (kind == 2 ? 4 : 1);
         for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(a));
            if (kind == 0) hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
            if (kind == 1) hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
            if (kind == 2) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
         }
         float ms = 0;
         CK(hipEventElapsedTime(&ms, a, b));
         const double per_wave = (double)iters * (kind == 2 ? 32 : 64);   // instrs of the measured kind per wave (perm: + as many v_and)
         const double ns_per = ms * 1e6 / (per_wave * wps);               // per SIMD (perm/chain) : time per instr with wps waves sharing the SIMD
         printf("%s waves/SIMD %d: %.3f ms, %.3f ns per instr per SIMD (%.2f cycles at 2.4 GHz)%s\n", kind == 0 ? "perm+and x8 indep" : kind == 1 ? "perm dependent  " : "ds_read_b64 x4  ",
                wps, ms, ns_per, ns_per * 2.4, kind == 2 ? "  [per CU: x4 waves -> divide by 4]" : "");
      }
   return 0;
}
