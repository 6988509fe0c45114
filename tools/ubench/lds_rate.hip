// Microbenchmark (GPU box): how many LDS instructions a CU issues per cycle when every lane brings its own address -- the table lookups of the
// tile kernels (one ds_read_b64 per input byte and lane).  Independent reads (eight in flight per lane, results xor-ed: <= 2 VALU per read), widths of
// 1 / 4 / 8 / 16 bytes, addresses drawn from `spread` distinct table entries per wave (1 = every lane the same entry; 0 = lane-linear: consecutive
// lanes, consecutive entries), 2 or 4 waves per SIMD.  The clock is calibrated with a dependent v_perm_b32 chain (4 cycles per instruction and wave).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_rate.hip -o tools/ubench/lds_rate && tools/ubench/lds_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_chain(uint32_t* out, int iters, uint32_t seed) {
   uint32_t a = (threadIdx.x + seed) & 0x07070707u;
   const uint32_t x = seed * 0x01020304u & 0x07070707u, y = seed * 0x04030201u & 0x07070707u;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 64; ++u) a = __builtin_amdgcn_perm(x, y, a);
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <int W>   // bytes per lane and read
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, int iters, uint32_t seed, uint32_t spread) {
   __shared__ __attribute__((aligned(16))) uint8_t tab[256 * 16];
   for (uint32_t i = threadIdx.x; i < 256 * 4; i += 256) reinterpret_cast<uint32_t*>(tab)[i] = i * 0x9E3779B9u + seed;
   __syncthreads();
   // eight entry numbers per lane, fixed for the run (the address pattern is what is measured)
   uint32_t x = threadIdx.x * 2654435761u + seed + blockIdx.x * 977u;
   uint32_t e[8];   // byte offsets
#pragma unroll
   for (int k = 0; k < 8; ++k) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      // (spread distinct entries at pseudo-random places of the table: entry number (167 r + 13) mod 256 for r < spread -- an odd multiplier permutes 0..255)
      e[k] = (spread == 0u ? ((threadIdx.x + 8u * (uint32_t)k) & 255u) : (spread == 1u ? (uint32_t)(k * 31) & 255u : ((((x >> 8) % spread) * 167u + 13u) & 255u))) * (W >= 4 ? (uint32_t)W : 4u);
   }
   uint32_t acc0 = 0, acc1 = 0;
   for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
         uint32_t off = e[k];
         asm volatile("" : "+v"(off));   // (the same address every round, but opaque: the read is not hoisted and costs no address arithmetic)
         if (W == 1) acc0 ^= tab[off];
         else if (W == 4) acc0 ^= *reinterpret_cast<const uint32_t*>(tab + off);
         else if (W == 8) {
            const uint2 v = *reinterpret_cast<const uint2*>(tab + off);
            acc0 ^= v.x;
            acc1 ^= v.y;
         } else {
            const uint4 v = *reinterpret_cast<const uint4*>(tab + off);
            acc0 ^= v.x ^ v.z;
            acc1 ^= v.y ^ v.w;
         }
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 ^ acc1;
}

int main() {
   hipDeviceProp_t prop;
   CK(hipGetDeviceProperties(&prop, 0));
   const int cus = prop.multiProcessorCount;
   uint32_t* out = nullptr;
   CK(hipMalloc((void**)&out, (size_t)cus * 8 * 256 * 4));
   hipEvent_t e0, e1;
   CK(hipEventCreate(&e0));
   CK(hipEventCreate(&e1));
   auto timed = [&](auto launch) -> float {
      for (int w = 0; w < 3; ++w) launch();
      (void)hipEventRecord(e0, nullptr);
      for (int r = 0; r < 5; ++r) launch();
      (void)hipEventRecord(e1, nullptr);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      return ms / 5.0f;
   };
   // clock: one wave per SIMD, a dependent chain of 64 * iters v_perm_b32 at 4 cycles each
   const int citers = 20000;
   const float cms = timed([&]() { hipLaunchKernelGGL(k_chain, dim3(cus), dim3(256), 0, nullptr, out, citers, 7u); });
   const double ghz = 64.0 * citers * 4.0 / (cms * 1e-3) / 1e9;
   printf("# %s, %d CUs; shader clock by a dependent v_perm_b32 chain (4 cycles per instruction): %.3f GHz\n", prop.name, cus, ghz);
   printf("# cycles per LDS instruction and CU (64 lanes, each its own address); 128 B / cycle would be 2 (4 B), 4 (8 B), 8 (16 B) cycles; 16 addresses / cycle: 4\n");
   const int iters = 4000;
   const uint32_t spreads[] = {0u, 1u, 4u, 16u, 64u, 256u};
   for (int W : {1, 4, 8, 16})
      for (uint32_t sp : spreads)
         for (int wps : {2, 4}) {
            const int blocks = cus * wps;   // blocks of four waves: wps blocks per CU = wps waves per SIMD
            float ms = 0;
            if (W == 1) ms = timed([&]() { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, nullptr, out, iters, 11u, sp); });
            else if (W == 4) ms = timed([&]() { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(256), 0, nullptr, out, iters, 11u, sp); });
            else if (W == 8) ms = timed([&]() { hipLaunchKernelGGL(k_lds<8>, dim3(blocks), dim3(256), 0, nullptr, out, iters, 11u, sp); });
            else ms = timed([&]() { hipLaunchKernelGGL(k_lds<16>, dim3(blocks), dim3(256), 0, nullptr, out, iters, 11u, sp); });
            const double instr_per_cu = (double)wps * 4.0 * iters * 8.0;   // wave-level LDS instructions per CU
            printf("ds_read %2d B  %-26s %d waves/SIMD: %7.3f ms  %5.2f cycles per instruction and CU\n", W,
                   sp == 0u ? "lane-linear addresses" : (sp == 1u ? "one address per wave" : (sp == 4u ? "4 distinct entries" : (sp == 16u ? "16 distinct entries" : (sp == 64u ? "64 distinct entries" : "256 distinct entries")))),
                   wps, ms, ms * 1e-3 * ghz * 1e9 / instr_per_cu);
         }
   CK(hipGetLastError());
   return 0;
}
