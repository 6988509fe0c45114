// Does the raw-buffer range check include soffset?  (gfx950; decides where long-row tile loads put their row offsets)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* p, uint32_t nrec, uint32_t soff, uint32_t* out) {
   __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec, 0x00020000);
   u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);          // voffset small, soffset carries the distance
   u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16 + soff, 0, 0);      // everything in voffset
   out[threadIdx.x * 2] = a.x;
   out[threadIdx.x * 2 + 1] = b.x;
}
int main() {
   uint8_t* d;
   uint32_t* o;
   hipMalloc(&d, 1 << 20);
   hipMemset(d, 0x5A, 1 << 20);
   hipMalloc(&o, 64 * 8);
   uint32_t h[128];
   const uint32_t cases[][2] = {{1024, 0}, {1024, 512}, {1024, 1024}, {1024, 4096}, {8192, 4096}};
   for (auto& c : cases) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1], o);
      hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
      printf("num_records %u soffset %u: lane0 via-soffset %08x via-voffset %08x | lane 40 (voff 640) %08x %08x\n", c[0], c[1], h[0], h[1], h[80], h[81]);
   }
   return 0;
}
