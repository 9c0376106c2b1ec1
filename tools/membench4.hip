// membench4.hip — the copy ceiling by workgroups in flight and by cache policy of the loads / stores (round 4:
// a copy kernel at 16 waves per CU moves 6.35 TB/s where hipMemcpyDtoDAsync and the same kernel at 32 waves per CU move
// 5.5).  hipcc --offload-arch=gfx950 -O3 tools/membench4.hip -o /tmp/membench4 && /tmp/membench4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1);} } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_copy(const v4u *a, v4u *b, size_t n) {
  typedef __attribute__((address_space(1))) v4u gv4u;
  for (size_t i = blockIdx.x*(size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x*256) {
    const v4u v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
    if (NTS) __builtin_nontemporal_store(v, (gv4u *)(uintptr_t)(b + i)); else b[i] = v;
  }
}
// each workgroup takes CONTIGUOUS chunks (1 MB at a time) instead of striding over the whole buffer
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k_copy_chunks(const v4u *a, v4u *b, size_t n, size_t chunk16) {
  typedef __attribute__((address_space(1))) v4u gv4u;
  for (size_t c = blockIdx.x*chunk16; c < n; c += (size_t)gridDim.x*chunk16) {
    const size_t e = c + chunk16 < n ? c + chunk16 : n;
    for (size_t i = c + threadIdx.x; i < e; i += 256) {
      const v4u v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
      if (NTS) __builtin_nontemporal_store(v, (gv4u *)(uintptr_t)(b + i)); else b[i] = v;
    }
  }
}
int main() {
  const size_t N = (size_t)1194393600/16;      // the fused kernel's volume each way (48 x 4K 4:2:0)
  v4u *a, *b; CK(hipMalloc(&a, N*16)); CK(hipMalloc(&b, N*16));
  CK(hipMemset(a, 1, N*16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 200; w++) hipLaunchKernelGGL((k_copy<1, 1>), dim3(1024), dim3(256), 0, 0, a, b, N);   // clocks
  for (int pol = 0; pol < 4; pol++) {
    for (int grid : {256, 512, 768, 1024, 1280, 1536, 2048, 4096, 16384}) {
      float best = 1e9;
      for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0, 0));
        for (int q = 0; q < 10; q++) {
          if (pol == 0) hipLaunchKernelGGL((k_copy<0, 0>), dim3(grid), dim3(256), 0, 0, a, b, N);
          if (pol == 1) hipLaunchKernelGGL((k_copy<0, 1>), dim3(grid), dim3(256), 0, 0, a, b, N);
          if (pol == 2) hipLaunchKernelGGL((k_copy<1, 0>), dim3(grid), dim3(256), 0, 0, a, b, N);
          if (pol == 3) hipLaunchKernelGGL((k_copy<1, 1>), dim3(grid), dim3(256), 0, 0, a, b, N);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; if (ms < best) best = ms;
      }
      printf("strided  loads %s stores %s grid %5d (%4.1f waves/CU): %.4f ms  %.0f GB/s\n", pol & 2 ? "nt   " : "plain", pol & 1 ? "nt   " : "plain",
             grid, grid*4/256.0 > 32 ? 32.0 : grid*4/256.0, best, 2.0*N*16/best/1e6);
    }
  }
  for (int grid : {512, 1024, 2048}) for (size_t chunk : {(size_t)4096, (size_t)65536}) {
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < 10; q++) hipLaunchKernelGGL((k_copy_chunks<1, 1>), dim3(grid), dim3(256), 0, 0, a, b, N, chunk);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; if (ms < best) best = ms;
    }
    printf("chunks of %zu KB, nt/nt, grid %5d: %.4f ms  %.0f GB/s\n", chunk*16/1024, grid, best, 2.0*N*16/best/1e6);
  }
  return 0;
}
