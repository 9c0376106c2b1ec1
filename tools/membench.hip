// membench.hip — HBM ceilings on this box: copy / read-only / write-only, float4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_copy(const uint4 *a, uint4 *b, size_t n) {
  for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const uint4 *a, uint4 *b, size_t n) {
  uint4 s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) { uint4 v = a[i]; s.x ^= v.x; s.y ^= v.y; s.z ^= v.z; s.w ^= v.w; }
  if ((s.x ^ s.y ^ s.z ^ s.w) == 0x12345u) b[0] = s;
}
__global__ __launch_bounds__(256) void k_write(uint4 *b, size_t n) {
  uint4 s = {1, 2, 3, 4};
  for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) b[i] = s;
}
// copy where each wave reads 8 KB contiguous and writes 8 rows x 1 KB at a row pitch (tile-like)
__global__ __launch_bounds__(256) void k_tile(const uint4 *a, uint4 *b, size_t nwaves, size_t pitch16) {
  size_t w = (blockIdx.x*(size_t)blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  size_t stride = ((size_t)gridDim.x*blockDim.x) >> 6;
  for (; w < nwaves; w += stride) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = a[w*512 + lane*8 + k];       // lane owns 128 B
    size_t row0 = (w/11)*8, col = (w%11)*64;                           // 11 waves per row group
#pragma unroll
    for (int k = 0; k < 8; k++) b[(row0 + k)*pitch16 + col + lane] = v[k];
  }
}
int main() {
  const size_t N = (size_t)796*1024*1024/16;   // 796 MB per buffer
  uint4 *a, *b; CK(hipMalloc(&a, N*16)); CK(hipMalloc(&b, N*16 + (1 << 20)));
  CK(hipMemset(a, 1, N*16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256*4, 256*8, 256*16, 256*32}) {
    for (int mode = 0; mode < 4; mode++) {
      float best = 1e9;
      for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, N);
        if (mode == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, N);
        if (mode == 2) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, N);
        if (mode == 3) hipLaunchKernelGGL(k_tile, dim3(grid), dim3(256), 0, 0, a, b, N/512, (size_t)704);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double bytes = (mode == 0 || mode == 3 ? 2.0 : 1.0)*N*16;
      printf("grid %5d %-6s %.4f ms  %.0f GB/s\n", grid, mode == 0 ? "copy" : mode == 1 ? "read" : mode == 2 ? "write" : "tile", best, bytes/best/1e6);
    }
  }
  return 0;
}
