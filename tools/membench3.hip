// membench3.hip — bandwidth of the block-decode access shape vs occupancy and vs
// bytes in flight per wave.  Dynamic LDS is used only to cap waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1);} } while (0)
// NB blocks of 128 B per lane per iteration (NB=1: 8 KB per wave in flight, NB=2: 16 KB)
template <int NB, int PERSIST>
__global__ __launch_bounds__(256) void k(const uint4 *a, unsigned char *b, size_t nwaves, size_t pitch) {
  extern __shared__ char dummy[];
  size_t w = (blockIdx.x*(size_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t stride = ((size_t)gridDim.x*blockDim.x) >> 6;
  if (lane == 1000) dummy[threadIdx.x] = 1;
  for (; w*NB < nwaves; w += stride) {
    uint4 v[NB][8];
#pragma unroll
    for (int n = 0; n < NB; n++)
#pragma unroll
      for (int k = 0; k < 8; k++) v[n][k] = a[(w*NB + n)*512 + lane*8 + k];
#pragma unroll
    for (int n = 0; n < NB; n++) {
      const size_t ww = w*NB + n;
      unsigned char *base = b + ((ww/7)*8)*pitch + (ww%7)*1536 + lane*24;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        unsigned char *p = base + k*pitch;
        uint2 h = {v[n][k].x, v[n][k].w};
        __builtin_nontemporal_store(__builtin_bit_cast(u4v, v[n][k]), reinterpret_cast<u4v *>(p));
        __builtin_nontemporal_store(__builtin_bit_cast(u2v, h), reinterpret_cast<u2v *>(p + 16));
      }
    }
    if (!PERSIST) break;
  }
}
template <int NB, int PERSIST> static void run(const uint4 *a, unsigned char *b, size_t N, int waves_per_cu) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t nwaves = N/512;
  // cap occupancy with dynamic LDS: blocks of 4 waves; 160 KB / lds_per_block = blocks per CU
  const int blocks_per_cu = waves_per_cu/4;
  size_t lds = 160*1024/blocks_per_cu - 512; if (lds > 64*1024) lds = 64*1024;
  CK(hipFuncSetAttribute((const void *)k<NB, PERSIST>, hipFuncAttributeMaxDynamicSharedMemorySize, 64*1024));
  const int grid = PERSIST ? 256*blocks_per_cu : (int)((nwaves/NB + 3)/4);
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<NB, PERSIST>), dim3(grid), dim3(256), lds, 0, a, b, nwaves, (size_t)11264);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  printf("NB %d persist %d waves/CU %2d : %.4f ms  %.0f GB/s\n", NB, PERSIST, waves_per_cu, best, 2.5*N*16/best/1e6);
}
int main() {
  const size_t N = (size_t)796*1024*1024/16;
  uint4 *a; unsigned char *b; CK(hipMalloc(&a, N*16)); CK(hipMalloc(&b, N*32 + (16 << 20)));
  CK(hipMemset(a, 1, N*16));
  for (int w : {8, 12, 16, 20, 24, 32}) { run<1, 0>(a, b, N, w); run<1, 1>(a, b, N, w); run<2, 0>(a, b, N, w); }
  return 0;
}
