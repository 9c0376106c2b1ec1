// membench2.hip — which part of the block-decode access pattern costs HBM efficiency?
// Each wave moves 8 KB in and 8 KB out (like 64 blocks -> 8 rows x 1 KB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1);} } while (0)
// RD: 0 coalesced (lane stride 16 B), 1 per-lane 128 B
// WR: 0 contiguous 8 KB, 1 pitched 8 rows x 1 KB, 2 pitched 8 rows x 1 KB with 24-B lane stride (x4+x2)
// ORDER: 0 waves sweep tiles in raster order, 1 workgroup (4 waves) takes 4 horizontally adjacent tiles
template <int RD, int WR, int NT>
__global__ __launch_bounds__(256) void k(const uint4 *a, uint4 *b, size_t nwaves, int wpr, size_t pitch16) {
  size_t w = (blockIdx.x*(size_t)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t stride = ((size_t)gridDim.x*blockDim.x) >> 6;
  for (; w < nwaves; w += stride) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = RD ? a[w*512 + lane*8 + k] : a[w*512 + k*64 + lane];
    if (WR == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (NT) __builtin_nontemporal_store(__builtin_bit_cast(u4v, v[k]), reinterpret_cast<u4v *>(&b[w*512 + k*64 + lane])); else b[w*512 + k*64 + lane] = v[k];
      }
    }
    else if (WR == 2) {
      // 64 lanes x 24 B = 1536-B run per row: dwordx4 + dwordx2 per lane (RGB shape);
      // moves 8 rows x 1536 B = 12 KB out per 8 KB in -> report uses 8+12 KB
      const size_t row0 = (w/wpr)*8;
      unsigned char *base = reinterpret_cast<unsigned char *>(b) + (w%wpr)*1536 + lane*24;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        unsigned char *p = base + (row0 + k)*pitch16*16;
        uint2 h = {v[k].x, v[k].w};
        if (NT) {
          __builtin_nontemporal_store(__builtin_bit_cast(u4v, v[k]), reinterpret_cast<u4v *>(p));
          __builtin_nontemporal_store(__builtin_bit_cast(u2v, h), reinterpret_cast<u2v *>(p + 16));
        }
        else { *reinterpret_cast<uint4 *>(p) = v[k]; *reinterpret_cast<uint2 *>(p + 16) = h; }
      }
    }
    else {
      const size_t row0 = (w/wpr)*8, col = (w%wpr)*64;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        uint4 *p = &b[(row0 + k)*pitch16 + col + lane];
        if (NT) __builtin_nontemporal_store(__builtin_bit_cast(u4v, v[k]), reinterpret_cast<u4v *>(p)); else *p = v[k];
      }
    }
  }
}
template <int RD, int WR, int NT> static void run(const char *name, const uint4 *a, uint4 *b, size_t N, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int wpr = WR == 2 ? 7 : 11; const size_t pitch16 = 704;
  float best = 1e9;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<RD, WR, NT>), dim3(grid), dim3(256), 0, 0, a, b, N/512, wpr, pitch16);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  printf("grid %5d %-44s %.4f ms  %.0f GB/s\n", grid, name, best, (WR == 2 ? 2.5 : 2.0)*N*16/best/1e6);
}
int main() {
  const size_t N = (size_t)796*1024*1024/16;
  uint4 *a, *b; CK(hipMalloc(&a, N*16)); CK(hipMalloc(&b, N*32 + (16 << 20)));
  CK(hipMemset(a, 1, N*16));
  for (int grid : {256*2, 256*8}) {
    run<1, 2, 0>("per-lane-128B read, RGB-shaped write", a, b, N, grid);
    run<1, 2, 1>("per-lane-128B read, RGB-shaped NT write", a, b, N, grid);
    run<0, 2, 1>("coalesced read, RGB-shaped NT write", a, b, N, grid);
    run<0, 0, 0>("coalesced read, contiguous write", a, b, N, grid);
    run<1, 0, 0>("per-lane-128B read, contiguous write", a, b, N, grid);
    run<0, 1, 0>("coalesced read, pitched write", a, b, N, grid);
    run<1, 1, 0>("per-lane-128B read, pitched write", a, b, N, grid);
    run<0, 0, 1>("coalesced read, contiguous NT write", a, b, N, grid);
    run<1, 1, 1>("per-lane-128B read, pitched NT write", a, b, N, grid);
  }
  return 0;
}
