// valu_ubench.hip — per-instruction VALU issue rate on gfx950 (lanes/clk/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1);} } while (0)
#define ITERS 2048

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
  float a[8];
  v2f p[8];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){a[i], a[i] + 1.f}; u[i] = i; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      if (OP == 4) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
      if (OP == 6) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));
      if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
      if (OP == 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      if (OP == 9) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
      if (OP == 10) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      if (OP == 11) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]));
      if (OP == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)u[i];
  if (s == 1234.5678f) out[0] = s;
}

static int g_wps = 8;   // waves per SIMD
template <int OP> static void run(const char *name, int lanes_per_inst) {
  float *d; CK(hipMalloc(&d, 4));
  const int blocks = 256*g_wps;     // g_wps blocks of 4 waves per CU -> g_wps waves per SIMD
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double insts = (double)blocks*4*ITERS*8;             // wave-instructions
  double per_simd_per_s = insts/(ms*1e-3)/(256*4);
  printf("%-22s %.3f ms  %.2f Gwave-inst/s/SIMD  => %.2f clk/inst @2.4GHz, %.1f lane-ops/clk/SIMD\n",
   name, ms, per_simd_per_s/1e9, 2.4e9/per_simd_per_s, 64.0*lanes_per_inst*per_simd_per_s/2.4e9);
}

int main(int argc, char **argv) {
  if (argc > 1) {
    for (g_wps = 1; g_wps <= 8; g_wps++) {
      printf("== %d wave(s) per SIMD\n", g_wps);
      run<0>("v_add_f32", 1); run<4>("v_floor_f32", 1); run<2>("v_pk_add_f32", 2);
    }
    return 0;
  }
  run<0>("v_add_f32", 1); run<1>("v_mul_f32", 1); run<2>("v_pk_add_f32", 2); run<3>("v_pk_mul_f32", 2);
  run<4>("v_floor_f32", 1); run<5>("v_med3_f32", 1); run<6>("v_cvt_pk_u8_f32", 1); run<7>("v_fma_f32", 1);
  run<8>("v_pk_fma_f32", 2); run<9>("v_cvt_f32_i32_sdwa", 1); run<10>("v_pk_mul_lo_u16", 2);
  run<11>("v_max3_f32", 1); run<12>("v_mov_b32", 1);
  return 0;
}
