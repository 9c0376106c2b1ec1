// probe_cvt.hip — print what v_cvt_pk_u8_f32 does on gfx950 (saturation, rounding)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float *in, unsigned *out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main() {
  float h[] = {-5.f, -0.5f, -0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 127.5f, 254.5f, 254.9f,
   255.0f, 255.4f, 255.6f, 256.f, 300.f, 1e9f, -1e9f, __builtin_nanf(""), 100.999f, 7.0f};
  const int n = sizeof(h)/sizeof(h[0]);
  float *d; unsigned *o, r[64];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, n*4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n);
  hipMemcpy(r, o, n*4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) printf("cvt_pk_u8_f32(%g) = %u\n", h[i], r[i] & 255);
  return 0;
}
