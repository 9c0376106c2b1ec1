// copy_kernel.hip — what THIS device's memory system gives a kernel that does nothing but move the
// fused block-decode kernel's byte volume: the measured ceiling bench.py prints beside the kernels'
// rates.  16 bytes per lane per trip, grid-stride, whole 1 KB runs per
// wave, non-temporal stores like the kernels' own.  hipMemcpyDtoDAsync of the same volume is printed
// next to it: the runtime's copy reads 4.5-5.5 TB/s from box to box, below the kernels it is
// supposed to bound on some of them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t ck_v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void jga_stream_copy_kernel(const ck_v4u *src, ck_v4u *dst, size_t n16) {
  typedef __attribute__((address_space(1))) ck_v4u global_v4u;
  const size_t stride = (size_t)gridDim.x*256u;
  for (size_t i = (size_t)blockIdx.x*256u + threadIdx.x; i < n16; i += stride) {
    const ck_v4u v = __builtin_nontemporal_load(src + i);
    __builtin_nontemporal_store(v, (global_v4u *)(uintptr_t)(dst + i));
  }
}

extern "C" int jga_launch_stream_copy(void *dst, const void *src, size_t bytes, int grid, void *stream) {
  hipLaunchKernelGGL(jga_stream_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
   (const ck_v4u *)src, (ck_v4u *)dst, bytes/16);
  return (int)hipGetLastError();
}
