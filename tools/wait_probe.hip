// wait_probe.hip — does a wait on a hipEventBlockingSync event sleep or spin?  It depends on what
// was queued last: prints wall and thread-CPU milliseconds of the wait after (a) a kernel,
// (b) a kernel + a small device-to-host copy, (c) as (b) + an empty kernel behind the copy.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/wait_probe tools/wait_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <chrono>
__global__ void burn(unsigned *p, int n) {
  unsigned x = threadIdx.x;
  for (int i = 0; i < n; i++) x = x*1664525u + 1013904223u;
  if (x == 42u) p[0] = x;
}
__global__ void nothing() {}
static double cpu_ms() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec*1e3 + t.tv_nsec*1e-6; }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventBlockingSync);
  unsigned *d, *h; hipMalloc(&d, 256); hipHostMalloc(&h, 256, hipHostMallocDefault);
  for (int mode = 0; mode < 4; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(burn, dim3(256), dim3(256), 0, st, d, 3000000);
      if (mode >= 1 && mode <= 2) hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, st);
      if (mode == 2) hipLaunchKernelGGL(nothing, dim3(1), dim3(64), 0, st);
      const auto t0 = std::chrono::steady_clock::now(); const double c0 = cpu_ms();
      if (mode == 3) hipStreamSynchronize(st);
      else { hipEventRecord(ev, st); hipEventSynchronize(ev); }
      const double c1 = cpu_ms();
      const double w = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("%-58s wall %6.2f ms, thread CPU %6.2f ms\n",
       mode == 0 ? "kernel; record + sync a blocking event" : mode == 1 ? "kernel, 64-byte D2H copy; record + sync"
       : mode == 2 ? "kernel, D2H copy, empty kernel; record + sync" : "kernel; hipStreamSynchronize", w, c1 - c0);
    }
  }
  return 0;
}
