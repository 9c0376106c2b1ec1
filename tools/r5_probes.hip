// r5_probes.hip — three questions the short-job work of round 5 starts from (MI355X, ROCm 7.2):
//   pull    how fast does a KERNEL read pinned / registered host memory over the link (no copy call at all),
//           by workgroups in flight, against hipMemcpyAsync of the same bytes (one call, and one call per 0.77 MB file)
//   chain   what does a chain of N dependent small kernels cost launched one by one on a stream against the same
//           chain replayed as a hipGraph: host time in the API calls, wall time until the last one is through
//   params  what hipGraphExecKernelNodeSetParams costs per node (a graph replayed with other arguments)
// build: hipcc -O2 --offload-arch=gfx950 tools/r5_probes.hip -o tools/bin/r5_probes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include <algorithm>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void pull_copy(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x*256;
  for (size_t i = (size_t)blockIdx.x*256 + threadIdx.x; i < n16; i += stride) {
    const v4u v = __builtin_nontemporal_load(src + i);
    __builtin_nontemporal_store(v, dst + i);
  }
}
// the same with four loads in flight per lane
__global__ __launch_bounds__(256) void pull_copy4(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x*256;
  size_t i = (size_t)blockIdx.x*256 + threadIdx.x;
  for (; i + 3*stride < n16; i += 4*stride) {
    const v4u a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                c = __builtin_nontemporal_load(src + i + 2*stride), d = __builtin_nontemporal_load(src + i + 3*stride);
    __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2*stride); __builtin_nontemporal_store(d, dst + i + 3*stride);
  }
  for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
// contiguous 16 KB per workgroup trip (what a clean-up kernel reading chunks would do)
__global__ __launch_bounds__(256) void pull_chunks(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n16) {
  const size_t nchunks = (n16 + 1023)/1024;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    v4u v[4];
    for (int k = 0; k < 4; k++) {
      const size_t i = c*1024 + (size_t)k*256 + threadIdx.x;
      if (i < n16) v[k] = __builtin_nontemporal_load(src + i);
    }
    for (int k = 0; k < 4; k++) {
      const size_t i = c*1024 + (size_t)k*256 + threadIdx.x;
      if (i < n16) __builtin_nontemporal_store(v[k], dst + i);
    }
  }
}

__global__ void small_step(unsigned *p, int spin) {
  // one workgroup, a dependent chain of `spin` loads (~ a few microseconds of latency-bound work)
  unsigned v = p[0];
  for (int i = 0; i < spin; i++) v = p[v & 63] + 1;
  if (threadIdx.x == 0) p[0] = v & 63;
}

// notes the device clock at its start (ts[slot]) and at its end (ts[slot + 1]) and spins `spin` clock ticks in between
__global__ void stamp_spin(unsigned long long *ts, int slot, int spin) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) ts[slot] = t0;
  while (wall_clock64() - t0 < (unsigned long long)spin) { }
  if (threadIdx.x == 0) ts[slot + 1] = wall_clock64();
}

// slot = 128 bytes = 8 lanes x 16; mode 0: all of it, 1: first half only, 2: first half of ~79 % of the slots (by a hash)
__global__ __launch_bounds__(256) void half_lines(v4u *buf, size_t nslots, int mode, int write, unsigned *sink) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x*32;
  for (size_t slot = (size_t)blockIdx.x*32 + (threadIdx.x >> 3); slot < nslots; slot += stride) {
    const unsigned part = threadIdx.x & 7u;
    const bool sparse = mode == 1 || (mode == 2 && ((unsigned)(slot*2654435761u) >> 24) < 202u);
    if (sparse && part >= 4u) continue;
    v4u *p = buf + slot*8 + part;
    if (write) { const v4u v = {(unsigned)slot, part, 3u, 4u}; __builtin_nontemporal_store(v, p); }
    else { const v4u v = __builtin_nontemporal_load(p); acc += v.x ^ v.w; }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

static float time_pull(int which, int grid, const void *src, void *dst, size_t bytes, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int r = 0; r < 6; r++) {
    OK(hipEventRecord(e0, st));
    if (which == 0) hipLaunchKernelGGL(pull_copy, dim3(grid), dim3(256), 0, st, (const v4u *)src, (v4u *)dst, bytes/16);
    else if (which == 1) hipLaunchKernelGGL(pull_copy4, dim3(grid), dim3(256), 0, st, (const v4u *)src, (v4u *)dst, bytes/16);
    else hipLaunchKernelGGL(pull_chunks, dim3(grid), dim3(256), 0, st, (const v4u *)src, (v4u *)dst, bytes/16);
    OK(hipEventRecord(e1, st));
    OK(hipEventSynchronize(e1));
    float ms;
    OK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const char *what = argc > 1 ? argv[1] : "all";
  OK(hipSetDevice(0));
  hipStream_t st, st2;
  OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  OK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  if (!strcmp(what, "pull") || !strcmp(what, "all")) {
    const size_t bytes = 96u << 20;
    void *pinned = nullptr, *dev = nullptr;
    OK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
    void *plain = aligned_alloc(4096, bytes);
    memset(pinned, 1, bytes); memset(plain, 2, bytes);
    OK(hipHostRegister(plain, bytes, hipHostRegisterDefault));
    void *plain_dev = nullptr;
    OK(hipHostGetDevicePointer(&plain_dev, plain, 0));
    OK(hipMalloc(&dev, bytes));
    printf("== pull: a kernel reads %zu MB of host memory into HBM (best of 5, HIP events)\n", bytes >> 20);
    for (int src_kind = 0; src_kind < 2; src_kind++) {
      const void *src = src_kind ? plain_dev : pinned;
      for (int which = 0; which < 3; which++) {
        for (int grid : {32, 64, 128, 256, 512, 1024, 2048, 4096}) {
          const float ms = time_pull(which, grid, src, dev, bytes, st, e0, e1);
          printf("  %-22s %-12s grid %5d: %.3f ms  %.1f GB/s\n", src_kind ? "hipHostRegister(malloc)" : "hipHostMalloc",
           which == 0 ? "1 load/lane" : which == 1 ? "4 loads/lane" : "16KB chunks", grid, ms, bytes/ms/1e6);
        }
      }
    }
    // the copy engine on the same bytes: one call, and one call per 0.77 MB
    for (int src_kind = 0; src_kind < 2; src_kind++) {
      const char *src = (const char *)(src_kind ? plain : pinned);
      float best = 1e9f;
      for (int r = 0; r < 6; r++) {
        OK(hipEventRecord(e0, st));
        OK(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, st));
        OK(hipEventRecord(e1, st));
        OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
      }
      printf("  %-22s hipMemcpyAsync, one call: %.3f ms  %.1f GB/s\n", src_kind ? "hipHostRegister(malloc)" : "hipHostMalloc", best, bytes/best/1e6);
      const size_t piece = 770u << 10;
      for (int streams = 1; streams <= 2; streams++) {
        best = 1e9f;
        double best_host = 1e9;
        for (int r = 0; r < 6; r++) {
          OK(hipStreamSynchronize(st)); OK(hipStreamSynchronize(st2));
          const double t0 = now_ms();
          int k = 0;
          for (size_t o = 0; o + piece <= bytes; o += piece, k++) {
            OK(hipMemcpyAsync((char *)dev + o, src + o, piece, hipMemcpyHostToDevice, streams == 2 && (k & 1) ? st2 : st));
          }
          const double t1 = now_ms();
          OK(hipStreamSynchronize(st)); OK(hipStreamSynchronize(st2));
          const double t2 = now_ms();
          if (r && t2 - t0 < best) { best = (float)(t2 - t0); best_host = t1 - t0; }
        }
        printf("  %-22s hipMemcpyAsync per 0.75 MB on %d stream(s) (%d calls): %.3f ms wall  %.1f GB/s, the calls took %.3f ms (%.1f us each)\n",
         src_kind ? "hipHostRegister(malloc)" : "hipHostMalloc", streams, (int)(bytes/piece), best, bytes/best/1e6, best_host, best_host*1e3/(bytes/piece));
      }
    }
    // pull kernel in pieces of 12 MB, one launch each, one stream: do they follow each other at the link's rate?
    {
      const size_t piece = 12u << 20;
      float best = 1e9f;
      for (int r = 0; r < 6; r++) {
        OK(hipEventRecord(e0, st));
        for (size_t o = 0; o < bytes; o += piece) {
          hipLaunchKernelGGL(pull_copy4, dim3(512), dim3(256), 0, st, (const v4u *)((const char *)pinned + o), (v4u *)((char *)dev + o), piece/16);
        }
        OK(hipEventRecord(e1, st));
        OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
      }
      printf("  pull in 8 launches of 12 MB on one stream: %.3f ms  %.1f GB/s\n", best, bytes/best/1e6);
    }
    // small transfers: latency of one 0.77 MB pull against one 0.77 MB copy call (wall, incl. the wait)
    {
      const size_t piece = 770u << 10;
      double bp = 1e9, bc = 1e9;
      for (int r = 0; r < 20; r++) {
        OK(hipStreamSynchronize(st));
        double t0 = now_ms();
        hipLaunchKernelGGL(pull_copy4, dim3(64), dim3(256), 0, st, (const v4u *)pinned, (v4u *)dev, piece/16);
        OK(hipStreamSynchronize(st));
        double t1 = now_ms();
        OK(hipMemcpyAsync(dev, pinned, piece, hipMemcpyHostToDevice, st));
        OK(hipStreamSynchronize(st));
        double t2 = now_ms();
        if (r > 2) { bp = std::min(bp, t1 - t0); bc = std::min(bc, t2 - t1); }
      }
      printf("  one 0.75 MB transfer, launch to host-visible completion: pull kernel %.1f us, hipMemcpyAsync %.1f us\n", bp*1e3, bc*1e3);
    }
  }
  if (!strcmp(what, "contend") || !strcmp(what, "all")) {
    // does a kernel that reads host memory slow down kernels that work in HBM beside it?  An HBM copy (256 MB,
    // 1024 workgroups) timed alone and with a fetch of host memory running on another stream, by the fetch's grid
    const size_t hb = 256u << 20, fb = 256u << 20;
    void *a = nullptr, *b2 = nullptr, *host = nullptr, *dst = nullptr;
    OK(hipMalloc(&a, hb)); OK(hipMalloc(&b2, hb)); OK(hipMalloc(&dst, fb));
    OK(hipHostMalloc(&host, fb, hipHostMallocDefault));
    memset(host, 3, fb);
    OK(hipMemset(a, 1, hb));
    hipEvent_t f0, f1;
    OK(hipEventCreate(&f0)); OK(hipEventCreate(&f1));
    printf("== contend: an HBM copy of %zu MB (1024 workgroups) beside a kernel fetching host memory\n", hb >> 20);
    for (int which = 1; which < 3; which++) {
      for (int grid : {0, 8, 16, 32, 64, 160, 512, 2048}) {
        float best = 1e9f, fbest = 0.f;
        for (int r = 0; r < 5; r++) {
          OK(hipDeviceSynchronize());
          if (grid) {
            OK(hipEventRecord(f0, st2));
            if (which == 1) hipLaunchKernelGGL(pull_copy4, dim3(grid), dim3(256), 0, st2, (const v4u *)host, (v4u *)dst, fb/16);
            else hipLaunchKernelGGL(pull_chunks, dim3(grid), dim3(256), 0, st2, (const v4u *)host, (v4u *)dst, fb/16);
            OK(hipEventRecord(f1, st2));
          }
          // (the fetch takes ~4.5 ms: ten copies of ~0.1 ms each run inside it)
          OK(hipEventRecord(e0, st));
          for (int k = 0; k < 10; k++) hipLaunchKernelGGL(pull_copy4, dim3(1024), dim3(256), 0, st, (const v4u *)a, (v4u *)b2, hb/16);
          OK(hipEventRecord(e1, st));
          OK(hipEventSynchronize(e1));
          float ms; OK(hipEventElapsedTime(&ms, e0, e1));
          float fms = 0.f;
          if (grid) { OK(hipEventSynchronize(f1)); OK(hipEventElapsedTime(&fms, f0, f1)); }
          if (r && ms < best) { best = ms; fbest = fms; }
        }
        printf("  fetch %-12s grid %5d: HBM copy %.3f ms each = %.0f GB/s (read+write)%s", which == 1 ? "4 loads/lane" : "16KB chunks", grid, best/10, 2.0*hb/(best/10)/1e6,
         grid ? "" : "  (alone)\n");
        if (grid) printf(", the fetch %.3f ms = %.1f GB/s\n", fbest, fb/fbest/1e6);
      }
    }
  }
  if (!strcmp(what, "xstream") || !strcmp(what, "all")) {
    // what a dependency ACROSS streams costs: stream A runs a kernel (or a 12 MB H2D copy) and records an event,
    // stream B waits for the event and runs a kernel that notes the device clock when it starts; against the same
    // two on ONE stream.  (wall_clock64: 100 MHz)
    unsigned long long *ts = nullptr;
    OK(hipMalloc(&ts, 64));
    const size_t cb = 12u << 20;
    void *hp = nullptr, *dp = nullptr;
    OK(hipHostMalloc(&hp, cb, hipHostMallocDefault)); OK(hipMalloc(&dp, cb));
    hipEvent_t ev; OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    printf("== xstream: the gap between the end of work on stream A and the start of a kernel that depends on it\n");
    for (int mode = 0; mode < 4; mode++) {
      // 0: kernel -> kernel, one stream; 1: kernel -> event -> kernel on another stream; 2: copy -> kernel, one stream;
      // 3: copy -> event -> kernel on another stream (the copy's end is stamped by a kernel right behind it on A)
      double best = 1e9, worst = 0, sum = 0; int cnt = 0;
      for (int r = 0; r < 30; r++) {
        OK(hipDeviceSynchronize());
        if (mode < 2) hipLaunchKernelGGL(stamp_spin, dim3(1), dim3(64), 0, st, ts, 0, 20000);      // ~0.2 ms, stamps its END at ts[0]
        else {
          OK(hipMemcpyAsync(dp, hp, cb, hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(stamp_spin, dim3(1), dim3(64), 0, st, ts, 0, 0);
        }
        hipStream_t sb = (mode & 1) ? st2 : st;
        if (mode & 1) { OK(hipEventRecord(ev, st)); OK(hipStreamWaitEvent(sb, ev, 0)); }
        hipLaunchKernelGGL(stamp_spin, dim3(1), dim3(64), 0, sb, ts, 2, 0);                         // stamps its START at ts[2]
        OK(hipDeviceSynchronize());
        unsigned long long h[4];
        OK(hipMemcpy(h, ts, 32, hipMemcpyDeviceToHost));
        const double gap = ((double)h[2] - (double)h[1])/100.0;   // us (end stamp of the first at ts[1])
        if (r > 3) { best = std::min(best, gap); worst = std::max(worst, gap); sum += gap; cnt++; }
      }
      const char *names[4] = {"kernel -> kernel, one stream", "kernel -> event -> kernel on another stream", "12 MB copy -> (stamp) -> kernel, one stream",
       "12 MB copy -> (stamp) -> event -> kernel on another stream"};
      printf("  %-60s gap min %.1f  mean %.1f  max %.1f us\n", names[mode], best, sum/cnt, worst);
    }
  }
  if (!strcmp(what, "halfline") || !strcmp(what, "all")) {
    // does HBM move HALF lines?  1 GB of 128-byte slots: read (or write) all 128 bytes of every slot, the first 64 of
    // every slot, the first 64 of 79 % of the slots and all of the rest (what a compact coefficient-plane format
    // for blocks with empty rows 4-7 would do).  8 lanes per slot, 16 bytes each.
    const size_t bytes = 1u << 30;
    void *buf = nullptr, *sink = nullptr;
    OK(hipMalloc(&buf, bytes)); OK(hipMalloc(&sink, 1 << 20));
    OK(hipMemset(buf, 1, bytes));
    printf("== halfline: %zu MB of 128-byte slots, by which half of each is touched (best of 5)\n", bytes >> 20);
    for (int write = 0; write < 2; write++) {
      for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int r = 0; r < 6; r++) {
          OK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(half_lines, dim3(4096), dim3(256), 0, st, (v4u *)buf, bytes/128, mode, write, (unsigned *)sink);
          OK(hipEventRecord(e1, st));
          OK(hipEventSynchronize(e1));
          float ms; OK(hipEventElapsedTime(&ms, e0, e1));
          if (r && ms < best) best = ms;
        }
        const double frac = mode == 0 ? 1.0 : mode == 1 ? 0.5 : 0.79*0.5 + 0.21;
        printf("  %-5s %-44s %.3f ms  (%.0f GB/s of the bytes touched, %.0f GB/s of the slots' 128)\n", write ? "write" : "read",
         mode == 0 ? "all 128 bytes of every slot" : mode == 1 ? "the first 64 bytes of every slot" : "first 64 of 79 % of the slots, 128 of the rest",
         best, bytes*frac/best/1e6, bytes/best/1e6);
      }
    }
  }
  if (!strcmp(what, "chain") || !strcmp(what, "all")) {
    unsigned *p = nullptr;
    OK(hipMalloc(&p, 256));
    OK(hipMemset(p, 0, 256));
    printf("== chain: N dependent one-workgroup kernels, stream launches against a replayed hipGraph (best of 20)\n");
    for (int spin : {0, 40, 400}) {
      for (int N : {1, 8, 20, 40}) {
        double b_host = 1e9, b_wall = 1e9;
        for (int r = 0; r < 24; r++) {
          OK(hipStreamSynchronize(st));
          const double t0 = now_ms();
          for (int k = 0; k < N; k++) hipLaunchKernelGGL(small_step, dim3(1), dim3(64), 0, st, p, spin);
          const double t1 = now_ms();
          OK(hipStreamSynchronize(st));
          const double t2 = now_ms();
          if (r > 3) { b_host = std::min(b_host, t1 - t0); b_wall = std::min(b_wall, t2 - t0); }
        }
        hipGraph_t g; hipGraphExec_t ge;
        OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(small_step, dim3(1), dim3(64), 0, st, p, spin);
        OK(hipStreamEndCapture(st, &g));
        OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double g_host = 1e9, g_wall = 1e9;
        for (int r = 0; r < 24; r++) {
          OK(hipStreamSynchronize(st));
          const double t0 = now_ms();
          OK(hipGraphLaunch(ge, st));
          const double t1 = now_ms();
          OK(hipStreamSynchronize(st));
          const double t2 = now_ms();
          if (r > 3) { g_host = std::min(g_host, t1 - t0); g_wall = std::min(g_wall, t2 - t0); }
        }
        printf("  spin %3d, N %2d: stream  calls %.1f us, done after %.1f us (%.1f per kernel) | graph  call %.1f us, done after %.1f us (%.1f per kernel)\n",
         spin, N, b_host*1e3, b_wall*1e3, b_wall*1e3/N, g_host*1e3, g_wall*1e3, g_wall*1e3/N);
        if (N == 20 && spin == 40) {
          // replay with other arguments: set every node's parameters, then launch
          size_t nn = 0;
          OK(hipGraphGetNodes(g, nullptr, &nn));
          std::vector<hipGraphNode_t> nodes(nn);
          OK(hipGraphGetNodes(g, nodes.data(), &nn));
          double s_best = 1e9;
          for (int r = 0; r < 12; r++) {
            int sp = spin + (r & 1);
            void *args[2] = {&p, &sp};
            hipKernelNodeParams kp;
            memset(&kp, 0, sizeof(kp));
            kp.func = (void *)small_step; kp.gridDim = dim3(1 + (r & 1)); kp.blockDim = dim3(64); kp.kernelParams = args;
            const double t0 = now_ms();
            for (size_t k = 0; k < nn; k++) OK(hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp));
            const double t1 = now_ms();
            OK(hipGraphLaunch(ge, st));
            OK(hipStreamSynchronize(st));
            if (r > 1) s_best = std::min(s_best, t1 - t0);
          }
          printf("  params: hipGraphExecKernelNodeSetParams on %zu nodes: %.1f us (%.2f us per node)\n", nn, s_best*1e3, s_best*1e3/nn);
        }
        OK(hipGraphExecDestroy(ge)); OK(hipGraphDestroy(g));
      }
    }
    // the same chain with a host-visible readback at the end (what a decode's verdict copy is)
  }
  return 0;
}
