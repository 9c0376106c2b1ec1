// pipeline.cpp — pipelined batch decoder for one GPU.
//
// The reference decodes one image per frame on one thread and uploads it with
// glTexSubImage2D before three draws (src/jpeg_gpu.c:1231-1237, 1375-1397).
// Here N host threads run the entropy stage into PINNED slots while the GPU
// works on earlier images: each worker owns one HIP stream and two slots
// (double buffer): entropy decode -> hipMemcpyAsync H2D -> fused kernel ->
// optional D2H, all asynchronous on the worker's stream, so copies and
// kernels of different workers overlap each other and the host Huffman code
// (north_star: "overlapped with the GPU via pinned hipMemcpyAsync on a side
// stream").  Images are independent: no inter-GPU or inter-worker exchange.
// With cfg.transport = 1 the workers produce the reference's PACK wire format
// (src/xjpeg.c:484-496, 513-519, 531-535) instead of dense planes; only the words and
// the block index cross PCIe and jga_unpack_batch() expands them in HBM.
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "jga_internal.h"

namespace {

struct slot {
  short *h_coef = nullptr;          // pinned
  short *d_coef = nullptr;
  unsigned short *h_q = nullptr;    // pinned, 3*64
  unsigned short *d_q = nullptr;
  unsigned char *d_out = nullptr;
  unsigned char *h_out = nullptr;   // pinned (copy_back only)
  short *h_pack = nullptr;          // pinned PACK words (transport 1; capacity = cap_coef)
  short *d_pack = nullptr;
  int *h_index = nullptr;           // pinned block index
  int *d_index = nullptr;
  long long cap_coef = 0, cap_out = 0, cap_index = 0;
  hipEvent_t done = nullptr;
  jga_job *job = nullptr;           // in flight when non-null
  long long out_bytes = 0;
};

struct worker {
  hipStream_t stream = nullptr;
  slot slots[2];
};

}  // namespace

struct jga_pipeline {
  jga_pipeline_config cfg;
  std::vector<worker> workers;
};

namespace {

bool hip_ok(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  jga_fail("pipeline: HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return false;
}
#define HOK(call) hip_ok((call), #call)

void free_slot(slot &s) {
  if (s.h_coef) (void)hipHostFree(s.h_coef);
  if (s.d_coef) (void)hipFree(s.d_coef);
  if (s.h_pack) (void)hipHostFree(s.h_pack);
  if (s.d_pack) (void)hipFree(s.d_pack);
  if (s.h_index) (void)hipHostFree(s.h_index);
  if (s.d_index) (void)hipFree(s.d_index);
  if (s.h_q) (void)hipHostFree(s.h_q);
  if (s.d_q) (void)hipFree(s.d_q);
  if (s.d_out) (void)hipFree(s.d_out);
  if (s.h_out) (void)hipHostFree(s.h_out);
  if (s.done) (void)hipEventDestroy(s.done);
  s = slot();
}

bool ensure_slot(slot &s, long long coef_shorts, long long out_bytes, bool copy_back,
 long long index_count) {
  if (!s.done && !HOK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming))) return false;
  if (!s.h_q) {
    if (!HOK(hipHostMalloc((void **)&s.h_q, 3*64*sizeof(unsigned short), hipHostMallocDefault))) return false;
    if (!HOK(hipMalloc((void **)&s.d_q, 3*64*sizeof(unsigned short)))) return false;
  }
  if (coef_shorts > s.cap_coef) {
    if (s.h_coef) (void)hipHostFree(s.h_coef);
    if (s.d_coef) (void)hipFree(s.d_coef);
    if (s.h_pack) (void)hipHostFree(s.h_pack);
    if (s.d_pack) (void)hipFree(s.d_pack);
    s.h_coef = nullptr; s.d_coef = nullptr; s.h_pack = nullptr; s.d_pack = nullptr; s.cap_coef = 0;
    if (index_count) {   // PACK transport: the dense planes exist on the device only
      // a block is at most 1 + 63 words, so the dense size (+ slack for the producer's per-block check) holds any stream
      if (!HOK(hipHostMalloc((void **)&s.h_pack, (coef_shorts + 128)*sizeof(short), hipHostMallocDefault))) return false;
      if (!HOK(hipMalloc((void **)&s.d_pack, (coef_shorts + 128)*sizeof(short)))) return false;
    }
    else if (!HOK(hipHostMalloc((void **)&s.h_coef, coef_shorts*sizeof(short), hipHostMallocDefault))) return false;
    if (!HOK(hipMalloc((void **)&s.d_coef, coef_shorts*sizeof(short)))) return false;
    s.cap_coef = coef_shorts;
  }
  if (index_count > s.cap_index) {
    if (s.h_index) (void)hipHostFree(s.h_index);
    if (s.d_index) (void)hipFree(s.d_index);
    s.h_index = nullptr; s.d_index = nullptr; s.cap_index = 0;
    if (!HOK(hipHostMalloc((void **)&s.h_index, index_count*sizeof(int), hipHostMallocDefault))) return false;
    if (!HOK(hipMalloc((void **)&s.d_index, index_count*sizeof(int)))) return false;
    s.cap_index = index_count;
  }
  if (out_bytes > s.cap_out) {
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.h_out) (void)hipHostFree(s.h_out);
    s.d_out = nullptr; s.h_out = nullptr; s.cap_out = 0;
    if (!HOK(hipMalloc((void **)&s.d_out, out_bytes))) return false;
    if (copy_back && !HOK(hipHostMalloc((void **)&s.h_out, out_bytes, hipHostMallocDefault))) return false;
    s.cap_out = out_bytes;
  }
  return true;
}

// Wait for the slot's GPU work and hand the result to the job.
void retire(slot &s, bool copy_back) {
  if (!s.job) return;
  if (!HOK(hipEventSynchronize(s.done))) s.job->status = EXIT_FAILURE;
  else if (copy_back && s.job->host_out) memcpy(s.job->host_out, s.h_out, (size_t)s.out_bytes);
  s.job = nullptr;
}

void run_worker(jga_pipeline *pl, worker *w, jga_job *jobs, int n,
 std::atomic<int> *next) {
  const bool rgb = pl->cfg.out == JPEG_DECODE_RGB;
  const bool copy_back = pl->cfg.copy_back != 0;
  const bool packed = pl->cfg.transport == 1;
  int cur = 0;
  if (!HOK(hipSetDevice(pl->cfg.device))) return;
  for (;;) {
    const int i = next->fetch_add(1);
    if (i >= n) break;
    jga_job *job = &jobs[i];
    slot &s = w->slots[cur];
    jpeg_header hdr;
    jga_geom g;
    retire(s, copy_back);                     // slot reuse: previous image done?
    job->status = EXIT_FAILURE;
    if (jga_parse_header(job->jpeg, job->size, &hdr) != EXIT_SUCCESS) continue;
    if (jga_geom_from_header(&g, &hdr) != EXIT_SUCCESS) continue;
    job->width = g.width; job->height = g.height; job->nplanes = g.nplanes;
    const long long out_bytes = rgb ? g.rgb_bytes : g.yuv_bytes;
    long long want_coef = g.coef_shorts, want_out = (out_bytes + 15) & ~15ll;
    if (pl->cfg.max_coef_shorts > want_coef) want_coef = pl->cfg.max_coef_shorts;
    if (pl->cfg.max_out_bytes > want_out) want_out = pl->cfg.max_out_bytes;
    const long long nindex = packed ? jga_index_count(&g) : 0;
    if (!ensure_slot(s, want_coef, want_out, copy_back, nindex)) continue;
    // host entropy stage, straight into pinned memory
    long long nwords = 0;
    if (packed) {
      if (jga_entropy_decode_pack(job->jpeg, job->size, &g, s.h_pack, s.cap_coef + 128, s.h_index,
       &nwords, nullptr) != EXIT_SUCCESS) {
        continue;
      }
    }
    else if (jga_entropy_decode(job->jpeg, job->size, &g, s.h_coef, 0) != EXIT_SUCCESS) continue;
    memset(s.h_q, 0, 3*64*sizeof(unsigned short));
    for (int p = 0; p < g.nplanes; p++) {
      memcpy(s.h_q + 64*p, hdr.comp[p].quant->tbl, 64*sizeof(unsigned short));
    }
    unsigned char *dst = job->dev_out ? job->dev_out : s.d_out;
    const long long dst_cap = job->dev_out ? want_out : s.cap_out;
    if (!HOK(hipMemcpyAsync(s.d_q, s.h_q, 3*64*sizeof(unsigned short), hipMemcpyHostToDevice, w->stream))) continue;
    if (packed) {
      const long long even = (nwords + 1) & ~1ll;
      if (!HOK(hipMemcpyAsync(s.d_pack, s.h_pack, even*sizeof(short), hipMemcpyHostToDevice, w->stream))) continue;
      if (!HOK(hipMemcpyAsync(s.d_index, s.h_index, nindex*sizeof(int), hipMemcpyHostToDevice, w->stream))) continue;
      if (jga_unpack_batch(&g, 1, (const unsigned short *)s.d_pack, even, nwords, s.d_index, nindex,
       s.d_coef, g.coef_shorts, w->stream) != EXIT_SUCCESS) {
        continue;
      }
      job->h2d_bytes = even*(long long)sizeof(short) + nindex*(long long)sizeof(int);
    }
    else {
      if (!HOK(hipMemcpyAsync(s.d_coef, s.h_coef, g.coef_shorts*sizeof(short), hipMemcpyHostToDevice, w->stream))) continue;
      job->h2d_bytes = g.coef_shorts*(long long)sizeof(short);
    }
    if ((rgb ? jga_idct_rgb_batch(&g, 1, s.d_coef, g.coef_shorts, s.d_q, 1, dst, dst_cap, w->stream)
     : jga_idct_yuv_batch(&g, 1, s.d_coef, g.coef_shorts, s.d_q, 1, dst, dst_cap, w->stream))
     != EXIT_SUCCESS) {
      continue;
    }
    if (copy_back && job->host_out
     && !HOK(hipMemcpyAsync(s.h_out, dst, out_bytes, hipMemcpyDeviceToHost, w->stream))) {
      continue;
    }
    if (!HOK(hipEventRecord(s.done, w->stream))) continue;
    job->status = EXIT_SUCCESS;               // provisional; retire() may fail it
    s.job = job;
    s.out_bytes = out_bytes;
    cur ^= 1;
  }
  retire(w->slots[0], copy_back);
  retire(w->slots[1], copy_back);
}

}  // namespace

extern "C" {

JGA_EXPORT jga_pipeline *jga_pipeline_create(const jga_pipeline_config *cfg) {
  jga_pipeline *pl = new jga_pipeline();
  pl->cfg = *cfg;
  if (pl->cfg.transport != 0 && pl->cfg.transport != 1) {
    jga_fail("pipeline: transport must be 0 (planes) or 1 (PACK)");
    delete pl;
    return nullptr;
  }
  if (pl->cfg.out != JPEG_DECODE_YUV && pl->cfg.out != JPEG_DECODE_RGB) {
    jga_fail("pipeline: out must be JPEG_DECODE_YUV or JPEG_DECODE_RGB");
    delete pl;
    return nullptr;
  }
  if (pl->cfg.nthreads <= 0) {
    unsigned hc = std::thread::hardware_concurrency();
    pl->cfg.nthreads = hc ? (int)hc : 4;
  }
  if (!hip_ok(hipSetDevice(pl->cfg.device), "hipSetDevice")) {
    delete pl;
    return nullptr;
  }
  pl->workers.resize(pl->cfg.nthreads);
  for (auto &w : pl->workers) {
    if (!hip_ok(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking), "hipStreamCreate")) {
      jga_pipeline_destroy(pl);
      return nullptr;
    }
  }
  return pl;
}

JGA_EXPORT int jga_pipeline_run(jga_pipeline *pl, jga_job *jobs, int n) {
  std::atomic<int> next(0);
  std::vector<std::thread> threads;
  int failed = 0;
  const int nt = n < (int)pl->workers.size() ? n : (int)pl->workers.size();
  for (int i = 0; i < n; i++) jobs[i].status = EXIT_FAILURE;
  for (int t = 0; t < nt; t++) {
    threads.emplace_back(run_worker, pl, &pl->workers[t], jobs, n, &next);
  }
  for (auto &th : threads) th.join();
  for (int i = 0; i < n; i++) failed += jobs[i].status != EXIT_SUCCESS;
  return failed ? EXIT_FAILURE : EXIT_SUCCESS;
}

JGA_EXPORT void jga_pipeline_destroy(jga_pipeline *pl) {
  if (!pl) return;
  (void)hipSetDevice(pl->cfg.device);
  for (auto &w : pl->workers) {
    free_slot(w.slots[0]);
    free_slot(w.slots[1]);
    if (w.stream) (void)hipStreamDestroy(w.stream);
  }
  delete pl;
}

}  // extern "C"
