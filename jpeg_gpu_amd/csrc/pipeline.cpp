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
// With cfg.transport = 2 no Huffman decoding happens on the host at all: a few
// "lanes" (driver thread + stream + jga_huff_batch) take groups of cfg.batch jobs;
// the host only parses markers and unstuffs the scan into pinned memory, the
// compressed bytes cross PCIe, and the GPU entropy stage + fused kernel do the
// rest.  While one lane waits for its GPU work the others prepare their next group.
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <time.h>
#include <sys/prctl.h>
#include <unordered_map>
#include <vector>
#include "jga_internal.h"
#include "host_wait.h"

namespace {

// What jga_pipeline_config's scheduling fields come to once the defaults are filled in (and, in the
// tuning build, the JGA_PIPE_* variables of rounds 2-3 have had their say): one helper for
// jga_pipeline_create and jga_pipeline_plan_cfg, so that a plan is the plan a run makes.
struct sched_knobs {
  int lanes, batch, link_slots, dev_slots, groups_per_lane, min_group_eq, offload_at, copy_streams;
  bool ramp_first, blocking, trace;
  int short_ramp;                                // 0 equal groups, 1 rising, 2 falling (tuning build only)
};
sched_knobs resolve_knobs(const jga_pipeline_config &c) {
  sched_knobs k;
  k.lanes = c.depth > 0 ? c.depth : 6;
  k.batch = c.batch > 0 ? c.batch : 48;
  k.blocking = c.spin_waits == 0;
  k.trace = c.trace != 0;
  // (not configuration since round 5 — every other value measured slower or the same, profiles/r4_host_side_steps.md
  // — but still A/B-able in the tuning build)
  k.link_slots = 2;
  k.dev_slots = 3;
  k.groups_per_lane = 4;
  k.min_group_eq = 4;
  k.ramp_first = true;
  k.short_ramp = 0;
  k.offload_at = 8;
  k.copy_streams = 0;
  if (const char *e = jga_tune("JGA_PIPE_DEVICE_SLOTS")) k.dev_slots = atoi(e) > 0 ? atoi(e) : 1;
  if (const char *e = jga_tune("JGA_PIPE_SPIN")) k.blocking = atoi(e) == 0;
  if (const char *e = jga_tune("JGA_PIPE_COPY_STREAMS")) k.copy_streams = atoi(e) > 8 ? 8 : atoi(e) > 0 ? atoi(e) : 0;
  if (const char *e = jga_tune("JGA_PIPE_GROUPS_PER_LANE")) k.groups_per_lane = atoi(e) > 0 ? atoi(e) : 1;
  if (const char *e = jga_tune("JGA_PIPE_MIN_GROUP")) k.min_group_eq = atoi(e) > 0 ? atoi(e) : 1;
  if (const char *e = jga_tune("JGA_PIPE_RAMP_FIRST")) k.ramp_first = atoi(e) != 0;
  if (const char *e = jga_tune("JGA_PIPE_SHORT_RAMP")) k.short_ramp = atoi(e);
  if (const char *e = jga_tune("JGA_PIPE_LINK_SLOTS")) k.link_slots = atoi(e) > 0 ? atoi(e) : 0;
  if (const char *e = jga_tune("JGA_PIPE_OFFLOAD_AT")) k.offload_at = atoi(e);
  if (jga_tune("JGA_PIPE_TRACE")) k.trace = true;
  if (k.dev_slots > k.lanes) k.dev_slots = k.lanes;
  return k;
}

// Callers' pageable JPEG buffers registered with the device (jga_pipeline_config.input_cache_mb): address ->
// registration.  A lane ACQUIRES the buffers of its group before it queues work that reads them and RELEASES them when
// its stream has drained.  Two lifetimes (round 6):
//   run-scoped (the default, input_cache_mb = 0): the last user's release() undoes the registration at once — nothing of
//     the caller's memory is registered when jga_pipeline_run() returns, so the caller may free() it at will;
//   persistent (input_cache_mb > 0): entries stay, least recently used out first when the cache is full; only buffers
//     nobody holds are evicted or forgotten; the caller promises jga_pipeline_forget_input() before it frees one.
// hipHostRegister runs outside the lock (lanes register different buffers side by side) — a buffer in the middle of
// being registered by one lane reads as "not registered" to the others, who copy it as before.
// A registration is made for a FILE, not for an address: the entry keeps a fingerprint of the buffer's contents
// (size, first and last 64 bytes, sixteen 8-byte words spread over the rest) which the host re-reads at every sight.
// The caller's buffer is plain malloc memory (reference src/jpeg_info.c:31-62: malloc, fread, free in
// jpeg_info_clear): freed and handed out again at the same address for another file, it may be backed by other
// pages than the ones the device has mapped — such a buffer fails the check, loses its registration and is
// registered afresh (or copied, if somebody still holds the old one).  A buffer that meets "already registered" is
// checked against the cache's OWN entries first (a stale entry under another key that covers it is dropped, or — if
// it is in use — the buffer is copied): only ranges no own entry covers count as the caller's registration.
struct input_cache {
  struct fingerprint {
    size_t bytes = 0;
    unsigned long long head[8], tail[8], mid[16];
    bool operator==(const fingerprint &o) const {
      return bytes == o.bytes && !memcmp(head, o.head, sizeof(head)) && !memcmp(tail, o.tail, sizeof(tail))
       && !memcmp(mid, o.mid, sizeof(mid));
    }
  };
  static fingerprint print_of(const void *p, size_t bytes) {
    fingerprint f;
    const unsigned char *c = static_cast<const unsigned char *>(p);
    f.bytes = bytes;
    memcpy(f.head, c, 64);                               // (bytes >= MIN_BYTES)
    memcpy(f.tail, c + bytes - 64, 64);
    const size_t step = (bytes - 136)/16;
    for (int k = 0; k < 16; k++) memcpy(&f.mid[k], c + 64 + (size_t)k*step, 8);
    return f;
  }
  struct entry {
    size_t bytes = 0;
    unsigned long long last = 0;
    int users = 0, sights = 0;
    bool registered = false, busy = false, foreign = false;   // foreign: somebody else registered it (we never unregister it)
    fingerprint print;
  };
  std::mutex m;
  std::unordered_map<const void *, entry> map;
  size_t cap = 0, held = 0;
  unsigned long long tick = 0;
  int sight = 1;
  // input_cache_mb > 0: registrations outlive the run that made them (the caller's contract: the buffer stays
  // allocated until jga_pipeline_forget_input() or jga_pipeline_destroy()).  Default: a registration lives exactly as
  // long as the groups that hold it — released by its last user, it is unregistered at once, so that nothing of the
  // caller's memory is still registered when jga_pipeline_run() returns (free() of registered memory).
  bool persistent = false;
  static constexpr size_t MIN_BYTES = 64u << 10;     // (below this a copy is cheaper than a registration can ever be)
  static constexpr size_t MAX_TRACKED = 4096;        // addresses whose sights are being counted
  std::atomic<long long> n_registered{0}, n_evicted{0}, us_register{0}, n_in_place{0}, n_copied{0}, host_bytes{0}, n_stale{0};
  bool enabled() const { return cap > 0; }
  void drop(const void *p, entry &e) {               // (lock held, nobody uses it)
    if (e.registered && !e.foreign) (void)hipHostUnregister(const_cast<void *>(p));
    if (e.registered) held -= e.bytes;
  }
  // make room for `bytes` by unregistering idle entries, oldest first (lock held)
  bool make_room(size_t bytes) {
    while (held + bytes > cap) {
      const void *victim = nullptr;
      unsigned long long oldest = ~0ull;
      for (auto &kv : map) {
        if (kv.second.registered && !kv.second.busy && kv.second.users == 0 && kv.second.last < oldest) {
          oldest = kv.second.last; victim = kv.first;
        }
      }
      if (!victim) return false;
      drop(victim, map[victim]);
      map.erase(victim);
      n_evicted++;
    }
    return true;
  }
  // true: [p, p + bytes) is registered — for THESE contents — and now held by the caller (release() it).
  // `count_sight`: a run is looking at the buffer (explicit registration passes false and registers at once)
  bool acquire(const void *p, size_t bytes, bool count_sight = true) {
    if (!enabled() || !p || bytes < MIN_BYTES || bytes > cap) return false;
    const fingerprint now = print_of(p, bytes);
    {
      std::lock_guard<std::mutex> lk(m);
      auto it = map.find(p);
      if (it != map.end() && it->second.busy) return false;
      if (it != map.end() && it->second.registered) {
        if (it->second.print == now) { it->second.users++; it->second.last = ++tick; return true; }
        // the same address, another file (or another length): the registration is of no use — and may name pages
        // the buffer no longer has
        n_stale++;
        if (it->second.users > 0) return false;        // still being read by a running group: copy
        drop(p, it->second);
        map.erase(it);
        it = map.end();
      }
      // (addresses seen once and never again — a caller that decodes every file out of a fresh buffer — must not
      // pile up: the sight counts of buffers that are not registered are dropped when there are too many)
      if (it == map.end() && map.size() >= MAX_TRACKED) {
        for (auto jt = map.begin(); jt != map.end(); ) {
          if (!jt->second.registered && !jt->second.busy) jt = map.erase(jt); else ++jt;
        }
      }
      entry &e = it != map.end() ? it->second : map[p];
      e.last = ++tick;
      if (count_sight && ++e.sights < sight) return false;
      if (!make_room(bytes)) return false;
      e.busy = true;                                   // (`e` stays valid: unordered_map nodes do not move)
      e.bytes = bytes;
      held += bytes;
    }
    const auto t0 = std::chrono::steady_clock::now();
    // (tuning build, tests: JGA_PIPE_REGISTER_FAIL=1 — every registration fails the way it does when the process may
    // not lock more memory: the buffer is copied instead, same pixels)
    hipError_t rc = jga_tune("JGA_PIPE_REGISTER_FAIL") ? hipErrorOutOfMemory
     : hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault);
    bool foreign = false;
    if (rc == hipErrorHostMemoryAlreadyRegistered) {
      (void)hipGetLastError();
      // Whose registration is it?  If it is one of OURS under another key — a file since freed whose heap range now
      // holds this buffer — nothing ties this buffer's life to that entry: it could be evicted, dropped as stale or
      // forgotten while a lane still reads through it.  Such entries go (idle ones) and the registration
      // is tried again; if one of them is in use, this buffer is copied.
      bool ours_in_use = false, dropped = false;
      {
        std::lock_guard<std::mutex> lk(m);
        const unsigned char *lo = static_cast<const unsigned char *>(p), *hi = lo + bytes;
        for (auto jt = map.begin(); jt != map.end(); ) {
          const unsigned char *a = static_cast<const unsigned char *>(jt->first), *b = a + jt->second.bytes;
          const bool overlaps = jt->first != p && jt->second.registered && !jt->second.foreign && a < hi && lo < b;
          if (!overlaps) { ++jt; continue; }
          if (jt->second.users > 0 || jt->second.busy) { ours_in_use = true; ++jt; continue; }
          drop(jt->first, jt->second);
          jt = map.erase(jt);
          dropped = true;
        }
      }
      if (dropped && !ours_in_use) rc = hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault);
      if (rc != hipSuccess) (void)hipGetLastError();
      if (ours_in_use) rc = hipErrorUnknown;                   // (falls through to the failure path: copied)
    }
    if (rc == hipErrorHostMemoryAlreadyRegistered) {
      // the caller (or another pipeline) registered this range and did not say so: the device can address
      // it as it is — if it really covers the whole file
      // (EVERY page: a range whose two ends lie in other people's registrations — buffers back to back in one malloc
      // arena share the page at their seam — is not covered by them, and a device read of its middle ends the process)
      bool covered = true;
      unsigned char *c = const_cast<unsigned char *>(static_cast<const unsigned char *>(p));
      for (size_t o = 0; covered && o < bytes; o += 4096) {
        void *dp = nullptr;
        covered = hipHostGetDevicePointer(&dp, c + o, 0) == hipSuccess;
      }
      void *de = nullptr;
      covered = covered && hipHostGetDevicePointer(&de, c + bytes - 1, 0) == hipSuccess;
      if (covered) {
        rc = hipSuccess;
        foreign = true;
      }
      else (void)hipGetLastError();
    }
    us_register += (long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lk(m);
    entry &e = map[p];
    e.busy = false;
    if (rc != hipSuccess) {
      (void)hipGetLastError();
      held -= bytes;
      map.erase(p);
      return false;
    }
    e.registered = true;
    e.foreign = foreign;
    e.print = now;
    e.users = 1;
    n_registered++;
    return true;
  }
  void release(const void *p) {
    std::lock_guard<std::mutex> lk(m);
    auto it = map.find(p);
    if (it == map.end() || it->second.users <= 0) return;
    if (--it->second.users == 0 && !persistent && !it->second.busy) {
      drop(p, it->second);                                 // (run-scoped: the last user takes the registration with it)
      map.erase(it);
    }
  }
  int forget(const void *p) {
    std::lock_guard<std::mutex> lk(m);
    auto it = map.find(p);
    if (it == map.end()) return EXIT_SUCCESS;
    if (it->second.users > 0 || it->second.busy) return EXIT_FAILURE;
    drop(p, it->second);
    map.erase(it);
    return EXIT_SUCCESS;
  }
  void clear() {
    std::lock_guard<std::mutex> lk(m);
    for (auto &kv : map) if (kv.second.registered && !kv.second.foreign) (void)hipHostUnregister(const_cast<void *>(kv.first));
    map.clear();
    held = 0;
  }
};

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

// transport 2: one GPU-entropy lane
struct hlane {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;        // blocking: a lane waiting for its group sleeps
  jga_huff_batch *hb = nullptr;
  int hb_images = 0;
  long long hb_scan = 0;
  short *d_coef = nullptr;
  short *d_dc = nullptr;            // DC values beside the planes (jga_huff_decode_split)
  unsigned short *d_q = nullptr;
  unsigned char *d_out = nullptr, *h_out = nullptr;
  short *h_coef = nullptr;          // pinned, host-entropy fallback only
  long long cap_coef = 0, cap_dc = 0, cap_q = 0, cap_out = 0, cap_hout = 0, cap_hcoef = 0;
};

}  // namespace

struct jga_pipeline {
  jga_pipeline_config cfg;
  std::vector<worker> workers;
  std::vector<hlane> lanes;
  // transport 2: how many lanes may have kernels queued at a time.  The other lanes
  // prepare and upload their next group meanwhile, so the device always finds work: with
  // every lane free to launch, the streams share the device evenly, all groups finish
  // together and all lanes then parse and upload together with the device idle (23 % of
  // the time, measured).
  std::mutex dev_mutex;
  std::condition_variable dev_cv;
  int dev_slots = 3;
  int dev_capacity = 0, dev_free = 0;          // dev_slots x batch frame equivalents (device_turn)
  bool trace = false;                          // cfg.trace: a timeline of every run on stderr, offsets from run_t0
  std::chrono::steady_clock::time_point run_t0;
  double since_run_start_ms() const {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - run_t0).count();
  }
  input_cache inputs;                          // cfg.input_cache_mb: callers' pageable buffers kept registered
  // Lanes wait for the device several times per group; spinning in hipStreamSynchronize would
  // hold a core each, and a container may grant fewer cores than there are lanes: they poll and
  // sleep instead (host_wait.h; JGA_PIPE_SPIN=1 restores the spinning).
  int blocking = 1;
  // cfg.unstuff = 0 (auto): the scan clean-up runs on the device when the host is short of
  // cores — with 4 CPUs the device clean-up of pageable files runs 100 Gpixel/s against 80, of
  // pinned files (DMA'd where they lie) 124; 2 CPUs: 84 / 67 / 114 — and on the host when it is
  // not: there the device is the limit, and the extra kernels cost it 5-10 % (16 CPUs: 138 from
  // pageable or pinned files with the host clean-up, 124-129 with the device's;
  // profiles/r2_host_waits.txt)
  bool offload_cleanup = false;
  // Uploads of the lanes' groups go through a few shared copy streams, round robin, so that they
  // cross the link in the order the groups were prepared (two at a time: one stream alone moves
  // 33 GB/s, two 56): the first group's kernels start when ITS bytes are there.  With a copy
  // stream per lane the DMA engines share the link evenly, every group of a short job arrives at
  // the same moment and the device idles until then (JGA_PIPE_COPY_STREAMS=0: the old way).
  std::vector<hipStream_t> copy_streams;
  std::atomic<unsigned> copy_next{0};
  int groups_per_lane = 4, min_group_eq = 4;      // group sizing for jobs too short to reach a steady state
  int ramp_first = 1;                             // a long job's first groups rise in size (JGA_PIPE_RAMP_FIRST=0: all equal)
  // The groups' uploads take turns on the link, in the order their lanes got through prepare: with
  // every lane free to upload, three or four blobs share the link, all of them arrive late (8 ms
  // per 148 MB instead of 2.7) and the device waits for the first.  A turn is two units: one big
  // blob takes both (a single copy fills the link), smaller uploads and those made of many copy
  // calls (files DMA'd where they lie) one each, so that two of them hide each other's gaps.
  // [MI355X] 1536 x 4K pageable 116 -> 131-134 Gpixel/s, lighter content 315-327 -> 373-387,
  // 1024 x 1080p 83-89 -> 106-108 (profiles/r3_link_turns.txt; JGA_PIPE_LINK_SLOTS=0: off).
  int link_slots = 2, link_free = 2;
  bool copy_while_waiting = true;
  // SHORT runs on a host with cores to spare (round 6): the groups' uploads are queued, in the order their lanes get
  // through prepare, on two copy streams the lanes share — each group ONE copy call out of its pinned blob, into which
  // its lane's threads have copied the files (the first two groups to arrive name files that lie in pinned memory
  // instead: their bytes start crossing at once while the others stage).  The link then runs copy behind copy with
  // no host thread in between: with turns (a lane waits for its copy to arrive, gives the turn, the next lane wakes
  // and queues its own) every hand-over cost the link ~0.1 ms of idleness — 98 MB of a 128-file shard crossed in
  // 2.0-2.4 ms where the link needs 1.8 (profiles/r5_short_runs.md, r6_short_runs.md).
  hipStream_t fifo_streams[2] = {nullptr, nullptr};
  bool short_fifo = true, short_fifo_pinned = false;
  int fifo_nstreams = 1, fifo_threads = 1;       // (one copy stream moves 12 MB copies at the link's rate; a lane stages alone)
  int fifo_named = 0;                            // the run's first groups that name pinned files instead (tuning: JGA_PIPE_FIFO_NAMED)
  int short_ramp = 0;
  std::atomic<unsigned> run_ticket{0};
  std::mutex link_mutex;
  std::condition_variable link_cv;
  // transport 2: the lane threads live as long as the pipeline (a run used to create its eight
  // threads: 0.3 ms, the first thing a lone image's latency held); a run hands them its groups and
  // waits for all of them to report back
  std::vector<std::thread> lane_threads;
  std::mutex run_mutex;
  std::condition_variable run_cv, done_cv;
  unsigned long long run_gen = 0;
  int run_done = 0;
  bool quit = false;
  std::vector<std::vector<jga_job *>> *run_groups = nullptr;
  std::atomic<int> *run_next = nullptr;
  int run_threads = 1;
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
      // a block is at most 1 + 63 words, so the dense size holds any stream
      if (!HOK(hipHostMalloc((void **)&s.h_pack, coef_shorts*sizeof(short), hipHostMallocDefault))) return false;
      if (!HOK(hipMalloc((void **)&s.d_pack, coef_shorts*sizeof(short)))) return false;
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
void retire(slot &s, bool copy_back, bool blocking) {
  if (!s.job) return;
  // (usually over already; JGA_PIPE_SPIN=1 trades the nap's latency for a spinning core)
  if (!HOK(blocking ? jga_event_wait_sleeping(s.done) : hipEventSynchronize(s.done))) s.job->status = EXIT_FAILURE;
  else if (copy_back && s.job->host_out && !(s.job->pinned & 2)) memcpy(s.job->host_out, s.h_out, (size_t)s.out_bytes);
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
    retire(s, copy_back, pl->blocking != 0);  // slot reuse: previous image done?
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
      if (jga_entropy_decode_pack(job->jpeg, job->size, &g, s.h_pack, s.cap_coef, s.h_index,
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
    // From here on asynchronous work reads the slot's pinned buffers: a failure part-way must
    // drain the stream before the loop comes round and the host writes into them again.
    auto queue_frame = [&]() -> bool {
      if (!HOK(hipMemcpyAsync(s.d_q, s.h_q, 3*64*sizeof(unsigned short), hipMemcpyHostToDevice, w->stream))) return false;
      if (packed) {
        const long long even = (nwords + 1) & ~1ll;
        if (!HOK(hipMemcpyAsync(s.d_pack, s.h_pack, even*sizeof(short), hipMemcpyHostToDevice, w->stream))) return false;
        if (!HOK(hipMemcpyAsync(s.d_index, s.h_index, nindex*sizeof(int), hipMemcpyHostToDevice, w->stream))) return false;
        if (jga_unpack_batch(&g, 1, (const unsigned short *)s.d_pack, even, nwords, s.d_index, nindex,
         s.d_coef, g.coef_shorts, w->stream) != EXIT_SUCCESS) {
          return false;
        }
        job->h2d_bytes = even*(long long)sizeof(short) + nindex*(long long)sizeof(int);
      }
      else {
        if (!HOK(hipMemcpyAsync(s.d_coef, s.h_coef, g.coef_shorts*sizeof(short), hipMemcpyHostToDevice, w->stream))) return false;
        job->h2d_bytes = g.coef_shorts*(long long)sizeof(short);
      }
      if ((rgb ? jga_idct_rgb_batch(&g, 1, s.d_coef, g.coef_shorts, s.d_q, 1, dst, dst_cap, w->stream)
       : jga_idct_yuv_batch(&g, 1, s.d_coef, g.coef_shorts, s.d_q, 1, dst, dst_cap, w->stream))
       != EXIT_SUCCESS) {
        return false;
      }
      // (a destination the caller says is pinned takes the pixels straight from the device)
      if (copy_back && job->host_out
       && !HOK(hipMemcpyAsync((job->pinned & 2) ? job->host_out : s.h_out, dst, out_bytes, hipMemcpyDeviceToHost, w->stream))) {
        return false;
      }
      return HOK(hipEventRecord(s.done, w->stream));
    };
    if (!queue_frame()) {
      (void)hipStreamSynchronize(w->stream);
      continue;
    }
    job->status = EXIT_SUCCESS;               // provisional; retire() may fail it
    s.job = job;
    s.out_bytes = out_bytes;
    cur ^= 1;
  }
  retire(w->slots[0], copy_back, pl->blocking != 0);
  retire(w->slots[1], copy_back, pl->blocking != 0);
}


// ---- transport 2: GPU entropy lanes ------------------------------------------

void free_lane(hlane &l) {
  if (l.hb) jga_huff_destroy(l.hb);
  if (l.d_coef) (void)hipFree(l.d_coef);
  if (l.d_dc) (void)hipFree(l.d_dc);
  if (l.d_q) (void)hipFree(l.d_q);
  if (l.d_out) (void)hipFree(l.d_out);
  if (l.h_out) (void)hipHostFree(l.h_out);
  if (l.h_coef) (void)hipHostFree(l.h_coef);
  if (l.done) (void)hipEventDestroy(l.done);
  if (l.stream) (void)hipStreamDestroy(l.stream);
  l = hlane();
}

bool grow(void **p, long long *cap, long long want, bool pinned) {
  if (want <= *cap) return true;
  if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); }
  *p = nullptr; *cap = 0;
  if (pinned ? !HOK(hipHostMalloc(p, (size_t)want, hipHostMallocDefault)) : !HOK(hipMalloc(p, (size_t)want))) return false;
  *cap = want;
  return true;
}

// A share of the device, held for a group's kernels.  The budget is counted in 4K-frame
// equivalents — dev_slots full-size groups (dev_slots x cfg.batch frames): the small groups a short
// job is cut into are bound by the latency of their launches, not by the device's throughput,
// so more of them may run side by side (a 128-file 1080p shard: eight groups of 16 files, all at
// once, 5.1 -> 3.x ms).
struct device_turn {
  jga_pipeline *pl;
  int held = 0;
  explicit device_turn(jga_pipeline *p) : pl(p) {}
  void take(int units) {
    if (units < 1) units = 1;
    if (units > pl->dev_capacity) units = pl->dev_capacity;
    std::unique_lock<std::mutex> lk(pl->dev_mutex);
    pl->dev_cv.wait(lk, [this, units] { return pl->dev_free >= units; });
    pl->dev_free -= units;
    held = units;
  }
  void give() {
    if (!held) return;
    { std::lock_guard<std::mutex> lk(pl->dev_mutex); pl->dev_free += held; }
    pl->dev_cv.notify_all();
    held = 0;
  }
  ~device_turn() { give(); }
};

// A turn on the link, held from the moment a group's upload is queued until it has arrived.
struct link_turn {
  jga_pipeline *pl;
  int held = 0;
  explicit link_turn(jga_pipeline *p) : pl(p) {}
  static void take_hook(void *arg, long long bytes, int copies) {
    link_turn *t = static_cast<link_turn *>(arg);
    if (t->held) return;                                       // (the poll below already took it)
    t->take(copies == 1 && bytes >= (64ll << 20) ? t->pl->link_slots : 1);
  }
  // "would I get a turn now?" — takes it if so (jga_huff_set_upload_poll: a group that has to wait spends the wait
  // copying its files into its pinned blob, and then crosses the link as one copy call)
  static int poll_hook(void *arg) {
    link_turn *t = static_cast<link_turn *>(arg);
    if (t->held || t->pl->link_slots <= 0) return 1;
    std::lock_guard<std::mutex> lk(t->pl->link_mutex);
    if (t->pl->link_free < 1) return 0;
    t->pl->link_free -= 1;
    t->held = 1;
    return 1;
  }
  void take(int units) {
    if (pl->link_slots <= 0) return;
    if (units > pl->link_slots) units = pl->link_slots;
    std::unique_lock<std::mutex> lk(pl->link_mutex);
    pl->link_cv.wait(lk, [this, units] { return pl->link_free >= units; });
    pl->link_free -= units;
    held = units;
  }
  void give() {
    if (!held) return;
    { std::lock_guard<std::mutex> lk(pl->link_mutex); pl->link_free += held; }
    pl->link_cv.notify_all();
    held = 0;
  }
  ~link_turn() { give(); }
};

// Decode jobs[0..m) as ONE batch of the GPU entropy stage.  Fails as a whole (mixed
// geometry, an unparsable member, ...): GROUP_REJECTED when prepare() turned the group down
// (its per-member verdicts then say who is to blame), EXIT_FAILURE for anything else.
// `short_run`: the run gives every lane a group or two — what it waits for is the LAST group's chain of kernels
// behind the last byte over the link, so the groups take the route with the fewest host steps: clean-up on the
// device whatever the core count (no host pass over the bytes before the first upload can start), the files read
// where they lie, one host wait per group.
enum { GROUP_REJECTED = 2 };
uint64_t geometry_key(const unsigned char *p, int size);
int lane_group(jga_pipeline *pl, hlane &l, jga_job *const *jobv, int m, int threads, bool shared = true,
 bool reserve_full = false, bool short_run = false) {
  const bool rgb = pl->cfg.out == JPEG_DECODE_RGB;
  const bool copy_back = pl->cfg.copy_back != 0;
  std::vector<const unsigned char *> ptrs((size_t)m);
  std::vector<int> sizes((size_t)m);
  long long total = 0;
  jga_geom g;
  for (int i = 0; i < m; i++) {
    ptrs[i] = jobv[i]->jpeg;
    sizes[i] = jobv[i]->size;
    total += jobv[i]->size + 4096;
  }
  // In a run that gives every lane several groups, buffers are sized for a FULL group of this
  // geometry (cfg.batch frame equivalents) whatever this group holds: the ramped first groups of a
  // long job must not leave the lanes re-allocating — pinned host memory at that — a few groups
  // later.  A lone image or a short run takes what it needs (a full 4K group is ~2.5 GB of HBM per
  // lane and, with copy_back, over 1 GB of pinned host memory); and if the full size cannot be had
  // the group's own size is tried before the jobs are failed.
  int full = m;
  if (reserve_full) {
    const uint64_t key = geometry_key(jobv[0]->jpeg, jobv[0]->size);
    const long long px = (long long)((key >> 48) & 0xffff)*(long long)((key >> 32) & 0xffff);
    const long long batch = pl->cfg.batch > 0 ? pl->cfg.batch : 48;
    long long cap = px > 0 ? (batch*3840ll*2160 + px/2)/px : batch;
    cap = cap < 1 ? 1 : cap > 16*batch ? 16*batch : cap;
    if (cap > full) full = (int)cap;
  }
  // (with the clean-up on the device the batch holds the raw scans AND the clean streams)
  const bool on_device = pl->cfg.unstuff == 2 || (pl->cfg.unstuff == 0 && (pl->offload_cleanup || short_run));
  const long long total_full = (total/m*full + total/4)*(on_device ? 2 : 1);
  if (on_device) total *= 2;
  if (!l.hb || m > l.hb_images || total > l.hb_scan) {
    if (l.hb) jga_huff_destroy(l.hb);
    const int had_images = l.hb_images;
    const long long had_scan = l.hb_scan;
    l.hb_images = full > had_images ? full : had_images;
    l.hb_scan = total_full > had_scan ? total_full : had_scan;
    l.hb = jga_huff_create(l.hb_images, l.hb_scan);
    if (!l.hb && (l.hb_images > m || l.hb_scan > total)) {     // not at the full size: at this group's, then
      l.hb_images = m > had_images ? m : had_images;
      l.hb_scan = total > had_scan ? total : had_scan;
      full = m;
      l.hb = jga_huff_create(l.hb_images, l.hb_scan);
    }
    if (!l.hb) { l.hb_images = 0; l.hb_scan = 0; return EXIT_FAILURE; }
    (void)jga_huff_set_option(l.hb, JGA_HUFF_OPT_TRACE, pl->trace);
  }
  const bool trace = pl->trace;
  auto thread_cpu_ms = []() {          // CPU time this lane thread has burnt so far
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec*1e3 + (double)ts.tv_nsec*1e-6;
  };
  const auto t_a = std::chrono::steady_clock::now();
  const double c_a = trace ? thread_cpu_ms() : 0.0;
  jga_huff_set_threads(l.hb, threads);
  // From the first queued copy or fetch on, the device may be reading the callers' JPEG buffers (pinned, cached and
  // named inputs are read where they lie) or writing their pixel buffers (pinned destinations): no return may
  // leave that in flight — the caller is free to release both the moment jga_pipeline_run() is back — and no
  // buffer of the input cache may be let go (and evicted: unregistered) before the streams have drained.  Declared
  // in this order so that the drain runs first.
  struct held_inputs {
    input_cache &c;
    std::vector<const void *> v;
    ~held_inputs() { for (const void *p : v) c.release(p); }
  } held{pl->inputs, {}};
  struct drain_on_failure {
    hipStream_t st;
    bool armed = true;
    ~drain_on_failure() { if (armed) (void)hipStreamSynchronize(st); }
  } guard{l.stream};
  // How each file reaches the device (clean-up on the device only; jga_huff_set_input_flags): read where it lies by
  // a copy call that names it — the caller says it is pinned, or the input cache holds it registered (or registers it
  // now), or, with no cache, ordinary memory the runtime pins per copy — or copied into the pinned blob by this thread.
  std::vector<unsigned char> in_place((size_t)m, 0);
  // (pinned files keep the turns of round 5: named where they lie by the first groups, copied into the blob by a group
  // that waits for the link — 3.50-3.53 ms median per 128-file shard against 3.6-3.9 through the FIFO; ordinary files
  // take the FIFO: 3.5-3.9 against 4.0 with a registration per file and turns, profiles/r6_short_runs.md)
  bool all_pinned = true;
  for (int i = 0; i < m; i++) all_pinned = all_pinned && (jobv[i]->pinned & 1);
  // (a caller that asked for the persistent cache keeps round 5's route: its buffers are registered once and read
  // where they lie from then on)
  const bool fifo = short_run && shared && on_device && !pl->offload_cleanup && pl->short_fifo && pl->fifo_streams[0]
   && (!all_pinned || pl->short_fifo_pinned) && !pl->inputs.persistent;
  const unsigned ticket = fifo ? pl->run_ticket.fetch_add(1) : 0u;
  {
    for (int i = 0; i < m && on_device; i++) {
      if (fifo) in_place[(size_t)i] = (int)ticket < pl->fifo_named && (jobv[i]->pinned & 1);    // (everything else goes through the blob)
      else if (jobv[i]->pinned & 1) in_place[(size_t)i] = 1;
      else if (pl->inputs.acquire(jobv[i]->jpeg, (size_t)jobv[i]->size)) {
        in_place[(size_t)i] = 1;
        held.v.push_back(jobv[i]->jpeg);
      }
      // [MI355X] the headline on the 2 CPUs a rank of 8 gets: 131 Gpixel/s with named copies, 145 through the
      // cache, 86 with a host copy of every file into the pinned blob (rounds 2-3)
      else if (pl->cfg.input_cache_mb == -1) in_place[(size_t)i] = 1;
    }
    jga_huff_set_device_unstuff(l.hb, on_device);
    jga_huff_set_inputs_pinned(l.hb, 0);
    jga_huff_set_input_flags(l.hb, in_place.data(), m);
    if (fifo) jga_huff_set_threads(l.hb, pl->fifo_threads);       // (24 threads copying on a 16-CPU grant move LESS than 8)
    jga_huff_set_blocking_waits(l.hb, pl->blocking);
    jga_huff_set_device_shared(l.hb, shared ? (short_run ? 1 : 2) : 0);   // (2: a long run — throughput, not the last chain, is what counts)
    jga_huff_set_copy_stream(l.hb, fifo ? pl->fifo_streams[pl->fifo_nstreams > 1 ? ticket & 1u : 0u] : pl->copy_streams.empty() ? nullptr
     : pl->copy_streams[pl->copy_next.fetch_add(1)%pl->copy_streams.size()]);
  }
  // A lone image whose Huffman tables do not fit the device lookup format takes the host
  // entropy stage (csrc/entropy.c) instead; everything after it is the same.
  bool host_entropy = false;
  jpeg_header hdr;
  link_turn link(pl);
  jga_huff_set_upload_gate(l.hb, pl->link_slots > 0 && !fifo ? &link_turn::take_hook : nullptr, &link);
  // (short runs with turns — JGA_PIPE_SHORT_FIFO=0 in the tuning build: a group waiting for the link copies its
  // files into its blob meanwhile — JGA_PIPE_COPY_WAITING=0: never)
  if (pl->link_slots > 0 && !fifo && short_run && on_device && !pl->offload_cleanup && pl->copy_while_waiting) {
    jga_huff_set_upload_poll(l.hb, &link_turn::poll_hook);
  }
  const int prc = jga_huff_prepare(l.hb, ptrs.data(), sizes.data(), m, &g, l.stream);
  jga_huff_set_upload_gate(l.hb, nullptr, nullptr);
  if (link.held) {                                  // the upload is on its way: hold the turn until it is there
    (void)jga_huff_wait_upload(l.hb);
    link.give();
  }
  if (prc != EXIT_SUCCESS) {
    if (m != 1) return GROUP_REJECTED;              // run_lane looks at the per-member verdicts
    if (jga_huff_prepare_verdict(l.hb, 0) != 2) return EXIT_FAILURE;
    if (jga_parse_header(jobv[0]->jpeg, jobv[0]->size, &hdr) != EXIT_SUCCESS
     || jga_geom_from_header(&g, &hdr) != EXIT_SUCCESS) {
      return EXIT_FAILURE;
    }
    host_entropy = true;
  }
  const long long cstride = (g.coef_shorts + 127) & ~127ll;
  const long long out_bytes = rgb ? g.rgb_bytes : g.yuv_bytes;
  const long long ostride = (out_bytes + 255) & ~255ll;
  // Where the pixels go: the lane's own buffer, or the callers'.  Destinations that lie evenly
  // spaced (dev_out[i] = dev_out[0] + i*pitch — a caller that keeps every output hands the
  // pipeline slices of one big buffer) take one launch like the lane's buffer does; anything
  // else one launch per image.
  bool scattered = false, all_given = true, strided = false;
  long long pitch = ostride;
  for (int i = 0; i < m; i++) {
    scattered = scattered || jobv[i]->dev_out != nullptr;
    all_given = all_given && jobv[i]->dev_out != nullptr;
  }
  if (all_given) {
    if (m > 1) pitch = (long long)(jobv[1]->dev_out - jobv[0]->dev_out);
    strided = pitch >= out_bytes && pitch % 16 == 0 && ((uintptr_t)jobv[0]->dev_out) % 16 == 0;
    for (int i = 1; i < m && strided; i++) strided = jobv[i]->dev_out == jobv[0]->dev_out + pitch*i;
  }
  const long long dcstride = (g.coef_shorts/64 + 127) & ~127ll;
  auto reserve = [&](int cnt) {
    return grow((void **)&l.d_coef, &l.cap_coef, cstride*2*cnt, false)
     && grow((void **)&l.d_dc, &l.cap_dc, dcstride*2*cnt, false)
     && grow((void **)&l.d_q, &l.cap_q, 384ll*cnt, false)
     && (all_given || grow((void **)&l.d_out, &l.cap_out, ostride*cnt, false))
     && (!copy_back || grow((void **)&l.h_out, &l.cap_hout, ostride*cnt, true));
  };
  if (!reserve(full) && (full == m || !reserve(m))) return EXIT_FAILURE;   // (the full size first, this group's if that fails)
  device_turn turn(pl);
  turn.take((int)(((long long)m*g.width*g.height + 3840ll*2160 - 1)/(3840ll*2160)));   // (the group's upload is already in flight)
  const auto t_b = std::chrono::steady_clock::now();
  const double c_b = trace ? thread_cpu_ms() : 0.0;
  const unsigned short *d_q = l.d_q;                         // quantisers: they came up with prepare()'s descriptors
  if (host_entropy) {
    unsigned short q[192];
    memset(q, 0, sizeof(q));
    for (int p = 0; p < g.nplanes; p++) memcpy(q + 64*p, hdr.comp[p].quant->tbl, 128);
    if (!grow((void **)&l.h_coef, &l.cap_hcoef, cstride*2, true)) return EXIT_FAILURE;
    if (jga_entropy_decode(jobv[0]->jpeg, jobv[0]->size, &g, l.h_coef, 0) != EXIT_SUCCESS) return EXIT_FAILURE;
    if (!HOK(hipMemcpyAsync(l.d_q, q, sizeof(q), hipMemcpyHostToDevice, l.stream))) return EXIT_FAILURE;
    if (!HOK(hipStreamSynchronize(l.stream))) return EXIT_FAILURE;       // q is on this stack frame
    if (!HOK(hipMemcpyAsync(l.d_coef, l.h_coef, (size_t)g.coef_shorts*2, hipMemcpyHostToDevice, l.stream))) return EXIT_FAILURE;
  }
  else {
    d_q = jga_huff_qtabs_device(l.hb);
    if (!d_q) return EXIT_FAILURE;
    // (DC values in their own array: the block-decode kernel takes them from there)
    if (jga_huff_decode_split_begin(l.hb, l.d_coef, cstride, l.d_dc, dcstride, l.stream) != EXIT_SUCCESS) return EXIT_FAILURE;
  }
  const auto t_c = std::chrono::steady_clock::now();
  const double c_c = trace ? thread_cpu_ms() : 0.0;
  const short *dcv = host_entropy ? nullptr : l.d_dc;       // (the host entropy stage makes finished planes)
  // Everything behind the entropy decode: block decode into the pixels' place, copies back.  Queued right behind
  // the decode's first half — the lane waits ONCE per group — and again if the decode's second half says that
  // what was queued in between did not see the final planes (streams that need more rounds than were queued).
  auto queue_behind = [&]() -> bool {
    if (!scattered || strided) {
      unsigned char *base = strided ? jobv[0]->dev_out : l.d_out;
      const long long step = strided ? pitch : ostride;
      if ((rgb ? jga_idct_rgb_batch_dc(&g, m, l.d_coef, cstride, dcv, dcstride, d_q, 1, base, step, l.stream)
       : jga_idct_yuv_batch_dc(&g, m, l.d_coef, cstride, dcv, dcstride, d_q, 1, base, step, l.stream)) != EXIT_SUCCESS) {
        return false;
      }
    }
    else {
      for (int i = 0; i < m; i++) {
        unsigned char *dst = jobv[i]->dev_out ? jobv[i]->dev_out : l.d_out + ostride*i;
        if ((rgb ? jga_idct_rgb_batch_dc(&g, 1, l.d_coef + cstride*i, cstride, dcv ? dcv + dcstride*i : nullptr, dcstride, d_q + 192*i, 1, dst, ostride, l.stream)
         : jga_idct_yuv_batch_dc(&g, 1, l.d_coef + cstride*i, cstride, dcv ? dcv + dcstride*i : nullptr, dcstride, d_q + 192*i, 1, dst, ostride, l.stream)) != EXIT_SUCCESS) {
          return false;
        }
      }
    }
    return true;
  };
  auto queue_copies_back = [&](bool only_damaged) -> bool {
    for (int i = 0; i < m && copy_back; i++) {
      if (!jobv[i]->host_out) continue;
      if (only_damaged && jga_huff_image_error(l.hb, i) == 0) continue;
      const unsigned char *src = jobv[i]->dev_out ? jobv[i]->dev_out : l.d_out + ostride*i;
      unsigned char *to = (jobv[i]->pinned & 2) ? jobv[i]->host_out : l.h_out + ostride*i;   // pinned destination: no staging
      if (!HOK(hipMemcpyAsync(to, src, (size_t)out_bytes, hipMemcpyDeviceToHost, l.stream))) return false;
    }
    return true;
  };
  auto wait_lane = [&]() { return HOK(pl->blocking ? jga_stream_wait_sleeping(l.stream, l.done) : hipStreamSynchronize(l.stream)); };
  if (!queue_behind() || !queue_copies_back(false)) return EXIT_FAILURE;
  bool damaged = false;
  if (host_entropy) {
    if (!wait_lane()) return EXIT_FAILURE;
  }
  else {
    // damaged data in some members does not spoil the others' planes: go on, and report the
    // damaged ones alone (anything else — a launch failure — fails the group)
    int valid_behind = 0;
    if (jga_huff_decode_split_end(l.hb, &valid_behind) != EXIT_SUCCESS) {
      if (jga_huff_image_errors(l.hb) <= 0) return EXIT_FAILURE;
      damaged = true;
    }
    if (!valid_behind) {
      if (!queue_behind() || !queue_copies_back(false) || !wait_lane()) return EXIT_FAILURE;
    }
  }
  if (damaged) {
    // a damaged member's planes hold whatever its lanes reached plus leftovers of earlier groups in this
    // lane's buffers: its pixels are not handed out — the job fails with a zeroed output
    for (int i = 0; i < m; i++) {
      if (jga_huff_image_error(l.hb, i) == 0) continue;
      unsigned char *dst = jobv[i]->dev_out ? jobv[i]->dev_out : strided ? jobv[0]->dev_out + pitch*i : l.d_out + ostride*i;
      if (!HOK(hipMemsetAsync(dst, 0, (size_t)out_bytes, l.stream))) return EXIT_FAILURE;
    }
    if (!queue_copies_back(true) || !wait_lane()) return EXIT_FAILURE;
  }
  guard.armed = false;                 // the streams hold nothing of this group any more
  turn.give();
  if (trace) {
    const auto t_d = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count(); };
    const double c_d = thread_cpu_ms();
    fprintf(stderr, "lane group of %d: began at %.2f ms; prepare + wait for a device slot %.2f ms (%.2f of this thread's CPU), "
     "entropy decode queued %.2f ms (%.2f), block decode queued + the group's one wait %.2f ms (%.2f); done at %.2f ms\n",
     m, ms(pl->run_t0, t_a), ms(t_a, t_b), c_b - c_a, ms(t_b, t_c), c_c - c_b, ms(t_c, t_d), c_d - c_c, pl->since_run_start_ms());
  }
  const long long up = host_entropy ? g.coef_shorts*2 : jga_huff_upload_bytes(l.hb)/m;
  if (copy_back) {
    // pinned staging -> the callers' buffers, spread over the lane's thread share (one thread
    // moves ~8 GB/s; a group of 24 x 4K RGB frames is 600 MB)
    std::atomic<int> next(0);
    auto mover = [&]() {
      for (int i = next.fetch_add(1); i < m; i = next.fetch_add(1)) {
        if (jobv[i]->host_out && !(jobv[i]->pinned & 2)) memcpy(jobv[i]->host_out, l.h_out + ostride*i, (size_t)out_bytes);
      }
    };
    const int movers = threads < m ? (threads < 1 ? 1 : threads) : m;
    std::vector<std::thread> team;
    for (int t = 1; t < movers; t++) team.emplace_back(mover);
    mover();
    for (auto &t : team) t.join();
  }
  for (int i = 0; i < m; i++) {
    jobv[i]->width = g.width; jobv[i]->height = g.height; jobv[i]->nplanes = g.nplanes;
    jobv[i]->h2d_bytes = up;
    const bool read_by_host = host_entropy || !in_place[(size_t)i] || jga_huff_image_copied(l.hb, i) == 1;
    jobv[i]->host_bytes = read_by_host ? jobv[i]->size : 0;
    (read_by_host ? pl->inputs.n_copied : pl->inputs.n_in_place)++;
    pl->inputs.host_bytes += jobv[i]->host_bytes;
    jobv[i]->status = (damaged && jga_huff_image_error(l.hb, i) != 0) ? EXIT_FAILURE : EXIT_SUCCESS;
  }
  return EXIT_SUCCESS;
}

// Frame size + sampling of a JPEG without a full parse: (Y<<48 | X<<32 | Nf<<24 | sampling
// bytes), 0 if no SOF0 is found.  Jobs are bucketed by it so that one batch of the GPU
// entropy stage holds one geometry even when the stream of jobs mixes sizes.
uint64_t geometry_key(const unsigned char *p, int size) {
  int i = 2;
  if (size < 4 || p[0] != 0xFF || p[1] != 0xD8) return 0;
  while (i + 4 <= size) {
    if (p[i] != 0xFF) return 0;
    const int m = p[i + 1];
    if (m == 0xFF) { i++; continue; }
    const int len = (p[i + 2] << 8) | p[i + 3];
    if (m == 0xC0) {
      if (i + 2 + len > size || len < 11) return 0;
      const unsigned char *f = p + i + 4;           // P, Y, X, Nf, then (C, HV, Tq) per component
      uint64_t k = ((uint64_t)((f[1] << 8) | f[2]) << 48) | ((uint64_t)((f[3] << 8) | f[4]) << 32)
       | ((uint64_t)f[5] << 24);
      for (int c = 0; c < f[5] && c < 3 && 8 + 3*c < len - 2; c++) k |= (uint64_t)f[7 + 3*c] << (8*c);
      return k | 1ull << 31;                        // (never 0; Nf only needs bits 24..26)
    }
    if (m == 0xDA || m == 0xD9) return 0;
    i += 2 + len;
  }
  return 0;
}

// How a run cuts its jobs into groups (transport 2) — host logic only, no device involved:
// bucket the jobs by geometry, in arrival order.  `batch` counts 4K frames; smaller frames fill
// a group to about the same number of pixels (a launch takes at least one run's latency however
// little it decodes).  A job too short to keep every lane busy for several groups (1024 1080p
// files are 256 frame equivalents: ONE group of 32 per lane) is cut finer, so that uploads,
// entropy stage and block decode of different groups overlap: about groups_per_lane groups
// per lane, none below min_group_eq frame equivalents.
struct plan_params { int lanes, batch, groups_per_lane, min_group_eq; bool ramp_first; int short_ramp = 0; };
void plan_groups(const plan_params &pp, const jga_job *jobs, int n, std::vector<std::vector<int>> &groups) {
  const int nl = pp.lanes, batch = pp.batch;
  std::unordered_map<uint64_t, long long> pixels;            // geometry -> pixels of all its jobs
  std::vector<uint64_t> keys((size_t)n);
  for (int i = 0; i < n; i++) {
    keys[(size_t)i] = geometry_key(jobs[i].jpeg, jobs[i].size);
    if (keys[(size_t)i]) {
      pixels[keys[(size_t)i]] += (long long)((keys[(size_t)i] >> 48) & 0xffff)*(long long)((keys[(size_t)i] >> 32) & 0xffff);
    }
  }
  const long long frame = 3840ll*2160;
  std::unordered_map<uint64_t, size_t> open;      // geometry -> its group still filling up
  std::unordered_map<uint64_t, int> made_groups;  // geometry -> groups closed so far
  for (int i = 0; i < n; i++) {
    const uint64_t key = keys[(size_t)i];
    auto it = key ? open.find(key) : open.end();
    if (it == open.end()) {
      groups.emplace_back();
      groups.back().reserve(key ? batch : 1);
      if (key) it = open.emplace(key, groups.size() - 1).first;
      else { groups.back().push_back(i); continue; }      // unparsable: fails on its own
    }
    groups[it->second].push_back(i);
    const long long px = (long long)((key >> 48) & 0xffff)*(long long)((key >> 32) & 0xffff);
    long long eq = (pixels[key]/frame + (long long)pp.groups_per_lane*nl - 1)/((long long)pp.groups_per_lane*nl);
    if (eq < pp.min_group_eq) eq = pp.min_group_eq;
    if (eq > batch) eq = batch;
    long long cap = px > 0 ? (eq*frame + px/2)/px : eq;          // images of this size per group
    cap = cap < 1 ? 1 : cap > 16ll*batch ? 16ll*batch : cap;
    // The first group of every lane of a LONG job (three groups per lane and more): a fraction of
    // a group, rising from lane to lane — all lanes start preparing at the same moment, and the
    // link has nothing to do until the first of them is through (4 ms for 48 4K frames).
    // [MI355X] 1536 x 4K 129-132 -> 134-136 Gpixel/s; a short job's few groups stay equal (the
    // 128-file shard: 4.25 ms against 4.57 with its eight groups ramped).
    int &made = made_groups[key];
    if (pp.ramp_first && made < nl && nl > 1 && pixels[key] >= 3ll*nl*cap*px) {
      cap = (cap*(made + 1) + nl - 1)/nl;
      if (cap < 1) cap = 1;
    }
    else if (pp.short_ramp && px > 0) {
      // A SHORT job (one group per lane): what it waits for is the link — every byte has to cross before the last
      // chain of kernels can start — so the first groups are small (their bytes are staged and crossing while the
      // bigger ones are still being staged) and rise to an even size: weights min(k + 1, 0.7 G + 0.15) over G groups
      // (128 files in 8 groups: 4, 8, 12, 16, 20, 23, 23, 22).
      const long long count = pixels[key]/px, G = (count + cap - 1)/cap;
      if (G >= 4 && G <= nl) {
        const double top = 0.7*(double)G + 0.15;
        // (2: the same weights from the other end — big groups first, so that the group that arrives LAST is small and
        // its chain of kernels, what the run ends with, short)
        auto weight = [&](long long k) { const long long j = pp.short_ramp == 2 ? G - 1 - k : k; return (double)(j + 1) < top ? (double)(j + 1) : top; };
        double sum = 0, upto = 0;
        for (long long k = 0; k < G; k++) sum += weight(k);
        for (long long k = 0; k <= made && k < G; k++) upto += weight(k);
        const double before = upto - weight(made < G ? made : G - 1);
        const long long a = (long long)((double)count*before/sum + 0.5), b = made + 1 >= G ? count : (long long)((double)count*upto/sum + 0.5);
        cap = b - a < 1 ? 1 : b - a;
      }
    }
    if ((long long)groups[it->second].size() >= cap) { open.erase(it); made++; }
  }
}

void lane_groups(jga_pipeline *pl, hlane *l, std::vector<std::vector<jga_job *>> *groups,
 std::atomic<int> *next, int threads) {
  const bool long_run = groups->size() > 2*pl->lanes.size();       // every lane sees several groups
  for (;;) {
    const int gi = next->fetch_add(1);
    if (gi >= (int)groups->size()) break;
    std::vector<jga_job *> &grp = (*groups)[gi];
    const int m = (int)grp.size();
    const int rc = lane_group(pl, *l, grp.data(), m, threads, groups->size() > 1, long_run, !long_run);
    if (rc == EXIT_SUCCESS || m == 1) continue;
    // One member with an unparsable header, or with Huffman tables outside the device lookup
    // format, must not cost the other 47 their batch: the members prepare() found usable go
    // again as one group, only the others are taken singly (a lone member with verdict 2
    // gets the host entropy stage inside lane_group).
    std::vector<jga_job *> good, rest;
    for (int i = 0; i < m; i++) {
      (rc == GROUP_REJECTED && jga_huff_prepare_verdict(l->hb, i) == 0 ? good : rest).push_back(grp[i]);
    }
    if (!rest.empty() && good.size() > 1
     && lane_group(pl, *l, good.data(), (int)good.size(), threads, true, false, !long_run) == EXIT_SUCCESS) {
      good.clear();
    }
    for (jga_job *j : good) (void)lane_group(pl, *l, &j, 1, 1, true, false, !long_run);
    for (jga_job *j : rest) (void)lane_group(pl, *l, &j, 1, 1, true, false, !long_run);
  }
}

// A lane's thread: sleeps until jga_pipeline_run() posts a run, works through its groups with the
// other lanes, reports back.
void run_lane(jga_pipeline *pl, hlane *l) {
  const bool dev_ok = HOK(hipSetDevice(pl->cfg.device));
  // the naps of host_wait.h are tens of microseconds: the default 50 us of timer slack would
  // more than double them
  if (!jga_tune("JGA_PIPE_KEEP_TIMERSLACK")) (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0UL, 0UL, 0UL);
  unsigned long long seen = 0;
  for (;;) {
    std::vector<std::vector<jga_job *>> *groups;
    std::atomic<int> *next;
    int threads;
    {
      std::unique_lock<std::mutex> lk(pl->run_mutex);
      pl->run_cv.wait(lk, [&] { return pl->quit || pl->run_gen != seen; });
      if (pl->quit) return;
      seen = pl->run_gen;
      groups = pl->run_groups; next = pl->run_next; threads = pl->run_threads;
    }
    if (dev_ok) lane_groups(pl, l, groups, next, threads);      // (jobs keep their EXIT_FAILURE otherwise)
    {
      std::lock_guard<std::mutex> lk(pl->run_mutex);
      if (++pl->run_done == (int)pl->lanes.size()) pl->done_cv.notify_one();
    }
  }
}

}  // namespace

extern "C" {

JGA_EXPORT jga_pipeline *jga_pipeline_create(const jga_pipeline_config *cfg) {
  // (read the two size fields before anything else: a caller built against another revision of
  // the header has something else — or nothing — where the later fields are)
  if (cfg->struct_size != (int)sizeof(jga_pipeline_config) || cfg->job_size != (int)sizeof(jga_job)) {
    jga_fail("pipeline: caller built against another revision of jpeg_gpu_amd.h (config %d bytes, job %d; "
     "this library: %d, %d) - use jga_pipeline_config_init()", cfg->struct_size, cfg->job_size,
     (int)sizeof(jga_pipeline_config), (int)sizeof(jga_job));
    return nullptr;
  }
  jga_pipeline *pl = new jga_pipeline();
  pl->cfg = *cfg;
  if (pl->cfg.transport < 0 || pl->cfg.transport > 2) {
    jga_fail("pipeline: transport must be 0 (planes), 1 (PACK) or 2 (GPU entropy stage)");
    delete pl;
    return nullptr;
  }
  if (pl->cfg.out != JPEG_DECODE_YUV && pl->cfg.out != JPEG_DECODE_RGB) {
    jga_fail("pipeline: out must be JPEG_DECODE_YUV or JPEG_DECODE_RGB");
    delete pl;
    return nullptr;
  }
  if (pl->cfg.nthreads <= 0) {
    // the workers also wait (for their slot's event, for a device turn): a few more of them
    // than CPUs we can keep busy, but nowhere near one per visible CPU of a throttled box
    const int cpus = jga_cpu_budget();
    pl->cfg.nthreads = pl->cfg.transport == 2 ? cpus + cpus/2 : 3*cpus;
    if (pl->cfg.nthreads > 96) pl->cfg.nthreads = 96;
  }
  const sched_knobs K = resolve_knobs(pl->cfg);
  pl->trace = K.trace;
  {
    // how many cores the host side can count on: what the process is granted, or fewer if the
    // caller asked for fewer threads (a rank that shares its grant with seven others does)
    const int cpus = jga_cpu_budget();
    pl->offload_cleanup = (cpus < pl->cfg.nthreads ? cpus : pl->cfg.nthreads) <= K.offload_at;
  }
  if (!hip_ok(hipSetDevice(pl->cfg.device), "hipSetDevice")) {
    delete pl;
    return nullptr;
  }
  if (pl->cfg.transport == 2) {
    pl->lanes.resize(K.lanes);
    pl->dev_slots = K.dev_slots;
    pl->dev_capacity = pl->dev_free = pl->dev_slots*K.batch;
    pl->blocking = K.blocking;
    pl->groups_per_lane = K.groups_per_lane;
    pl->min_group_eq = K.min_group_eq;
    pl->ramp_first = K.ramp_first;
    pl->link_slots = pl->link_free = K.link_slots;
    if (pl->cfg.input_cache_mb >= 0) {
      // (run-scoped registrations: the bound is on what the run's groups hold at one time — eight lanes' groups of
      // 32 x 4K files are 0.8 GB; what does not fit is copied)
      pl->inputs.cap = (size_t)(pl->cfg.input_cache_mb > 0 ? pl->cfg.input_cache_mb : 4096) << 20;
      pl->inputs.persistent = pl->cfg.input_cache_mb > 0;
      pl->inputs.sight = pl->inputs.persistent && pl->cfg.input_cache_sight > 0 ? pl->cfg.input_cache_sight : 1;
    }
    if (const char *e = jga_tune("JGA_PIPE_COPY_WAITING")) pl->copy_while_waiting = atoi(e) != 0;
    if (const char *e = jga_tune("JGA_PIPE_SHORT_FIFO")) { pl->short_fifo = atoi(e) != 0; pl->short_fifo_pinned = atoi(e) > 1; }
    if (const char *e = jga_tune("JGA_PIPE_FIFO_STREAMS")) pl->fifo_nstreams = atoi(e) > 1 ? 2 : 1;
    if (const char *e = jga_tune("JGA_PIPE_FIFO_THREADS")) pl->fifo_threads = atoi(e) > 0 ? atoi(e) : 1;
    if (const char *e = jga_tune("JGA_PIPE_FIFO_NAMED")) pl->fifo_named = atoi(e);
    pl->short_ramp = K.short_ramp;
    for (int i = 0; i < 2 && pl->short_fifo; i++) {
      if (!hip_ok(hipStreamCreateWithFlags(&pl->fifo_streams[i], hipStreamNonBlocking), "hipStreamCreate")) {
        jga_pipeline_destroy(pl);
        return nullptr;
      }
    }
    for (int i = 0; i < K.copy_streams; i++) {                 // (measured: profiles/r3_pipe_sweep.txt)
      hipStream_t cs = nullptr;
      if (!hip_ok(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "hipStreamCreate")) {
        jga_pipeline_destroy(pl);
        return nullptr;
      }
      pl->copy_streams.push_back(cs);
    }
    for (auto &l : pl->lanes) {
      if (!hip_ok(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking), "hipStreamCreate")
       || !hip_ok(hipEventCreateWithFlags(&l.done, hipEventDisableTiming), "hipEventCreate")) {
        jga_pipeline_destroy(pl);
        return nullptr;
      }
    }
    for (auto &l : pl->lanes) pl->lane_threads.emplace_back(run_lane, pl, &l);
    return pl;
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
  for (int i = 0; i < n; i++) jobs[i].status = EXIT_FAILURE;
  if (pl->cfg.transport == 2) {
    const bool trace = pl->trace;
    pl->run_t0 = std::chrono::steady_clock::now();
    const int batch = pl->cfg.batch > 0 ? pl->cfg.batch : 48;
    const int nl = (int)pl->lanes.size();
    std::vector<std::vector<jga_job *>> groups;
    {
      std::vector<std::vector<int>> plan;
      plan_groups({nl, batch, pl->groups_per_lane, pl->min_group_eq, pl->ramp_first != 0, pl->short_ramp}, jobs, n, plan);
      groups.resize(plan.size());
      for (size_t k = 0; k < plan.size(); k++) {
        groups[k].reserve(plan[k].size());
        for (int i : plan[k]) groups[k].push_back(&jobs[i]);
      }
    }
    // host threads of a lane's prepare: the run's share per lane that has a group to work on
    const int busy = (int)groups.size() < nl ? ((int)groups.size() > 0 ? (int)groups.size() : 1) : nl;
    int per = pl->cfg.nthreads/busy;
    if (per < 1) per = 1;
    if (trace) fprintf(stderr, "run: %d jobs in %d groups at %.2f ms\n", n, (int)groups.size(), pl->since_run_start_ms());
    {
      std::lock_guard<std::mutex> lk(pl->run_mutex);
      pl->run_groups = &groups; pl->run_next = &next; pl->run_threads = per;
      pl->run_done = 0;
      pl->run_ticket.store(0);
      pl->run_gen++;
    }
    pl->run_cv.notify_all();
    if (trace) fprintf(stderr, "run: posted to %d lanes at %.2f ms\n", nl, pl->since_run_start_ms());
    {
      std::unique_lock<std::mutex> lk(pl->run_mutex);
      pl->done_cv.wait(lk, [&] { return pl->run_done == nl; });
    }
    if (trace) fprintf(stderr, "run: lanes reported back at %.2f ms\n", pl->since_run_start_ms());
    for (int i = 0; i < n; i++) failed += jobs[i].status != EXIT_SUCCESS;
    return failed ? EXIT_FAILURE : EXIT_SUCCESS;
  }
  const int nt = n < (int)pl->workers.size() ? n : (int)pl->workers.size();
  for (int t = 0; t < nt; t++) {
    threads.emplace_back(run_worker, pl, &pl->workers[t], jobs, n, &next);
  }
  for (auto &th : threads) th.join();
  for (int i = 0; i < n; i++) failed += jobs[i].status != EXIT_SUCCESS;
  return failed ? EXIT_FAILURE : EXIT_SUCCESS;
}

// The plan jga_pipeline_run() would make for these jobs on a transport-2 pipeline of `lanes` lanes and
// `batch` 4K-frame equivalents per group (0: the defaults): group_of[i] = the group job i goes to.
// Returns the number of groups.  Host logic only: no device is touched.
JGA_EXPORT int jga_pipeline_plan_cfg(const jga_pipeline_config *cfg, const jga_job *jobs, int n, int *group_of) {
  std::vector<std::vector<int>> plan;
  const sched_knobs K = resolve_knobs(*cfg);     // (what jga_pipeline_create makes of the same configuration)
  plan_groups({K.lanes, K.batch, K.groups_per_lane, K.min_group_eq, K.ramp_first, K.short_ramp}, jobs, n, plan);
  for (size_t k = 0; k < plan.size(); k++) for (int i : plan[k]) group_of[i] = (int)k;
  return (int)plan.size();
}
JGA_EXPORT int jga_pipeline_plan(int lanes, int batch, const jga_job *jobs, int n, int *group_of) {
  jga_pipeline_config cfg;
  jga_pipeline_config_init(&cfg);
  cfg.depth = lanes;
  cfg.batch = batch;
  return jga_pipeline_plan_cfg(&cfg, jobs, n, group_of);
}

JGA_EXPORT int jga_pipeline_register_input(jga_pipeline *pl, const unsigned char *jpeg, int size) {
  if (!pl->inputs.enabled() || !pl->inputs.persistent) {
    return jga_fail("pipeline: no persistent input cache (jga_pipeline_config.input_cache_mb must be > 0)");
  }
  if (!hip_ok(hipSetDevice(pl->cfg.device), "hipSetDevice")) return EXIT_FAILURE;
  if (!pl->inputs.acquire(jpeg, (size_t)(size > 0 ? size : 0), false)) {
    return jga_fail("pipeline: could not register %d bytes at %p (cache of %d MB, buffers under 64 KB are never registered)",
     size, (const void *)jpeg, (int)(pl->inputs.cap >> 20));
  }
  pl->inputs.release(jpeg);
  return EXIT_SUCCESS;
}
JGA_EXPORT int jga_pipeline_forget_input(jga_pipeline *pl, const unsigned char *jpeg) {
  (void)hipSetDevice(pl->cfg.device);
  if (pl->inputs.forget(jpeg) != EXIT_SUCCESS) return jga_fail("pipeline: %p is in use by a running group", (const void *)jpeg);
  return EXIT_SUCCESS;
}
JGA_EXPORT int jga_pipeline_counters(const jga_pipeline *pl, long long *out, int n) {
  const input_cache &c = pl->inputs;
  const bool on_device = pl->cfg.unstuff == 2 || (pl->cfg.unstuff == 0 && pl->offload_cleanup);
  const long long v[9] = {c.n_registered.load(), (long long)(c.held >> 20), c.n_in_place.load(), c.n_copied.load(),
   c.n_evicted.load(), c.us_register.load(), on_device ? 1 : 0, c.host_bytes.load(), c.n_stale.load()};
  for (int i = 0; i < n && i < 9; i++) out[i] = v[i];
  return 9;
}

JGA_EXPORT void jga_pipeline_destroy(jga_pipeline *pl) {
  if (!pl) return;
  (void)hipSetDevice(pl->cfg.device);
  {
    std::lock_guard<std::mutex> lk(pl->run_mutex);
    pl->quit = true;
  }
  pl->run_cv.notify_all();
  for (auto &th : pl->lane_threads) th.join();
  for (auto &l : pl->lanes) free_lane(l);
  pl->inputs.clear();
  for (auto cs : pl->copy_streams) (void)hipStreamDestroy(cs);
  for (auto cs : pl->fifo_streams) if (cs) (void)hipStreamDestroy(cs);
  for (auto &w : pl->workers) {
    free_slot(w.slots[0]);
    free_slot(w.slots[1]);
    if (w.stream) (void)hipStreamDestroy(w.stream);
  }
  delete pl;
}

}  // extern "C"
