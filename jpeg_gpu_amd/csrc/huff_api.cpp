// huff_api.cpp — C-ABI of the GPU entropy stage (BASELINE config 5).
//   jga_huff_create / _prepare / _decode / _destroy
// prepare(): host parses the marker segments of a batch of same-geometry JPEGs,
//   builds device tables / restart segments / initial lane states and uploads them
//   together with the entropy-coded bytes (compressed: ~0.4 B/px instead of 3 B/px).
// decode(): everything on the GPU: synchronisation rounds until no lane moves, per-segment
//   prefix sums (which also zero the lines of blocks two lanes share), write pass, DC pass.
//   The output is the same packed QUANT-stage buffer jga_entropy_decode() produces on the
//   host; no plane is cleared beforehand (every slot that holds a block is written).
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <chrono>
#include <thread>
#include <time.h>
#include <vector>
#include "huff_kernels.h"
#include "host_wait.h"
#include "huff_prepare.h"
#include "unstuff_kernels.h"

#define HOK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return jga_fail("huff: HIP error %d (%s) at %s", (int)e_, hipGetErrorString(e_), #call); } while (0)

struct jga_huff_batch {
  int max_images;
  long long max_scan;
  // host staging (pinned)
  unsigned char *h_blob;       // scan | images | segs | tables: one upload (device blob: + sub_seg | S)
  size_t blob_cap;
  // device
  unsigned char *d_blob;
  uint64_t *d_last_in;
  uint32_t *d_R;
  uint32_t *d_B;
  uint32_t *d_list;            // work lists of the list rounds: two parities of sub_cap entries
  uint32_t *d_lcount;          // [4][max_images] their lengths, HJ_LIST_CSTRIDE words apart
  int16_t *d_dc;               // DC differences (scan order) | DC values (by buffer slot) of the current batch
  size_t dc_cap;               // entries of each half
  uint32_t *d_blkpos;          // small batches: where every block starts (hj_block_starts -> hj_write_blocks), dc_cap entries
  size_t blkpos_cap;
  uint32_t *d_dcpart;          // chunk totals of the DC prefix sums
  size_t dcpart_cap;           // entries (uint32)
  uint32_t max_seg_mcus, max_segs_image;
  uint32_t *d_ran, *d_errors;
  uint32_t *d_part;            // chunk totals of the prefix-sum pass
  int sub_log2, force_sub_log2;  // subsequence length of the current batch / JGA_HUFF_SUB
  uint32_t *h_ran;             // pinned readback
  uint64_t *h_states;          // pinned, lazily: S and last_in read back for assist_chains()
  size_t h_states_cap;         // entries
  int last_assisted;           // subsequences the host walked in the last decode
  int image_errors;            // images of the last decode whose data was damaged
  int assist_hint;             // the previous decode needed the host walk
  unsigned long long assist_left;   // list entries left at the previous look (decode_end: is the decode still getting anywhere?)
  int spec_rounds;             // rounds queued before the speculative tail (0: not yet decided)
  hipEvent_t ev_wait;          // hipEventBlockingSync: host waits that sleep instead of spinning
  int blocking_waits;
  int device_shared;                    // other decodes run beside this one (jga_huff_set_device_shared)
  void (*before_upload)(void *, long long, int);   // called right before prepare() queues its upload (jga_huff_set_upload_gate)
  void *before_upload_arg;
  int (*upload_poll)(void *);                       // "may I upload now?" (jga_huff_set_upload_poll)
  hipStream_t copy_stream;     // uploads go here (in the order they are queued), the caller's stream waits for them
  hipEvent_t ev_up;
  size_t sub_cap;
  // current batch
  int nimages;
  uint32_t total_sub, total_seg, max_nsub;
  size_t off_images, off_segs, off_subseg, off_tables, off_S, off_scan, blob_size, upload_size, scan_bytes;
  int prepare_threads;
  // on-device unstuffing (unstuff_kernels.hip): the raw scans go up, the clean streams and the
  // segment tables are made in HBM; the host-side copies assist_chains() needs are fetched lazily
  int device_unstuff, unstuffed_on_device;
  int inputs_pinned;           // callers' JPEG buffers are pinned/registered: DMA the scans straight from them
  long long host_bytes;        // bytes of the callers' files the last prepare() read on the host
  int assist_after, speculate, trace;   // jga_huff_set_option (0: defaults)
  hipEvent_t arrived;          // the event that says "the last prepare()'s bytes are on the device"
  bool qtab_on_device;         // the last prepare() put the quantisers into the blob (jga_huff_qtabs_device)
  size_t off_qtab;
  bool wide;                   // the last prepare() built and uploaded the 12-bit AC tables: one set per image (a small batch on a
  bool wide_shared;            // device of its own: every round takes them), or ONE set for a batch whose images all bring the same
                               // Huffman tables (the list rounds of a batch that fills or shares the device take it)
  size_t off_wide;
  // a decode in two halves (jga_huff_decode_split_begin / _end): what _begin queued
  struct {
    bool active, with_tail;
    short *d_coef, *d_dc;
    long long coef_stride, dc_stride;
    hipStream_t st;
    int round;
  } pend;
  size_t off_raw, off_uimg, off_part, off_bnd, off_info, off_perr;
  int max_chunks;
  jga_geom geom;
  std::vector<unsigned short> qtab;
  std::vector<unsigned char> verdict;   // last prepare(), per image: 0 ok, 1 unusable, 2 host entropy stage
  std::vector<unsigned char> shadow;    // device unstuffing: images | segs | clean scans read back (blob offsets)
  std::vector<unsigned char> copied;        // last prepare(), per image: 1 if a host core copied its scan into the blob
  std::vector<unsigned char> input_flags;   // per image: buffer pinned / registered (jga_huff_set_input_flags); empty: inputs_pinned for all
  int last_rounds;
  int list_state;              // work lists of the list rounds: 0 valid, 1 to be built (counters zero), 2 to be reset and built
};

// Same frame as far as every kernel is concerned: everything in a jga_geom but the restart
// interval (which only shapes the segments).  Two samplings can agree in size, subsamp class
// and plane total and still differ per plane (Y 4x2 / C 1x1 vs Y 2x4 / C 1x1).
static bool same_geometry(const jga_geom &a, const jga_geom &b) {
  if (a.width != b.width || a.height != b.height || a.nplanes != b.nplanes || a.subsamp != b.subsamp
   || a.w0 != b.w0 || a.nhmb != b.nhmb || a.nvmb != b.nvmb || a.coef_shorts != b.coef_shorts
   || a.coef_blocks != b.coef_blocks || a.yuv_bytes != b.yuv_bytes || a.rgb_bytes != b.rgb_bytes) {
    return false;
  }
  for (int p = 0; p < a.nplanes; p++) {
    const jga_plane_geom &x = a.plane[p], &y = b.plane[p];
    if (x.hblocks != y.hblocks || x.vblocks != y.vblocks || x.xdec != y.xdec || x.ydec != y.ydec
     || x.cstride != y.cstride || x.qidx != y.qidx || x.coef_off != y.coef_off
     || x.data_off != y.data_off) {
      return false;
    }
  }
  return true;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1)/a*a; }

// The 12-bit AC tables (hj_wide_ac, huff_common.h) are for batches that leave most of the device empty and have it
// to themselves: a few images whose first round is at most one workgroup per CU (82 KB of LDS each).
#define HJ_WIDE_MAX_IMAGES 4
static bool wants_wide(const jga_huff_batch *b, int n, uint64_t scan_bytes, int sub_log2) {
  if (b->device_shared || n > HJ_WIDE_MAX_IMAGES || jga_tune("JGA_HUFF_NO_WIDE")) return false;
  return (scan_bytes >> sub_log2) + 256u*(uint64_t)n <= 256u*256u;     // <= 256 workgroups of 256 subsequences: one per CU
}

// ... and for every other batch whose images all bring the same tables (one encoder: the usual case) ONE set goes up
// with the descriptors (32 KB) for the list rounds, whose workgroups are few: a quarter fewer look-ups per step of the
// chain.  Decides b->wide / b->wide_shared and builds what is wanted.
static void choose_wide(jga_huff_batch *b, std::vector<hj_prepared> &prep, int n, uint64_t scan_bytes) {
  b->wide = wants_wide(b, n, scan_bytes, b->sub_log2);
  b->wide_shared = false;
  if (b->wide) {
    for (int i = 0; i < n; i++) hj_prepare_wide(&prep[(size_t)i]);
    return;
  }
  if (jga_tune("JGA_HUFF_NO_WIDE") || jga_tune("JGA_HUFF_NO_SHARED_WIDE")) return;
  // (not in a long run of a pipeline, device_shared == 2: there the 85 KB of LDS a list workgroup with these tables
  // takes are the other lanes' dense and write workgroups' — 1080p in a steady state 130-135 Gpixel/s with them,
  // 139-141 without; a SHORT run, which waits for its last group's chain, keeps them: the 128-file shard 3.35-3.54 ms
  // median with, 3.65-3.70 without: tools/archive/r5_shared_wide_pipeline.sh)
  if (b->device_shared == 2) return;
  for (int i = 1; i < n; i++) {
    if (memcmp(&prep[(size_t)i].tabs, &prep[0].tabs, sizeof(hj_tables)) != 0) return;
  }
  hj_prepare_wide(&prep[0]);
  b->wide = b->wide_shared = true;
}

extern "C" {

JGA_EXPORT jga_huff_batch *jga_huff_create(int max_images, long long max_scan_bytes) {
  jga_huff_batch *b = new jga_huff_batch();
  memset(static_cast<void *>(b), 0, offsetof(jga_huff_batch, qtab));
  b->max_images = max_images;
  b->max_scan = max_scan_bytes;
  // (grow_batch covers streams cut into very many restart intervals)
  // (sized for the shortest subsequences a batch of this capacity can be given)
  b->sub_cap = (size_t)(max_scan_bytes >> hj_choose_sub_log2((uint64_t)max_scan_bytes, 1, 1)) + (size_t)max_images*4096 + 1024;
  if (const char *e = jga_tune("JGA_HUFF_SUB")) (void)jga_huff_set_option(b, JGA_HUFF_OPT_SUB_BYTES, atoi(e));
  if (const char *e = jga_tune("JGA_HUFF_DEVICE_UNSTUFF")) b->device_unstuff = atoi(e) != 0;
  if (const char *e = jga_tune("JGA_HUFF_ASSIST_AFTER")) b->assist_after = atoi(e) > 0 ? atoi(e) : 1;
  if (const char *e = jga_tune("JGA_HUFF_SPECULATE")) b->speculate = atoi(e) == 0 ? -1 : 0;
  if (jga_tune("JGA_PIPE_TRACE")) b->trace = 1;
  const size_t seg_cap = b->sub_cap;   // worst case one segment per subsequence
  b->blob_cap = align_up(sizeof(hj_image)*max_images, 256) + align_up(sizeof(hj_segment)*seg_cap, 256)
   + align_up(4*b->sub_cap, 256) + align_up(sizeof(hj_tables)*max_images, 256)
   + align_up(8*(b->sub_cap + seg_cap), 256) + align_up((size_t)max_scan_bytes + 64*max_images, 256)
   + align_up(384*(size_t)max_images, 256) + align_up(sizeof(hj_wide_ac)*HJ_WIDE_MAX_IMAGES, 256);
  bool ok = hipHostMalloc((void **)&b->h_blob, b->blob_cap, hipHostMallocDefault) == hipSuccess
   && hipMalloc((void **)&b->d_blob, b->blob_cap) == hipSuccess
   && hipMalloc((void **)&b->d_last_in, 8*b->sub_cap) == hipSuccess
   && hipMalloc((void **)&b->d_R, 4*b->sub_cap) == hipSuccess
   && hipMalloc((void **)&b->d_B, 4*b->sub_cap) == hipSuccess
   && hipMalloc((void **)&b->d_list, 8*b->sub_cap) == hipSuccess
   && hipMalloc((void **)&b->d_lcount, 16*HJ_LIST_CSTRIDE*(size_t)max_images) == hipSuccess
   && hipMalloc((void **)&b->d_part, hj_scan_part_bytes(b->sub_cap, b->sub_cap)) == hipSuccess
   && hipMalloc((void **)&b->d_ran, 4*HJ_MAX_ROUNDS + 4*(size_t)max_images) == hipSuccess    // (+ d_errors: one copy brings both back)
   && hipHostMalloc((void **)&b->h_ran, 4*HJ_MAX_ROUNDS + 4*(size_t)max_images, hipHostMallocDefault) == hipSuccess
   && hipEventCreateWithFlags(&b->ev_wait, hipEventDisableTiming) == hipSuccess
   && hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming) == hipSuccess;
  b->arrived = b->ev_up;
  if (ok) b->d_errors = b->d_ran + HJ_MAX_ROUNDS;
  if (!ok) {
    jga_fail("huff: allocation failed (%d images, %lld scan bytes)", max_images, max_scan_bytes);
    jga_huff_destroy(b);
    return NULL;
  }
  return b;
}

JGA_EXPORT void jga_huff_destroy(jga_huff_batch *b) {
  if (!b) return;
  if (b->h_blob) (void)hipHostFree(b->h_blob);
  if (b->h_ran) (void)hipHostFree(b->h_ran);
  if (b->h_states) (void)hipHostFree(b->h_states);
  if (b->d_blob) (void)hipFree(b->d_blob);
  if (b->d_last_in) (void)hipFree(b->d_last_in);
  if (b->d_R) (void)hipFree(b->d_R);
  if (b->d_B) (void)hipFree(b->d_B);
  if (b->d_list) (void)hipFree(b->d_list);
  if (b->d_lcount) (void)hipFree(b->d_lcount);
  if (b->d_part) (void)hipFree(b->d_part);
  if (b->d_dc) (void)hipFree(b->d_dc);
  if (b->d_blkpos) (void)hipFree(b->d_blkpos);
  if (b->d_dcpart) (void)hipFree(b->d_dcpart);
  if (b->d_ran) (void)hipFree(b->d_ran);                       // (d_errors lies in it)
  if (b->ev_wait) (void)hipEventDestroy(b->ev_wait);
  if (b->ev_up) (void)hipEventDestroy(b->ev_up);
  delete b;
}

// Streams cut into very many restart intervals (one MCU per interval, a 9 x 65528 frame...)
// need more subsequence / segment slots than the byte count suggests: grow the per-lane
// arrays and the upload blob in place.  Called between two barriers of prepare(), by one
// thread, with the scan region of h_blob already written (it is carried over).
static bool grow_batch(jga_huff_batch *b, size_t need_sub, size_t need_blob) {
  if (need_sub > b->sub_cap) {
    const size_t cap = need_sub + need_sub/4 + 1024;
    if (b->d_last_in) (void)hipFree(b->d_last_in);
    if (b->d_R) (void)hipFree(b->d_R);
    if (b->d_B) (void)hipFree(b->d_B);
    if (b->d_list) (void)hipFree(b->d_list);
    if (b->d_part) (void)hipFree(b->d_part);
    b->d_last_in = NULL; b->d_R = NULL; b->d_B = NULL; b->d_list = NULL; b->d_part = NULL; b->sub_cap = 0;
    if (hipMalloc((void **)&b->d_last_in, 8*cap) != hipSuccess
     || hipMalloc((void **)&b->d_R, 4*cap) != hipSuccess
     || hipMalloc((void **)&b->d_B, 4*cap) != hipSuccess
     || hipMalloc((void **)&b->d_list, 8*cap) != hipSuccess
     || hipMalloc((void **)&b->d_part, hj_scan_part_bytes(cap, cap)) != hipSuccess) {
      return false;
    }
    b->sub_cap = cap;
  }
  if (need_blob > b->blob_cap) {
    const size_t cap = need_blob + need_blob/4;
    unsigned char *h = NULL, *d = NULL;
    if (hipHostMalloc((void **)&h, cap, hipHostMallocDefault) != hipSuccess) return false;
    if (hipMalloc((void **)&d, cap) != hipSuccess) { (void)hipHostFree(h); return false; }
    memcpy(h, b->h_blob, b->scan_bytes);
    (void)hipHostFree(b->h_blob);
    (void)hipFree(b->d_blob);
    b->h_blob = h; b->d_blob = d; b->blob_cap = cap;
  }
  return true;
}

// A team of threads that runs a few phases over the images of a batch, with a barrier
// and a short serial section (thread 0) between phases: one spawn per prepare().
namespace {
struct phase_barrier {
  std::atomic<int> count{0}, gen{0};
  int n = 1;
  void wait() {
    const int g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      count.store(0, std::memory_order_relaxed);
      gen.fetch_add(1, std::memory_order_release);
    }
    else {
      while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
  }
};
}  // namespace


// The launch arguments of the synchronisation rounds.
static void fill_sync_args(const jga_huff_batch *b, hj_args &A) {
  memset(&A, 0, sizeof(A));
  A.images = (const hj_image *)(b->d_blob + b->off_images);
  A.segs = (const hj_segment *)(b->d_blob + b->off_segs);
  A.sub_seg = (const uint32_t *)(b->d_blob + b->off_subseg);
  A.tables = (const hj_tables *)(b->d_blob + b->off_tables);
  A.wide = b->wide ? (const hj_wide_ac *)(b->d_blob + b->off_wide) : NULL;
  A.wide_shared = b->wide && b->wide_shared;
  A.scan = b->d_blob + b->off_scan;
  A.S = (uint64_t *)(b->d_blob + b->off_S);
  A.last_in = b->d_last_in;
  A.R = b->d_R;
  A.B = b->d_B;
  A.list = b->d_list;
  A.list_stride = (uint32_t)b->sub_cap;
  A.list_count = b->d_lcount;
  A.scan_part = b->d_part;
  A.ran = b->d_ran;
  A.errors = b->d_errors;
  A.nimages = b->nimages;
  A.sub_log2 = b->sub_log2;
}
// tuning knobs of the rounds, read once (thread-safe: several pipeline lanes decode at the same time)
namespace {
struct round_knobs {
  int it0 = 0, it1 = 0, group = 6, flush_lanes = 16, lean = 1, list_from = -1, it_list = 8, by_block_subs = -1;
  round_knobs() {
    const char *e = jga_tune("JGA_HUFF_ITERS");     // "first,later,group": in-group iterations, rounds per host check
    if (e) sscanf(e, "%d,%d,%d", &it0, &it1, &group);
    if (e && it0 < 1) it0 = 1;
    if (e && it1 < 1) it1 = 1;
    if (group < 1) group = 1;
    e = jga_tune("JGA_HUFF_LIST");                   // "first list round[,steps inside a workgroup]"; 0: no list rounds (A/B knob)
    if (e) {
      sscanf(e, "%d,%d", &list_from, &it_list);
      if (list_from <= 0) list_from = HJ_MAX_ROUNDS;
      if (it_list < 1) it_list = 1;
    }
    e = jga_tune("JGA_HUFF_BY_BLOCK");               // largest batch (subsequences) whose write pass runs one lane per block; 0: none (A/B knob)
    if (e) by_block_subs = atoi(e);
    e = jga_tune("JGA_HUFF_LEAN");                   // 0: the dense kernel's stateless row reader (A/B knob)
    if (e) lean = atoi(e) != 0;
    e = jga_tune("JGA_HUFF_FLUSH");                  // write-pass batching
    if (e) flush_lanes = atoi(e);
    if (flush_lanes < 1) flush_lanes = 1;
  }
};
round_knobs &the_round_knobs_rw() { static round_knobs K; return K; }
const round_knobs &the_round_knobs() { return the_round_knobs_rw(); }
}  // namespace
// Tuning build only (jga_tune() is NULL elsewhere and nothing changes): read the JGA_HUFF_* knobs of the rounds again —
// tools/policy_sweep.py forces one plan after the other onto the same batch in one process.  Not thread-safe: no
// decode may be running.
extern "C" JGA_EXPORT void jga_huff_reload_tuning(void) { the_round_knobs_rw() = round_knobs(); }
// In-group iterations per launch: three — the long, thin tail of the propagation is cheaper as
// further launches than as resident groups — except for a small batch of frames cut into long
// restart intervals with 64-byte subsequences (hj_choose_sub_log2), whose chains are twice as
// many steps of half the length: six (one 1080p frame with an interval per MCU row 0.41 -> 0.39
// ms, the 8K frame of BASELINE config 5 0.64 -> 0.60; intervals of a few subsequences: worse).
// What the plan of a decode's rounds looks at (choose_rounds below; jga_huff_policy exposes the table to tests).
struct batch_facts {
  uint32_t total_sub, total_seg;
  int sub_log2, nslots, restart_interval;
  bool shared, own12;                    // other decodes share the device; the batch brought 12-bit tables per image
};
static batch_facts facts_of(const jga_huff_batch *b) {
  batch_facts F;
  F.total_sub = b->total_sub; F.total_seg = b->total_seg;
  F.sub_log2 = b->sub_log2;
  F.nslots = b->nimages > 0 ? ((const hj_image *)(b->h_blob + b->off_images))[0].nslots : 6;
  F.restart_interval = b->geom.restart_interval;
  F.shared = b->device_shared != 0;
  F.own12 = b->wide && !b->wide_shared;
  return F;
}
static int auto_iters(const batch_facts &F) {
  const bool long_intervals = F.restart_interval > 0 && F.sub_log2 == HJ_SUB_LOG2_MAX - 1
   && F.total_seg > 0 && F.total_sub/F.total_seg >= 64u;
  // (a small batch with its own 12-bit tables, every round by the dense kernel: four — fewer launches for the same
  // chain, a lone 1080p frame 0.355 -> 0.350 ms, 4K 0.386 -> 0.378, 4K 4:4:4 0.280 -> 0.269; five helps the 1080p frame
  // more and costs the 4K one: profiles/r5_lone_frame_chain.md)
  if (F.own12 && !long_intervals) return 4;
  if (long_intervals) return 6;
  // (round 6, tools/policy_sweep.py: frames of three blocks per MCU or fewer — 4:4:4, grey — fall into step inside
  // one subsequence: beyond a small batch the third in-group step only holds the groups' barriers — 32 x 4K 4:4:4
  // 1.86 -> 1.66 ms, 128 x 1080p 4:4:4 1.87 -> 1.75, 32 x 4K grey 1.12 -> 1.03; four blocks per MCU (4:2:2) from 24 MB
  // on: 32 x 4K 1.32 -> 1.21, 128 x 1080p 1.35 -> 1.24; a lone 8K 4:2:2 frame of 16 MB is better off with three)
  const uint64_t bytes = (uint64_t)F.total_sub << F.sub_log2;
  if ((F.nslots <= 3 && bytes > (8ull << 20)) || (F.nslots == 4 && bytes > (24ull << 20))) return 2;
  return 3;
}
// prepare() with the unstuffing left to the device: the host parses the marker segments
// (phase A, as below) and the device does the rest (unstuff_kernels.hip).  The RAW entropy-coded
// bytes of image i either stay where they lie — the copy engine reads them out of the caller's buffer, one copy
// call per file naming it (pinned or registered memory; or ordinary memory, which the runtime then pins per copy) —
// or a host core copies them into the pinned blob and runs of such files go up from there.
// (Round 5 also built a third way — the first clean-up pass READING pinned files over the link itself, no copy call
// at all: 57 GB/s from one launch against 33 per stream for copy calls per 0.8 MB file — and took it out again: a
// kernel with host reads in flight slows every kernel working in HBM beside it two- to fourfold, the copy engines
// do not; tools/r5_probes.hip `contend`, profiles/r5_fetch_by_kernel.md.)
// Per-lane arrays are sized by what the raw lengths allow (a clean stream is never longer than the raw one): image i
// may own up to ceil(avail/sub) + nseg subsequences; how many it really has is known on the device only.
static int prepare_raw(jga_huff_batch *b, const unsigned char *const *jpegs, const int *sizes, int n,
 jga_geom *geom, void *stream) {
  enum { COPIED = 0, NAMED = 1 };
  std::vector<hj_prepared> prep((size_t)n);
  std::atomic<int> next_a(0), next_b(0), failed(0), irregular(0);
  const bool trace = b->trace != 0;
  const auto t_0 = std::chrono::steady_clock::now();
  int nt = b->prepare_threads;
  if (nt <= 0) {
    nt = jga_cpu_budget();
    if (nt > 64) nt = 64;
  }
  if (nt > n) nt = n;
  std::vector<unsigned char> how((size_t)n, (unsigned char)(b->inputs_pinned ? NAMED : COPIED));
  if (!b->input_flags.empty()) {
    for (int i = 0; i < n; i++) how[(size_t)i] = (size_t)i < b->input_flags.size() && b->input_flags[(size_t)i] ? NAMED : COPIED;
  }
  int n_in_place = 0;
  for (int i = 0; i < n; i++) n_in_place += how[(size_t)i] != COPIED;
  // files that stay where they are leave a few microseconds of host work per image (marker parse; the tables of a
  // file that brings the same DHT segments as the one before it are not built again): starting and joining a
  // thread team costs more than it saves until the batch is large
  if (n_in_place == n && n <= 128) nt = 1;
  b->host_bytes = 0;
  b->qtab.assign((size_t)n*192, 0);
  b->verdict.assign((size_t)n, 0);
  b->nimages = 0;
  auto heads = [&]() {
    for (int i = next_a.fetch_add(1); i < n; i = next_a.fetch_add(1)) {
      const int rc = hj_prepare_head(jpegs[i], sizes[i], &prep[i]);
      if (rc != EXIT_SUCCESS) failed.fetch_add(1);
      if (rc == HJ_PREPARE_IRREGULAR) irregular.fetch_add(1);
      b->verdict[i] = (unsigned char)(rc == EXIT_SUCCESS ? 0 : rc == HJ_PREPARE_IRREGULAR ? 2 : 1);
    }
  };
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(heads);
    heads();
    for (auto &th : pool) th.join();
  }
  if (irregular.load()) {
    return jga_fail("huff: %d image(s) of the batch have Huffman tables too irregular (or a frame too "
     "large) for the GPU entropy stage", irregular.load());
  }
  if (failed.load()) return jga_fail("huff: %d image(s) of the batch could not be prepared", failed.load());
  {
    uint64_t raw_total = 0;
    for (int i = 0; i < n; i++) raw_total += prep[i].avail;
    b->sub_log2 = b->force_sub_log2 ? b->force_sub_log2 : hj_choose_sub_log2(raw_total, prep[0].im.nslots, prep[0].geom.restart_interval);
    choose_wide(b, prep, n, raw_total);
  }
  std::vector<hj_unstuff_image> uimg((size_t)n);
  std::vector<uint32_t> sub0v((size_t)n), seg0v((size_t)n);
  size_t o = 0, total_sub = 0, total_seg = 0, total_chunks = 0;
  uint32_t max_nsub = 0, max_chunks = 0;
  for (int i = 0; i < n; i++) {
    if (i && !same_geometry(prep[i].geom, prep[0].geom)) return jga_fail("huff: images of one batch must share a geometry");
    const jga_geom &g = prep[i].geom;
    hj_unstuff_image &u = uimg[(size_t)i];
    u.raw_off = (uint32_t)o;
    u.avail = prep[i].avail;
    u.ri = (uint32_t)g.restart_interval;
    u.total_mcus = (uint32_t)g.nhmb*(uint32_t)g.nvmb;
    u.nseg = u.ri ? (u.total_mcus + u.ri - 1)/u.ri : 1u;
    u.nchunks = (u.avail + HJ_UNSTUFF_CHUNK - 1)/HJ_UNSTUFF_CHUNK;
    if (u.nchunks == 0) u.nchunks = 1;
    u.chunk0 = (uint32_t)total_chunks;
    u.pad_ = 0;
    const uint32_t bound = ((u.avail + (1u << b->sub_log2) - 1) >> b->sub_log2) + u.nseg;
    sub0v[(size_t)i] = (uint32_t)total_sub;
    seg0v[(size_t)i] = (uint32_t)total_seg;
    total_sub += bound;
    total_seg += u.nseg;
    total_chunks += u.nchunks;
    if (bound > max_nsub) max_nsub = bound;
    if (u.nchunks > max_chunks) max_chunks = u.nchunks;
    o += align_up((size_t)u.avail + 16, 16);
  }
  if ((long long)o > b->max_scan + 64ll*n || o >= ((size_t)1 << 32)) {
    return jga_fail("huff: batch exceeds the capacity given to jga_huff_create");
  }
  b->scan_bytes = 0;                                         // (nothing to carry over if the blob grows)
  const size_t raw_bytes = align_up(o, 256);
  size_t q = 0;
  b->off_raw = q; q += raw_bytes;
  // [images | tables | unstuff descriptors | per-image results, preset | quantisers]: one copy call
  b->off_images = q; q += align_up(sizeof(hj_image)*n, 256);
  b->off_tables = q; q += align_up(sizeof(hj_tables)*n, 256);
  b->off_uimg = q; q += align_up(sizeof(hj_unstuff_image)*n, 256);
  b->off_info = q; q += align_up(sizeof(hj_unstuff_info)*n, 256);
  b->off_perr = q; q += align_up(4*(size_t)n, 256);
  b->off_qtab = q; q += align_up(384*(size_t)n, 256);
  b->off_wide = q; q += b->wide ? align_up(sizeof(hj_wide_ac)*(size_t)(b->wide_shared ? 1 : n), 256) : 0;
  b->upload_size = q;                                        // what crosses PCIe
  b->off_scan = q; q += raw_bytes;                           // the clean streams, same offsets as the raw ones
  b->off_segs = q; q += align_up(sizeof(hj_segment)*total_seg, 256);
  b->off_subseg = q; q += align_up(4*total_sub, 256);
  b->off_S = q; q += align_up(8*(total_sub + total_seg), 256);
  b->off_part = q; q += align_up(8*total_chunks, 256);
  b->off_bnd = q; q += align_up(4*total_seg, 256);
  b->blob_size = q;
  const size_t need_sub = total_sub > total_seg ? total_sub : total_seg;
  if ((need_sub > b->sub_cap || q > b->blob_cap) && !grow_batch(b, need_sub, q)) {
    return jga_fail("huff: batch exceeds the capacity given to jga_huff_create");
  }
  hj_image *images = (hj_image *)(b->h_blob + b->off_images);
  hj_tables *tables = (hj_tables *)(b->h_blob + b->off_tables);
  b->nimages = n;
  b->total_sub = (uint32_t)total_sub;
  b->total_seg = (uint32_t)total_seg;
  b->max_nsub = max_nsub;
  b->max_chunks = (int)max_chunks;
  b->geom = prep[0].geom;
  b->max_seg_mcus = 0; b->max_segs_image = 0;
  for (int i = 0; i < n; i++) {
    const hj_unstuff_image &u = uimg[(size_t)i];
    const uint32_t sm = u.ri && u.ri < u.total_mcus ? u.ri : u.total_mcus;
    if (sm > b->max_seg_mcus) b->max_seg_mcus = sm;
    if (u.nseg > b->max_segs_image) b->max_segs_image = u.nseg;
  }
  b->unstuffed_on_device = 1;
  b->shadow.clear();
  hipStream_t st = (hipStream_t)stream;
  hipStream_t up = b->copy_stream ? b->copy_stream : st;       // (see jga_huff_set_copy_stream)
  // A batch that has to wait for the link anyway spends the wait copying the files it would have had read where
  // they lie into its pinned blob: they then cross the link as ONE copy call at its full rate (55 GB/s) instead of
  // a call per file (0.8 MB files: ~40 GB/s from two batches side by side).  File by file, for as long as the
  // caller's poll says "not yet" (jga_huff_set_upload_poll).
  enum { COPIED_ALREADY = 2 };
  memcpy(b->h_blob + b->off_uimg, uimg.data(), sizeof(hj_unstuff_image)*(size_t)n);
  memset(b->h_blob + b->off_info, 0xFF, sizeof(hj_unstuff_info)*(size_t)n);   // (hard_end = "none yet")
  memset(b->h_blob + b->off_perr, 0, 4*(size_t)n);
  auto copies = [&]() {
    for (int i = next_b.fetch_add(1); i < n; i = next_b.fetch_add(1)) {
      hj_prepared &p = prep[i];
      if (how[(size_t)i] == COPIED) memcpy(b->h_blob + b->off_raw + uimg[(size_t)i].raw_off, jpegs[i] + p.desc->scan_off, p.avail);
      p.im.sub0 = sub0v[(size_t)i];
      p.im.seg0 = seg0v[(size_t)i];
      p.im.scan_off = uimg[(size_t)i].raw_off;
      p.im.nseg = uimg[(size_t)i].nseg;
      p.im.nsub = 0;                                         // (the device fills these two in)
      p.im.scan_len = 0;
      images[i] = p.im;
      tables[i] = p.tabs;
      if (b->wide && (!b->wide_shared || i == 0)) memcpy(b->h_blob + b->off_wide + sizeof(hj_wide_ac)*(size_t)i, p.wide.data(), sizeof(hj_wide_ac));
      memcpy(&b->qtab[(size_t)i*192], p.qtab, sizeof(p.qtab));
    }
  };
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(copies);
    copies();
    for (auto &th : pool) th.join();
  }
  memcpy(b->h_blob + b->off_qtab, b->qtab.data(), 384*(size_t)n);
  b->qtab_on_device = true;
  // (the poll comes LAST of the host work: a turn on the link it takes is used at once — ADVICE r5)
  if (b->upload_poll) {
    for (int i = 0; i < n && !b->upload_poll(b->before_upload_arg); i++) {
      if (how[(size_t)i] != NAMED) continue;
      memcpy(b->h_blob + b->off_raw + uimg[(size_t)i].raw_off, jpegs[i] + prep[i].desc->scan_off, prep[i].avail);
      how[(size_t)i] = COPIED_ALREADY;
    }
  }
  int ncopies = 1;
  b->copied.assign((size_t)n, 0);
  for (int i = 0; i < n; i++) {
    b->copied[(size_t)i] = how[(size_t)i] != NAMED;
    if (how[(size_t)i] != NAMED) b->host_bytes += (long long)prep[i].avail;
    if (how[(size_t)i] == NAMED || i == 0 || how[(size_t)i - 1] == NAMED) ncopies++;
  }
  if (b->before_upload) b->before_upload(b->before_upload_arg, (long long)b->upload_size, ncopies);
  const auto t_1 = std::chrono::steady_clock::now();
  // copy calls: files the copy engine reads where they lie, runs of files a host core put into the blob, and the
  // descriptors
  bool descriptors_sent = false;
  for (int i = 0; i < n; ) {
    if (how[(size_t)i] == NAMED) {
      HOK(hipMemcpyAsync(b->d_blob + b->off_raw + uimg[(size_t)i].raw_off, jpegs[i] + prep[i].desc->scan_off,
       prep[i].avail, hipMemcpyHostToDevice, up));
      i++;
      continue;
    }
    int j = i;
    while (j + 1 < n && how[(size_t)j + 1] != NAMED) j++;
    const size_t from = uimg[(size_t)i].raw_off, to = (size_t)uimg[(size_t)j].raw_off + uimg[(size_t)j].avail;
    if (j + 1 == n) {
      // the run reaches the last file: the descriptors lie right behind the raw bytes in both blobs — ONE copy call
      // for both (round 6: a second call per group held the link for ~40 us: the small copy itself and two hand-overs
      // in the copy queue, eight times per 128-file shard)
      HOK(hipMemcpyAsync(b->d_blob + b->off_raw + from, b->h_blob + b->off_raw + from, b->upload_size - (b->off_raw + from), hipMemcpyHostToDevice, up));
      descriptors_sent = true;
    }
    else HOK(hipMemcpyAsync(b->d_blob + b->off_raw + from, b->h_blob + b->off_raw + from, to - from, hipMemcpyHostToDevice, up));
    i = j + 1;
  }
  if (!descriptors_sent) {
    HOK(hipMemcpyAsync(b->d_blob + b->off_images, b->h_blob + b->off_images, b->upload_size - b->off_images,
     hipMemcpyHostToDevice, up));
  }
  HOK(hipEventRecord(b->ev_up, up));
  hj_unstuff_args U;
  memset(&U, 0, sizeof(U));
  U.raw = b->d_blob + b->off_raw;
  U.clean = b->d_blob + b->off_scan;
  U.images = (hj_image *)(b->d_blob + b->off_images);
  U.segs = (hj_segment *)(b->d_blob + b->off_segs);
  U.uimg = (const hj_unstuff_image *)(b->d_blob + b->off_uimg);
  U.part = (uint32_t *)(b->d_blob + b->off_part);
  U.bnd = (uint32_t *)(b->d_blob + b->off_bnd);
  U.info = (hj_unstuff_info *)(b->d_blob + b->off_info);
  U.errors = (uint32_t *)(b->d_blob + b->off_perr);
  U.nimages = n;
  U.sub_log2 = b->sub_log2;
  const auto t_2 = std::chrono::steady_clock::now();
  if (up != st) HOK(hipStreamWaitEvent(st, b->ev_up, 0));
  b->arrived = b->ev_up;
  if (hj_launch_unstuff(&U, b->max_chunks, st)) return jga_fail("huff: launch failed");
  if (trace) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) {
      return std::chrono::duration<double, std::milli>(c - a).count(); };
    fprintf(stderr, "  prepare (device clean-up, %d of %d files in place, %d threads): heads + blob %.2f ms, %d copy call(s) %.2f ms, 4 launches %.2f ms\n",
     n_in_place, n, nt, ms(t_0, t_1), ncopies, ms(t_1, t_2),
     ms(t_2, std::chrono::steady_clock::now()));
  }
  if (geom) *geom = b->geom;
  return EXIT_SUCCESS;
}

// Parse + stage a batch.  All images must share one geometry (returned in *geom).
// Host work per image (marker parse, table build, unstuffing straight into the pinned
// upload buffer, lane start states) is independent and fanned out over a thread team.
JGA_EXPORT int jga_huff_prepare(jga_huff_batch *b, const unsigned char *const *jpegs,
 const int *sizes, int n, jga_geom *geom, void *stream) {
  if (n < 1 || n > b->max_images) return jga_fail("huff: batch size %d out of range", n);
  if (b->device_unstuff) return prepare_raw(b, jpegs, sizes, n, geom, stream);
  b->unstuffed_on_device = 0;
  b->host_bytes = 0;
  b->qtab_on_device = false;
  b->wide = false;
  const auto t_p0 = std::chrono::steady_clock::now();
  std::vector<hj_prepared> prep((size_t)n);
  std::vector<uint32_t> scan_off((size_t)n), sub0v((size_t)n), seg0v((size_t)n);
  std::atomic<int> next_a(0), next_b(0), next_c(0), failed(0), irregular(0);
  std::atomic<int> fatal(0);       // set by the serial sections: 1 geometry, 2 capacity
  std::atomic<int> stop(0);        // decided in a serial section, read after the next barrier
  phase_barrier bar;
  int nt = b->prepare_threads;
  if (nt <= 0) {
    nt = jga_cpu_budget();
    if (nt > 64) nt = 64;
  }
  if (nt > n) nt = n;
  bar.n = nt;
  b->qtab.assign((size_t)n*192, 0);
  b->verdict.assign((size_t)n, 0);
  auto work = [&](int tid) {
    // phase A: headers, tables
    for (;;) {
      const int i = next_a.fetch_add(1);
      if (i >= n) break;
      const int rc = hj_prepare_head(jpegs[i], sizes[i], &prep[i]);
      if (rc != EXIT_SUCCESS) failed.fetch_add(1);
      if (rc == HJ_PREPARE_IRREGULAR) irregular.fetch_add(1);
      b->verdict[i] = (unsigned char)(rc == EXIT_SUCCESS ? 0 : rc == HJ_PREPARE_IRREGULAR ? 2 : 1);
    }
    bar.wait();
    if (tid == 0 && failed.load()) stop.store(1);
    if (tid == 0 && !failed.load()) {
      size_t o = 0;
      for (int i = 0; i < n; i++) {
        if (i && !same_geometry(prep[i].geom, prep[0].geom)) fatal.store(1);
        scan_off[i] = (uint32_t)o;
        o += align_up((size_t)prep[i].avail + 16, 16);
      }
      if ((long long)o > b->max_scan + 64ll*n || o >= ((size_t)1 << 32)) fatal.store(2);   // (hj_image::scan_off is 32 bits)
      // subsequence length of this batch (the stuffed length is close enough to the clean one)
      b->sub_log2 = b->force_sub_log2 ? b->force_sub_log2 : hj_choose_sub_log2(o, prep[0].im.nslots, prep[0].geom.restart_interval);
      for (int i = 0; i < n; i++) prep[i].sub_log2 = b->sub_log2;
      choose_wide(b, prep, n, o);
      b->off_scan = 0;
      b->scan_bytes = align_up(o, 256);
      if (fatal.load()) stop.store(1);
    }
    bar.wait();
    if (stop.load()) return;
    // phase B: clean scan bytes straight into the pinned blob, restart segments
    for (;;) {
      const int i = next_b.fetch_add(1);
      if (i >= n) break;
      if (hj_prepare_scan(jpegs[i], sizes[i], &prep[i], b->h_blob + scan_off[i]) != EXIT_SUCCESS) {
        failed.fetch_add(1);
        b->verdict[i] = 1;
      }
    }
    bar.wait();
    if (tid == 0 && failed.load()) stop.store(1);
    if (tid == 0 && !failed.load()) {
      size_t total_sub = 0, total_seg = 0;
      uint32_t max_nsub = 0;
      for (int i = 0; i < n; i++) {
        sub0v[i] = (uint32_t)total_sub;
        seg0v[i] = (uint32_t)total_seg;
        total_sub += prep[i].im.nsub;
        total_seg += prep[i].segs.size();
        if (prep[i].im.nsub > max_nsub) max_nsub = prep[i].im.nsub;
      }
      b->nimages = n;
      b->total_sub = (uint32_t)total_sub;
      b->total_seg = (uint32_t)total_seg;
      b->max_nsub = max_nsub;
      b->geom = prep[0].geom;
      b->max_seg_mcus = 0; b->max_segs_image = 0;
      for (int i = 0; i < n; i++) {
        for (const hj_segment &sg : prep[i].segs) if (sg.nmcu > b->max_seg_mcus) b->max_seg_mcus = sg.nmcu;
        if (prep[i].segs.size() > b->max_segs_image) b->max_segs_image = (uint32_t)prep[i].segs.size();
      }
      size_t o = b->scan_bytes;
      b->off_images = o; o += align_up(sizeof(hj_image)*n, 256);
      b->off_segs = o; o += align_up(sizeof(hj_segment)*total_seg, 256);
      b->off_tables = o; o += align_up(sizeof(hj_tables)*n, 256);
      b->off_qtab = o; o += align_up(384*(size_t)n, 256);
      b->off_wide = o; o += b->wide ? align_up(sizeof(hj_wide_ac)*(size_t)(b->wide_shared ? 1 : n), 256) : 0;
      b->upload_size = o;           // what crosses PCIe; the rest is written by hj_init_states
      b->off_subseg = o; o += align_up(4*total_sub, 256);
      b->off_S = o; o += align_up(8*(total_sub + total_seg), 256);
      b->blob_size = o;
      const size_t need_sub = total_sub > total_seg ? total_sub : total_seg;
      if ((need_sub > b->sub_cap || o > b->blob_cap) && !grow_batch(b, need_sub, o)) fatal.store(2);
      if (fatal.load()) stop.store(1);
    }
    bar.wait();
    if (stop.load()) return;
    // phase C: descriptors, tables, lane start states of each image
    hj_image *images = (hj_image *)(b->h_blob + b->off_images);
    hj_segment *segs = (hj_segment *)(b->h_blob + b->off_segs);
    hj_tables *tables = (hj_tables *)(b->h_blob + b->off_tables);
    for (;;) {
      const int i = next_c.fetch_add(1);
      if (i >= n) break;
      hj_prepared &p = prep[i];
      const uint32_t sub0 = sub0v[i], seg0 = seg0v[i];
      p.im.sub0 = sub0;
      p.im.seg0 = seg0;
      p.im.scan_off = scan_off[i];
      images[i] = p.im;
      tables[i] = p.tabs;
      if (b->wide && (!b->wide_shared || i == 0)) memcpy(b->h_blob + b->off_wide + sizeof(hj_wide_ac)*(size_t)i, p.wide.data(), sizeof(hj_wide_ac));
      memcpy(&b->qtab[(size_t)i*192], p.qtab, sizeof(p.qtab));
      for (size_t si = 0; si < p.segs.size(); si++) segs[seg0 + si] = p.segs[si];
    }
  };
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
  }
  b->nimages = 0;
  if (irregular.load()) {
    return jga_fail("huff: %d image(s) of the batch have Huffman tables too irregular (or a frame too "
     "large) for the GPU entropy stage", irregular.load());
  }
  if (failed.load()) return jga_fail("huff: %d image(s) of the batch could not be prepared", failed.load());
  if (fatal.load() == 1) return jga_fail("huff: images of one batch must share a geometry");
  if (fatal.load()) return jga_fail("huff: batch exceeds the capacity given to jga_huff_create");
  b->nimages = n;
  for (int i = 0; i < n; i++) b->host_bytes += (long long)prep[i].avail;
  memcpy(b->h_blob + b->off_qtab, b->qtab.data(), 384*(size_t)n);
  b->qtab_on_device = true;
  b->arrived = b->ev_up;
  // the scan region is sized from the raw lengths; the bytes between an image's clean
  // stream (+16 pad) and the next image's start are never read
  const bool trace = b->trace != 0;
  if (b->before_upload) b->before_upload(b->before_upload_arg, (long long)b->upload_size, 1);
  const auto t_h = std::chrono::steady_clock::now();
  if (b->copy_stream && b->copy_stream != (hipStream_t)stream) {
    HOK(hipMemcpyAsync(b->d_blob, b->h_blob, b->upload_size, hipMemcpyHostToDevice, b->copy_stream));
    HOK(hipEventRecord(b->ev_up, b->copy_stream));
    HOK(hipStreamWaitEvent((hipStream_t)stream, b->ev_up, 0));
  }
  else {
    HOK(hipMemcpyAsync(b->d_blob, b->h_blob, b->upload_size, hipMemcpyHostToDevice, (hipStream_t)stream));
    HOK(hipEventRecord(b->ev_up, (hipStream_t)stream));       // (jga_huff_wait_upload)
  }
  if (trace) {
    fprintf(stderr, "  prepare: host %.2f ms, hipMemcpyAsync call %.2f ms (%zu MB)\n",
     std::chrono::duration<double, std::milli>(t_h - t_p0).count(),
     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h).count(), b->upload_size >> 20);
  }
  if (geom) *geom = b->geom;
  return EXIT_SUCCESS;
}

// Per-image outcome of the last prepare(): 0 usable, 1 not (damaged / unsupported file), 2 a
// valid file the device format cannot hold — its caller should take the host entropy stage.
JGA_EXPORT int jga_huff_prepare_verdict(const jga_huff_batch *b, int i) {
  return (i >= 0 && (size_t)i < b->verdict.size()) ? (int)b->verdict[(size_t)i] : -1;
}

// 1: prepare() uploads the raw scans and the device removes stuffing / RSTn markers and builds the
// restart segments (unstuff_kernels.hip); 0: the host does (hj_prepare_scan).  Same results.
JGA_EXPORT void jga_huff_set_device_unstuff(jga_huff_batch *b, int on) { b->device_unstuff = on != 0; }
// The buffers handed to prepare() are pinned (hipHostMalloc / hipHostRegister, e.g. jga_host_register):
// with on-device unstuffing the scans are then DMA'd straight out of them.  The caller keeps them
// unchanged until the stream has passed the copies (any later jga_huff_decode has).
JGA_EXPORT void jga_huff_set_inputs_pinned(jga_huff_batch *b, int on) { b->inputs_pinned = on != 0; }
// 1: jga_huff_decode's host waits sleep (blocking event) instead of spinning on a core.
JGA_EXPORT void jga_huff_set_blocking_waits(jga_huff_batch *b, int on) { b->blocking_waits = on != 0; }
JGA_EXPORT void jga_huff_set_device_shared(jga_huff_batch *b, int on) { b->device_shared = on < 0 ? 0 : on > 2 ? 2 : on; }
// Wait until the last prepare()'s upload has arrived (its kernels, if it queued any, may still run).
JGA_EXPORT int jga_huff_wait_upload(jga_huff_batch *b) {
  HOK(b->blocking_waits ? jga_event_wait_sleeping(b->arrived) : hipEventSynchronize(b->arrived));
  return EXIT_SUCCESS;
}
JGA_EXPORT void jga_huff_set_upload_gate(jga_huff_batch *b, void (*fn)(void *, long long, int), void *arg) {
  b->before_upload = fn;
  b->before_upload_arg = arg;
  if (!fn) b->upload_poll = nullptr;
}
// With a gate set: `poll(arg)` (the gate's arg) answers "would the gate let me through now?" (non-zero: yes, and
// the turn is then mine: the gate call that follows returns at once).  prepare() with the clean-up on the device
// asks between copying, one by one, the files it was told to read where they lie into its pinned blob instead —
// a batch that waits for the link anyway then uploads as one copy call at the link's full rate.
JGA_EXPORT void jga_huff_set_upload_poll(jga_huff_batch *b, int (*poll)(void *)) { b->upload_poll = poll; }
// prepare() queues its uploads on `copy_stream` (a hipStream_t; NULL = on prepare()'s own stream)
// and makes its own stream wait for them.  Several batches that share one copy stream upload
// one after the other, in the order they were prepared — the first one's decode starts when ITS
// bytes have arrived, not when everybody's have (copies on separate streams share the link evenly
// and all finish together).
JGA_EXPORT void jga_huff_set_copy_stream(jga_huff_batch *b, void *copy_stream) { b->copy_stream = (hipStream_t)copy_stream; }

// Host threads prepare() may use (0 = up to 64, one per image).
JGA_EXPORT void jga_huff_set_threads(jga_huff_batch *b, int nthreads) { b->prepare_threads = nthreads; }
// Per-image "this buffer is pinned / registered" for the next prepare() (NULL: the all-or-nothing flag).
JGA_EXPORT void jga_huff_set_input_flags(jga_huff_batch *b, const unsigned char *flags, int n) {
  if (flags && n > 0) b->input_flags.assign(flags, flags + n);
  else b->input_flags.clear();
}
JGA_EXPORT long long jga_huff_host_bytes(const jga_huff_batch *b) { return b->host_bytes; }
// 1: a host core copied image i's scan (into the pinned blob, or while cleaning it up) in the last prepare(); 0: the
// copy engine read it where it lies; -1: no such image.
JGA_EXPORT int jga_huff_image_copied(const jga_huff_batch *b, int i) {
  if (i < 0 || i >= b->nimages) return -1;
  return b->unstuffed_on_device && (size_t)i < b->copied.size() ? (int)b->copied[(size_t)i] : 1;
}
JGA_EXPORT int jga_huff_set_option(jga_huff_batch *b, int option, int value) {
  switch (option) {
    case JGA_HUFF_OPT_SUB_BYTES :
      b->force_sub_log2 = value == 32 ? 5 : value == 64 ? 6 : value == 128 ? 7 : 0;
      return value == 0 || b->force_sub_log2 ? EXIT_SUCCESS : jga_fail("huff: subsequence length %d (32, 64 or 128)", value);
    case JGA_HUFF_OPT_ASSIST_AFTER : b->assist_after = value > 0 ? value : 0; return EXIT_SUCCESS;
    case JGA_HUFF_OPT_SPECULATE : b->speculate = value < 0 ? -1 : 0; return EXIT_SUCCESS;
    case JGA_HUFF_OPT_TRACE : b->trace = value != 0; return EXIT_SUCCESS;
    default : return jga_fail("huff: unknown option %d", option);
  }
}

// Bytes uploaded by the last prepare() (tables + states + compressed scan data).
JGA_EXPORT long long jga_huff_upload_bytes(const jga_huff_batch *b) { return (long long)b->upload_size; }
JGA_EXPORT int jga_huff_last_rounds(const jga_huff_batch *b) { return b->last_rounds; }
JGA_EXPORT int jga_huff_last_assisted(const jga_huff_batch *b) { return b->last_assisted; }
// Quantisation tables of the prepared batch: nimages*3*64 uint16 (host memory).
JGA_EXPORT const unsigned short *jga_huff_qtabs(const jga_huff_batch *b) { return b->qtab.data(); }
// The same in device memory: they went up with the last prepare()'s descriptors (valid once the stream prepare() was
// given has passed its upload, i.e. for anything queued on that stream afterwards; until the next prepare()).
JGA_EXPORT const unsigned short *jga_huff_qtabs_device(const jga_huff_batch *b) {
  return b->qtab_on_device && b->nimages ? (const unsigned short *)(b->d_blob + b->off_qtab) : NULL;
}

// Decode the prepared batch into d_coef (image i at d_coef + i*coef_stride shorts).
// May be called repeatedly on the same prepared batch (state is reset each time).
// Streams that do not self-synchronise.  Periodic data — flat areas, letterbox bars: blocks
// of "DC difference 0, EOB" — can be parsed out of step for ever, so the true states only
// travel down such a stretch one subsequence per run, at the speed of ONE lane (128 bytes in
// ~50 us: a 36 KB bar would take 14 ms, a flat 4K frame hundreds of rounds).  When the rounds
// have not settled after JGA_HUFF_ASSIST_AFTER of them (default 12; a photograph needs 4-6),
// the host walks those stretches itself: it reads the states back, and in every segment starts
// at each lane whose start state moved since its last run (the first such lane's state is
// true by induction from the segment start), decodes on with the same hj_sync_decode the
// kernels run (~100x a lane's speed) and writes the state at every subsequence boundary, until
// it arrives in a state from which the next lane has already run.  The corrected states go
// back up; one more round re-runs all those lanes at once, from true states.
static int assist_chains(jga_huff_batch *b, hipStream_t st) {
  const size_t ns = (size_t)b->total_sub + b->total_seg, nl = b->total_sub;
  if (b->h_states_cap < ns + nl) {
    if (b->h_states) (void)hipHostFree(b->h_states);
    b->h_states = NULL; b->h_states_cap = 0;
    HOK(hipHostMalloc((void **)&b->h_states, 8*(ns + nl), hipHostMallocDefault));
    b->h_states_cap = ns + nl;
  }
  uint64_t *S = b->h_states, *last_in = b->h_states + ns;
  HOK(hipMemcpyAsync(S, b->d_blob + b->off_S, 8*ns, hipMemcpyDeviceToHost, st));
  HOK(hipMemcpyAsync(last_in, b->d_last_in, 8*nl, hipMemcpyDeviceToHost, st));
  HOK(hipStreamSynchronize(st));
  const unsigned char *host = b->h_blob;
  if (b->unstuffed_on_device) {
    // the descriptors, segment tables and clean streams exist in HBM only: fetch them once
    // (a rare path: streams that never fall into step)
    if (b->shadow.empty()) {
      b->shadow.resize(b->off_segs + align_up(sizeof(hj_segment)*b->total_seg, 256));
      HOK(hipMemcpy(b->shadow.data() + b->off_images, b->d_blob + b->off_images, sizeof(hj_image)*(size_t)b->nimages, hipMemcpyDeviceToHost));
      HOK(hipMemcpy(b->shadow.data() + b->off_scan, b->d_blob + b->off_scan,
       b->off_segs - b->off_scan + sizeof(hj_segment)*(size_t)b->total_seg, hipMemcpyDeviceToHost));
    }
    host = b->shadow.data();
  }
  const hj_image *images = (const hj_image *)(host + b->off_images);
  const hj_segment *segs = (const hj_segment *)(host + b->off_segs);
  const hj_tables *tables = (const hj_tables *)(b->h_blob + b->off_tables);
  std::atomic<int> next(0), walked(0);
  auto work = [&]() {
    for (int i = next.fetch_add(1); i < b->nimages; i = next.fetch_add(1)) {
      const hj_image &im = images[i];
      walked.fetch_add(hj_walk_unsettled(im, segs + im.seg0, &tables[i], host + b->off_scan + im.scan_off,
       S + im.sub0 + im.seg0, last_in + im.sub0, b->sub_log2), std::memory_order_relaxed);
    }
  };
  {
    int nt = b->prepare_threads > 0 ? b->prepare_threads : 8;
    if (nt > b->nimages) nt = b->nimages;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
  }
  b->last_assisted += walked.load();
  b->list_state = 2;                                         // the lists are made afresh from the corrected states
  HOK(hipMemcpyAsync(b->d_blob + b->off_S, S, 8*ns, hipMemcpyHostToDevice, st));
  return EXIT_SUCCESS;
}

// Wait for everything queued on `st`.  hipStreamSynchronize spins on a host core — the right
// thing for one frame's latency, the wrong one for a pipeline whose lanes outnumber the CPUs
// the container grants: there the waiting lane polls and sleeps (host_wait.h; an event with
// hipEventBlockingSync spins just the same on this stack) and leaves the core to the lanes that
// are parsing headers.
static hipError_t wait_stream(jga_huff_batch *b, hipStream_t st) {
  if (!b->blocking_waits) return hipStreamSynchronize(st);
  return jga_stream_wait_sleeping(st, b->ev_wait);
}

static double thread_cpu_ms() {
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return (double)ts.tv_sec*1e3 + (double)ts.tv_nsec*1e-6;
}

// ---- the plan of a decode's rounds: ONE table (round 6) ---------------------------------------------------------
// What a batch is, as far as the choice goes:
//   crowd   alone and small  (nobody else on the device, at most 200 k subsequences: 8 x 4K, 32 x 1080p)
//           fills or shares  (a pipeline's lanes side by side, or a batch that fills the device by itself)
//   size    subsequences in the batch (b->total_sub)
//   own12   the batch brought 12-bit AC tables per image (at most four images, alone: hj_wide_ac)
//   dri     restart intervals (chains end at the next marker)
// and what is chosen for it:
//   lists     the rounds after the first launch run from work lists (hj_list_build + hj_sync_list) instead of the dense
//             kernel again
//   iters     in-group steps of a dense launch
//   by_block  the write pass one lane per BLOCK (hj_block_starts + hj_write_blocks) instead of hj_write
//
//   crowd            size        own12  dri  | lists  iters                       by_block | measured (ms, chosen / other)
//   alone and small  <= 64 k     any    any  | no     3 (4 own12, 6 long dri)     yes      | 1 x 1080p 0.40 / 0.45-0.49 lists; 0.41 -> 0.32 by block (r5_list_rounds, r5_lone_frame_chain)
//   alone and small  64-128 k    any    any  | no     3 (4 own12, 6 long dri)     no       | 8 x 2.7K 0.547 / 0.677 lists, 8 x 4K photo 0.582 / 0.729 (r6_policy_sweep; r5's 16 x 1080p: 0.543 / 0.522)
//   alone and small  128-200 k   yes    any  | no     4                           no       | 4 x 4K own tables 0.522 / 0.536 lists
//   alone and small  128-200 k   no     yes  | no     3 (6 long dri)              no       | 8K DRI frame 0.66 / 0.69 lists
//   alone and small  128-200 k   no     no   | yes    3                           no       | 8 x 4K 0.592 / 0.612 dense; 32 x 1080p 0.584 / 0.612
//   fills or shares  any         -      any  | yes    3 (6 long dri; 2: see below) no       | 48 x 4K 1.82-1.86 / 1.91-1.96; 64 x 1080p 0.87 / 0.95
//   iters = 2 for batches over 8 MB of frames with <= 3 blocks per MCU, over 24 MB with 4 (auto_iters)
// (the tuning build's JGA_HUFF_LIST / JGA_HUFF_ITERS / JGA_HUFF_BY_BLOCK force a column; tools/policy_sweep.py runs every
// forced alternative over geometries x samplings x batch sizes x qualities x contents: profiles/r6_policy_sweep.md)
struct round_choice { bool lists, by_block; int iters; };
static round_choice policy_rounds(const batch_facts &F) {
  const bool alone_small = !F.shared && F.total_sub <= 200u*1024u;
  const bool dri = F.restart_interval > 0;
  round_choice R;
  R.iters = auto_iters(F);
  R.lists = !(alone_small && (F.total_sub <= 128u*1024u || F.own12 || dri));
  R.by_block = !F.shared && F.total_sub <= 64u*1024u;
  return R;
}
static round_choice choose_rounds(const jga_huff_batch *b) {
  const round_knobs &K = the_round_knobs();
  round_choice R = policy_rounds(facts_of(b));
  if (K.list_from >= 0) R.lists = K.list_from < HJ_MAX_ROUNDS;                                  // (tuning build: a forced column)
  if (K.by_block_subs >= 0) R.by_block = !b->device_shared && b->total_sub <= (uint32_t)K.by_block_subs;
  return R;
}
// The table above as a function of the facts alone (host logic, no device): what a batch of `total_sub` subsequences of
// 1 << sub_log2 bytes in `total_seg` restart segments, `nslots` blocks per MCU, would get.  For tests/test_huff_emul.py.
extern "C" JGA_EXPORT void jga_huff_policy(unsigned total_sub, unsigned total_seg, int sub_log2, int nslots, int restart_interval,
 int shared, int own12, int *lists, int *iters, int *by_block) {
  batch_facts F;
  F.total_sub = total_sub; F.total_seg = total_seg; F.sub_log2 = sub_log2; F.nslots = nslots;
  F.restart_interval = restart_interval; F.shared = shared != 0; F.own12 = own12 != 0;
  const round_choice R = policy_rounds(F);
  if (lists) *lists = R.lists;
  if (iters) *iters = R.iters;
  if (by_block) *by_block = R.by_block;
}

// ---- a decode, in two halves -------------------------------------------------------------------------------
// decode_begin() queues the start states, the first burst of synchronisation rounds and — speculatively, see
// below — the whole tail (prefix sums, write pass, DC values, verdict copy) and RETURNS: nothing has been waited
// for.  decode_end() waits for the stream, looks at what the rounds reported and, if the speculation did not
// hold (or was not made), goes on: more rounds, the host's walk over stretches that never fall into step, the
// tail again.  A caller that runs a block-decode kernel next queues it between the two halves — one host wait
// per decode instead of two, and the kernel is in the queue before anybody else's work can get between the
// write pass and it — and asks decode_end() whether what it queued saw the final planes (`*valid_behind`).
struct decode_plan {
  hj_args A;
  int it0, it1, group, assist_after, lean, list_from, it_list;
  bool speculate, by_block;
};
static int make_plan(jga_huff_batch *b, short *d_coef, long long coef_stride, short *d_dc, long long dc_stride, decode_plan &P) {
  hj_args &A = P.A;
  fill_sync_args(b, A);
  {
    // DC differences (scan order) and, unless the caller brings its own array, DC values (by
    // buffer slot); one stride for both: the caller's, or the slot count rounded up
    const size_t stride = d_dc ? (size_t)dc_stride : (size_t)((b->geom.coef_shorts/64 + 127) & ~127ll);
    // (sized for the batch object's capacity, not for this batch: no re-allocation — it
    // synchronises the device — when a bigger batch of the same frames follows)
    const size_t need = stride*(size_t)(b->max_images > b->nimages ? b->max_images : b->nimages);
    if (need > b->dc_cap) {
      if (b->d_dc) (void)hipFree(b->d_dc);
      b->d_dc = NULL; b->dc_cap = 0;
      HOK(hipMalloc((void **)&b->d_dc, 2*sizeof(int16_t)*need));
      b->dc_cap = need;
    }
    A.dc_chunks_per_image = hj_dc_chunks_per_image(b->geom.nhmb*b->geom.nvmb, (int)b->max_segs_image);
    const size_t pneed = 3*(size_t)(b->max_images > b->nimages ? b->max_images : b->nimages)*(size_t)A.dc_chunks_per_image;
    if (pneed > b->dcpart_cap) {
      if (b->d_dcpart) (void)hipFree(b->d_dcpart);
      b->d_dcpart = NULL; b->dcpart_cap = 0;
      HOK(hipMalloc((void **)&b->d_dcpart, 4*pneed));
      b->dcpart_cap = pneed;
    }
    A.dc_diff = b->d_dc;
    A.dc_val = d_dc ? (int16_t *)d_dc : b->d_dc + b->dc_cap;
    A.dc_stride = (long long)stride;
    P.by_block = choose_rounds(b).by_block;
    if (P.by_block) {
      const size_t bneed = stride*(size_t)b->nimages;
      if (bneed > b->blkpos_cap) {
        if (b->d_blkpos) (void)hipFree(b->d_blkpos);
        b->d_blkpos = NULL; b->blkpos_cap = 0;
        HOK(hipMalloc((void **)&b->d_blkpos, sizeof(uint32_t)*bneed));
        b->blkpos_cap = bneed;
      }
      A.blk_pos = b->d_blkpos;
    }
  }
  A.coef = (int16_t *)d_coef;
  A.coef_stride = coef_stride;
  const round_knobs &K = the_round_knobs();
  const round_choice R = choose_rounds(b);
  P.it0 = K.it0 > 0 ? K.it0 : R.iters;
  P.it1 = K.it1 > 0 ? K.it1 : R.iters;
  P.group = K.group;
  P.lean = K.lean;
  P.assist_after = b->assist_after > 0 ? b->assist_after : 12;
  P.list_from = R.lists ? 1 : HJ_MAX_ROUNDS;
  P.it_list = K.it_list;
  A.flush_lanes = K.flush_lanes;
  A.sub_log2 = b->sub_log2;
  P.speculate = b->speculate >= 0 && !b->assist_hint;
  return EXIT_SUCCESS;
}
// What the write pass needs cleared (DC arrays, padding slots): workgroups of the decode's first launch (hj_init_states).
static void clear_regions(jga_huff_batch *b, const hj_args &A, hj_clear_args &C) {
  memset(&C, 0, sizeof(C));
  auto add = [&](void *base, uint64_t row_bytes, uint64_t stride, uint32_t rows) {
    hj_clear_region &r = C.region[C.nregions++];
    r.base = base; r.row_bytes = row_bytes; r.stride = stride; r.rows = rows;
  };
  // (the planes themselves need no clear: the write pass stores whole 128-byte lines, and the lines
  // of blocks that two lanes share are zeroed by hj_scan<true> — only the slots at the end of a
  // decimated plane that hold no block are cleared here, so that the buffer reads like the host
  // stage's)
  for (int p = 0; p < b->geom.nplanes; p++) {
    const jga_plane_geom &pg = b->geom.plane[p];
    const long long rs = (long long)b->geom.w0*8, used = (long long)pg.hblocks*pg.vblocks*64;
    const long long end = p + 1 < b->geom.nplanes ? b->geom.plane[p + 1].coef_off : b->geom.coef_shorts;
    if ((long long)pg.hblocks*64 != (rs >> pg.xdec)) {          // rows with gaps (rare samplings): clear it all
      C.nregions = 0;
      add(A.coef, (uint64_t)A.coef_stride*2*(uint64_t)b->nimages, 0, 1);
      break;
    }
    if (end > pg.coef_off + used) add(A.coef + pg.coef_off + used, (uint64_t)(end - pg.coef_off - used)*2, (uint64_t)A.coef_stride*2, (uint32_t)b->nimages);
  }
  // (so are the DC arrays: blocks a damaged stream never reaches, slots that hold no block)
  add(A.dc_diff, sizeof(int16_t)*(uint64_t)A.dc_stride*(uint64_t)b->nimages, 0, 1);
  add(A.dc_val, sizeof(int16_t)*(uint64_t)A.dc_stride*(uint64_t)b->nimages, 0, 1);
  // (and the block starts of a batch written one lane per block: 0 = "no such block")
  if (A.blk_pos) add(A.blk_pos, sizeof(uint32_t)*(uint64_t)A.dc_stride*(uint64_t)b->nimages, 0, 1);
}
// The tail of a decode: prefix sums, write pass, DC values, the images' verdicts.
static int queue_tail(jga_huff_batch *b, const hj_args &A, bool split, hipStream_t st) {
  if (hj_launch_scan(&A, (int)b->total_seg, (int)b->max_nsub, st)) return jga_fail("huff: launch failed");
  if (A.blk_pos) {
    if (hj_launch_write_blocks(&A, (int)b->max_nsub, (int)b->geom.coef_blocks, st)) return jga_fail("huff: launch failed");
  }
  else if (hj_launch_write(&A, (int)b->max_nsub, st)) return jga_fail("huff: launch failed");
  // DC differences -> DC values; into the planes too unless the caller takes the array itself
  if (hj_launch_dc(&A, (int)b->total_seg, (int)b->max_seg_mcus, b->d_dcpart, split ? 0 : (int)(b->geom.coef_shorts/64), st)) {
    return jga_fail("huff: launch failed");
  }
  // (the rounds' "something moved" flags and the images' verdicts lie back to back on both sides: one copy)
  HOK(hipMemcpyAsync(b->h_ran, b->d_ran, 4*HJ_MAX_ROUNDS + 4*(size_t)b->nimages, hipMemcpyDeviceToHost, st));
  return EXIT_SUCCESS;
}
static int queue_rounds(jga_huff_batch *b, const decode_plan &P, int &round, int count, hipStream_t st, bool tail_follows = false) {
  for (int k = 0; k < count && round < HJ_MAX_ROUNDS; k++, round++) {
    if (round >= P.list_from) {
      if (hj_launch_list_round(&P.A, (int)b->max_nsub, round, round - P.list_from, P.it_list, b->list_state, st)) return jga_fail("huff: launch failed");
      b->list_state = 0;
      if (jga_tune("JGA_HUFF_LIST_STATS")) {                   // (tuning build: what the round ran and what it left, synchronously)
        std::vector<uint32_t> c((size_t)4*HJ_LIST_CSTRIDE*(size_t)b->nimages);
        HOK(hipStreamSynchronize(st));
        HOK(hipMemcpy(c.data(), b->d_lcount, 4*c.size(), hipMemcpyDeviceToHost));
        unsigned long long in = 0, out = 0, mx = 0, solo = 0;
        for (int i = 0; i < b->nimages; i++) {
          const uint32_t x = c[((size_t)(round & 3)*b->nimages + i)*HJ_LIST_CSTRIDE], y = c[((size_t)((round + 1) & 3)*b->nimages + i)*HJ_LIST_CSTRIDE];
          in += x; out += y; mx = x > mx ? x : mx; solo += x <= 256;
        }
        fprintf(stderr, "  list round %d: %llu entries (largest list %llu, %llu of %d images in one workgroup), %llu left for the next\n",
         round, in, mx, solo, b->nimages, out);
      }
      continue;
    }
    if (hj_launch_round(&P.A, (int)b->max_nsub, round, round ? P.it1 : P.it0, P.lean, st)) {
      return jga_fail("huff: launch failed");
    }
    if (b->list_state == 0) b->list_state = 2;
  }
  if (!tail_follows) HOK(hipMemcpyAsync(b->h_ran, b->d_ran, 4*HJ_MAX_ROUNDS, hipMemcpyDeviceToHost, st));   // (else the tail's copy brings them)
  return EXIT_SUCCESS;
}

static int decode_begin(jga_huff_batch *b, short *d_coef, long long coef_stride, short *d_dc, long long dc_stride,
 hipStream_t st) {
  b->pend.active = false;
  // verdicts of an earlier decode must not outlive it: a launch failure below would otherwise
  // be read as "some members were damaged" by callers that look at jga_huff_image_errors()
  b->image_errors = 0;
  for (int i = 0; i < b->max_images; i++) b->h_ran[HJ_MAX_ROUNDS + i] = 0;
  if (!b->nimages) return jga_fail("huff: nothing prepared");
  if (coef_stride < b->geom.coef_shorts) return jga_fail("huff: coef_stride too small");
  if (d_dc && dc_stride < b->geom.coef_shorts/64) return jga_fail("huff: dc_stride too small");
  decode_plan P;
  if (make_plan(b, d_coef, coef_stride, d_dc, dc_stride, P) != EXIT_SUCCESS) return EXIT_FAILURE;
  // reset: states back to the guesses, "never ran"
  // (on-device unstuffing has already had its say about every image: early end, RSTn counters)
  // (with it, as extra workgroups of the same launch, what the write pass and the DC pass need cleared)
  hj_clear_args C;
  clear_regions(b, P.A, C);
  if (hj_launch_init(&P.A, (int)b->total_seg, (int)b->max_nsub,
   b->unstuffed_on_device ? (const uint32_t *)(b->d_blob + b->off_perr) : NULL, &C, st)) {
    return jga_fail("huff: launch failed");
  }
  b->last_assisted = 0;
  b->assist_left = 0;
  b->list_state = 1;                                         // (hj_init_states zeroed the lists' counters)
  // A photograph settles in 4-6 rounds, so the tail is queued SPECULATIVELY behind the first
  // group of rounds: one host round trip per decode instead of two (a lone 1080p frame: ~60 us of
  // its ~700).  If the last of those rounds still moved something, the tail ran on unsettled
  // states — bounded like a corrupt stream's, but wrong: outputs and verdicts are reset, the
  // rounds go on, the tail runs again, and the next decode of this batch object queues more
  // rounds first (JGA_HUFF_SPECULATE=0: never).
  if (b->spec_rounds < P.group) b->spec_rounds = P.group;
  int round = 0;
  if (queue_rounds(b, P, round, P.speculate ? b->spec_rounds : P.group, st, P.speculate) != EXIT_SUCCESS) return EXIT_FAILURE;
  if (P.speculate && queue_tail(b, P.A, d_dc != NULL, st) != EXIT_SUCCESS) return EXIT_FAILURE;
  b->pend.active = true;
  b->pend.with_tail = P.speculate;
  b->pend.d_coef = d_coef; b->pend.coef_stride = coef_stride;
  b->pend.d_dc = d_dc; b->pend.dc_stride = dc_stride;
  b->pend.st = st;
  b->pend.round = round;
  return EXIT_SUCCESS;
}

static int decode_end(jga_huff_batch *b, int *valid_behind) {
  if (valid_behind) *valid_behind = 0;
  if (!b->pend.active) return jga_fail("huff: no decode in flight");
  b->pend.active = false;
  hipStream_t st = b->pend.st;
  short *d_dc = b->pend.d_dc;
  const bool trace = b->trace != 0;
  double c0 = trace ? thread_cpu_ms() : 0.0, c_launch = 0.0, c_wait = 0.0;
  auto lap = [&](double &acc) { if (trace) { const double c = thread_cpu_ms(); acc += c - c0; c0 = c; } };
  decode_plan P;
  if (make_plan(b, b->pend.d_coef, b->pend.coef_stride, d_dc, b->pend.dc_stride, P) != EXIT_SUCCESS) return EXIT_FAILURE;
  int round = b->pend.round;
  bool with_tail = b->pend.with_tail, tail_done = false, first = true;
  for (;;) {
    HOK(wait_stream(b, st));
    lap(c_wait);
    if (b->h_ran[round - 1] == 0) { tail_done = with_tail; break; }   // a round in which nothing moved
    if (with_tail) {
      // not settled: undo what the speculative tail wrote (planes, DC arrays, verdicts)
      b->spec_rounds = round + 4 < 24 ? round + 4 : 24;
      if (b->unstuffed_on_device) {
        HOK(hipMemcpyAsync(b->d_errors, b->d_blob + b->off_perr, 4*(size_t)b->nimages, hipMemcpyDeviceToDevice, st));
      }
      else HOK(hipMemsetAsync(b->d_errors, 0, 4*(size_t)b->nimages, st));
      // (the planes: every line the final write pass touches is rewritten or zeroed first)
      HOK(hipMemsetAsync(P.A.dc_diff, 0, sizeof(int16_t)*(size_t)P.A.dc_stride*(size_t)b->nimages, st));
      HOK(hipMemsetAsync(P.A.dc_val, 0, sizeof(int16_t)*(size_t)P.A.dc_stride*(size_t)b->nimages, st));
      if (P.A.blk_pos) HOK(hipMemsetAsync(P.A.blk_pos, 0, sizeof(uint32_t)*(size_t)P.A.dc_stride*(size_t)b->nimages, st));
    }
    if (round >= HJ_MAX_ROUNDS) return jga_fail("huff: synchronisation did not converge");
    // (a batch object whose previous decode needed the walk — the same camera, the same
    // letterbox — gets it at the first check instead of waiting out twelve rounds)
    if (round >= (b->assist_hint ? 1 : P.assist_after)) {
      // Streams that NEVER fall into step (flat areas, letterbox bars) need the host's walk; streams that are merely
      // slow — a q97 8K frame: 310 k subsequences, chains of thirty steps and more — do not: 6.5 ms with the walk after
      // twelve rounds, 0.9 without (round 6, tools/policy_sweep.py).  With list rounds the lists' lengths say which it
      // is: while what is left shrinks by a third from look to look, the rounds go on (up to 96).
      bool stuck = true;
      if (!b->assist_hint && round > P.list_from && round < 96) {
        std::vector<uint32_t> left((size_t)b->nimages);
        HOK(hipMemcpy2D(left.data(), sizeof(uint32_t), b->d_lcount + (size_t)(round & 3)*(size_t)b->nimages*HJ_LIST_CSTRIDE,
         sizeof(uint32_t)*HJ_LIST_CSTRIDE, sizeof(uint32_t), (size_t)b->nimages, hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (uint32_t x : left) sum += x;
        stuck = b->assist_left != 0 && sum*3 > b->assist_left*2;
        b->assist_left = sum ? sum : 1;
      }
      if (stuck && assist_chains(b, st) != EXIT_SUCCESS) return EXIT_FAILURE;
    }
    with_tail = false;
    first = false;
    if (queue_rounds(b, P, round, P.group, st) != EXIT_SUCCESS) return EXIT_FAILURE;
    lap(c_launch);
  }
  b->last_rounds = 0;
  while (b->last_rounds < round && b->h_ran[b->last_rounds]) b->last_rounds++;
  b->assist_hint = b->last_assisted > 0;
  if (tail_done && b->last_rounds + 2 < b->spec_rounds && b->spec_rounds > P.group) b->spec_rounds--;   // (drifts back)
  if (!tail_done) {
    if (queue_tail(b, P.A, d_dc != NULL, st) != EXIT_SUCCESS) return EXIT_FAILURE;
    lap(c_launch);
    HOK(wait_stream(b, st));
    lap(c_wait);
  }
  // what the caller queued behind decode_begin() ran behind the FINAL tail only if the speculation held at the
  // first look
  if (valid_behind) *valid_behind = tail_done && first;
  if (trace) fprintf(stderr, "  huff decode (second half): this thread's CPU in launches + copies %.2f ms, in waits %.2f ms\n", c_launch, c_wait);
  // per-image verdicts stay readable (jga_huff_image_error): the other images of the batch
  // are decoded correctly whatever one damaged member did
  b->image_errors = 0;
  int bad = -1;
  for (int i = 0; i < b->nimages; i++) {
    if (b->h_ran[HJ_MAX_ROUNDS + i]) {
      if (bad < 0) bad = i;
      b->image_errors++;
    }
  }
  if (bad >= 0) {
    return jga_fail("huff: image %d: %s", bad, (b->h_ran[HJ_MAX_ROUNDS + bad] & 2)
     ? "Error indexing outside block." : "Error, entropy data ended early.");
  }
  return EXIT_SUCCESS;
}

// A launch or copy failed part-way: whatever was queued may still be running — the caller is free to reuse or
// release d_coef the moment the call returns, so wait it out first.
static int drained(jga_huff_batch *b, hipStream_t st, int rc) {
  if (rc != EXIT_SUCCESS && !b->image_errors) (void)hipStreamSynchronize(st);
  return rc;
}

static int decode_checked(jga_huff_batch *b, short *d_coef, long long coef_stride, short *d_dc, long long dc_stride,
 void *stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = decode_begin(b, d_coef, coef_stride, d_dc, dc_stride, st);
  if (rc == EXIT_SUCCESS) rc = decode_end(b, NULL);
  return drained(b, st, rc);
}

// Finished QUANT-stage planes (DC prediction applied), as jga_entropy_decode() makes them on the host.
JGA_EXPORT int jga_huff_decode(jga_huff_batch *b, short *d_coef, long long coef_stride,
 void *stream) {
  return decode_checked(b, d_coef, coef_stride, NULL, 0, stream);
}

// The same decode for a caller that runs a block-decode kernel next (jga_idct_*_batch_dc): the
// planes' DC positions are left holding the DC DIFFERENCES, the DC values come in d_dc — image i
// at d_dc + i*dc_stride, one int16 per 128-byte slot of the image's coefficient buffer
// (dc_stride >= coef_shorts/64) — and the strided 2-byte pass over the planes that would put them
// in place is saved.
JGA_EXPORT int jga_huff_decode_split(jga_huff_batch *b, short *d_coef, long long coef_stride,
 short *d_dc, long long dc_stride, void *stream) {
  if (!d_dc) return jga_fail("huff: jga_huff_decode_split needs a DC array");
  return decode_checked(b, d_coef, coef_stride, d_dc, dc_stride, stream);
}
// ... in two halves: _begin queues the decode on `stream` and returns without waiting; the caller queues what
// consumes the planes (the block-decode kernel, copies of its output) on the same stream; _end waits for the
// stream, finishes the decode if the first burst of rounds had not settled it, and says in *valid_behind whether
// the work queued in between saw the final planes and DC values (1) or has to be queued again (0).  Verdicts
// (jga_huff_image_errors / _error) are those of jga_huff_decode_split.
JGA_EXPORT int jga_huff_decode_split_begin(jga_huff_batch *b, short *d_coef, long long coef_stride,
 short *d_dc, long long dc_stride, void *stream) {
  if (!d_dc) return jga_fail("huff: jga_huff_decode_split needs a DC array");
  return drained(b, (hipStream_t)stream, decode_begin(b, d_coef, coef_stride, d_dc, dc_stride, (hipStream_t)stream));
}
JGA_EXPORT int jga_huff_decode_split_end(jga_huff_batch *b, int *valid_behind) {
  hipStream_t st = b->pend.st;
  return drained(b, st, decode_end(b, valid_behind));
}

JGA_EXPORT int jga_huff_image_errors(const jga_huff_batch *b) { return b->image_errors; }
JGA_EXPORT int jga_huff_image_error(const jga_huff_batch *b, int i) {
  return (i >= 0 && i < b->nimages) ? (int)b->h_ran[HJ_MAX_ROUNDS + i] : -1;
}

}  // extern "C"
