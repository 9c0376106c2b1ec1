// vtbl.cpp — HIPJPEG_DECODE_CTX_VTBL: the decoder plugin that drops in where
// the reference's XJPEG_DECODE_CTX_VTBL sits (src/jpeg_wrap.c:246-358).
//
//   decode_alloc  <-> xjpeg_decode_alloc    jpeg_wrap.c:254-261
//   decode_header <-> xjpeg_decode_header_  jpeg_wrap.c:263-319
//   decode_image  <-> xjpeg_decode_image_   jpeg_wrap.c:321-342
//   decode_reset  <-> xjpeg_decode_reset    jpeg_wrap.c:344-346
//   decode_free   <-> xjpeg_decode_free     jpeg_wrap.c:348-350
//
// Same call order, ownership (caller owns jpeg_info.buf and the image) and
// error convention (EXIT_FAILURE + one line on stderr).  Differences, all
// additive: the YUV and RGB stages are computed on the GPU (the reference
// computes YUV on the CPU and rejects RGB, jpeg_wrap.c:335-339), and malformed
// input is rejected instead of read out of bounds.  There is NO CPU fallback
// for the GPU stages: without a HIP device they fail loudly.
//
// For the YUV and RGB stages the scan itself is decoded on the GPU too
// (jga_huff_*): the host only parses the markers and unstuffs the
// scan into pinned memory.  jga_plugin_config.host_entropy selects the host entropy
// stage (csrc/entropy.c) instead; it is also what a file whose Huffman tables do
// not fit the device lookup format ("too irregular") is decoded with.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "jga_internal.h"

namespace {

const char *const OUT_NAMES[JPEG_DECODE_OUT_MAX] =
 {"pack", "quant", "dct", "yuv", "rgb"};

struct hipjpeg_ctx {
  const unsigned char *buf;
  int size;
  int have_header;
  jpeg_header header;
  jga_geom geom;
  // device side, created on the first GPU-stage decode and reused per frame
  hipStream_t stream;
  short *h_coef;              // pinned
  unsigned char *h_out;       // pinned
  short *d_coef;
  short *d_dc;                // DC values beside the planes (jga_huff_decode_split_begin)
  long long cap_dc;
  unsigned short *d_qtab;
  unsigned char *d_out;
  long long cap_coef, cap_out;
  jga_huff_batch *hb;         // GPU entropy stage, one image
  long long hb_scan;
  // caller buffers (img->pixels, plane data) registered with HIP so that the D2H copy
  // lands in them directly; the harness decodes into the same image every frame
  struct { void *ptr; size_t bytes; } reg[6];
  hipEvent_t ev_piece[8];     // copy_back_staged
  struct copy_team *team;     // ... and its two helpers, started by the first big frame
  struct { int register_buffers, host_entropy, copy_team; } opt;   // jga_plugin_configure, as of decode_alloc
};

// Two helper threads that move pieces of a frame from the pinned staging buffer into the caller's
// memory while the calling thread waits for the next piece to arrive (and then lends a hand):
// one core copies 15-28 GB/s depending on the box, the link carries 56 — alone, the calling thread
// was what a 4K RGB frame waited for (1.0 ms of copying against 0.45 ms of link).
struct copy_team {
  struct task { unsigned char *dst; const unsigned char *src; size_t len; };
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<task> q;
  size_t head = 0;
  int pending = 0;
  bool quit = false;
  std::thread th[2];
  copy_team() { for (auto &t : th) t = std::thread([this] { run(); }); }
  ~copy_team() {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    cv_work.notify_all();
    for (auto &t : th) t.join();
  }
  void done_one() {
    std::lock_guard<std::mutex> lk(m);
    if (--pending == 0) cv_done.notify_all();
  }
  void run() {
    for (;;) {
      task t;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [this] { return quit || head < q.size(); });
        if (head >= q.size()) return;            // quit
        t = q[head++];
      }
      memcpy(t.dst, t.src, t.len);
      done_one();
    }
  }
  void push(const task &t) {
    { std::lock_guard<std::mutex> lk(m); q.push_back(t); pending++; }
    cv_work.notify_one();
  }
  void finish() {                                // the caller takes what is left, then waits for the helpers
    for (;;) {
      task t;
      {
        std::lock_guard<std::mutex> lk(m);
        if (head >= q.size()) break;
        t = q[head++];
      }
      memcpy(t.dst, t.src, t.len);
      done_one();
    }
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [this] { return pending == 0; });
    q.clear();
    head = 0;
  }
};

// The plugin's settings (jga_plugin_configure): process-wide, copied into a context when it is
// allocated.  Builds made with -DJGA_TUNING also listen to the JGA_PLUGIN_* variables of rounds 2-3.
struct plugin_settings { int register_buffers, host_entropy, copy_team; };
std::mutex g_settings_mutex;
plugin_settings g_settings = {0, 0, 0};
plugin_settings current_settings() {
  std::lock_guard<std::mutex> lk(g_settings_mutex);
  plugin_settings s = g_settings;
  if (const char *e = jga_tune("JGA_PLUGIN_ENTROPY")) s.host_entropy = strcmp(e, "host") == 0;
  if (const char *e = jga_tune("JGA_PLUGIN_REGISTER")) s.register_buffers = atoi(e) < 0 ? -1 : atoi(e) > 0 ? 1 : 0;
  if (const char *e = jga_tune("JGA_PLUGIN_COPY_TEAM")) s.copy_team = atoi(e) == 0 ? -1 : 0;
  return s;
}

// True if [p, p+bytes) is (now) registered.  OPT-IN (jga_plugin_config.register_buffers): it is only
// safe for a caller that keeps the image's buffers alive and in place for as long as the
// decoder context lives (the harness does; a caller that frees and re-allocates its image
// between frames would leave a stale registration behind).  Default: the staged copy.
bool registered(hipjpeg_ctx *c, void *p, size_t bytes) {
  if (c->opt.register_buffers <= 0 || !p || !bytes) return false;
  int free_slot = -1;
  for (int i = 0; i < 6; i++) {
    if (c->reg[i].ptr == p && c->reg[i].bytes >= bytes) return true;
    if (c->reg[i].ptr == p) { (void)hipHostUnregister(p); c->reg[i].ptr = NULL; }
    if (!c->reg[i].ptr && free_slot < 0) free_slot = i;
  }
  if (free_slot < 0) {                     // a different image: start over
    for (int i = 0; i < 6; i++) { (void)hipHostUnregister(c->reg[i].ptr); c->reg[i].ptr = NULL; }
    free_slot = 0;
  }
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  c->reg[free_slot].ptr = p;
  c->reg[free_slot].bytes = bytes;
  return true;
}

// How the caller's buffers meet the device (jga_plugin_config.register_buffers; round 4 measured all four ways,
// profiles/r4_host_side_steps.md §2):
//    0  (default) the copies name the caller's ORDINARY memory and the runtime does the rest — it pins what a copy
//       touches and keeps that cached, so a frame's pixels arrive at link speed without any promise from the caller
//       and without a pass of a host core over them: one 4K frame 1.08 ms, the 8K frame of config 5 2.71 ms, the same
//       as with registered buffers, and the caller's own later copies of those buffers run at link speed too
//       (harness, 4K -o rgb: 657 FPS; rounds 2-3's staged default: 527-540)
//    1  buffers registered by the plugin for the life of the decoder context (hipHostRegister): the same times; for
//       callers that keep image and file in place (the reference's main loop does) and want no first-frame pinning
//   -1  copies staged through the context's pinned buffers, the pieces moved into the caller's memory by host threads
//       (rounds 2-3's default): 0.75 / 1.49 / 3.96 ms for a 1080p / 4K / 8K frame
// Tried and dropped: registering each buffer for the length of ONE decode_image call (hipHostRegister 40-60 us,
// hipHostUnregister 1 us: tools/register_probe.py) — the fastest decode_image in isolation, but the release throws
// away the runtime's cached pinning, so a caller that then copies the same buffer itself (the reference's
// glTexSubImage2D; the harness's upload) pays for it: 4K -o rgb 326 FPS against 540 staged.
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return jga_fail("hipjpeg: HIP error %d (%s) at %s", (int)e_, \
  hipGetErrorString(e_), #call); } while (0)

// Staged copy back, pipelined: the device -> pinned staging copy goes out in pieces, and while piece
// k+1 crosses the link this thread moves piece k from the staging buffer into the caller's memory
// (one core copies ~28 GB/s, the link carries 56: the frame's pixels arrive in about the time the
// slower of the two takes, not in their sum — a 4K RGB frame 1.3 -> 0.9 ms).
// `bytes` of device memory at d_src land in pinned staging at h_stage and from there in the caller's
// memory, which is `nseg` separate buffers: seg[k].len bytes at seg[k].dst, in the order they lie in
// d_src (a frame's RGB pixels are one segment, its Y/Cb/Cr planes three).
struct out_segment { unsigned char *dst; size_t len; };
void scatter(const out_segment *seg, int nseg, size_t o, size_t len, const unsigned char *h_stage, copy_team *team) {
  size_t base = 0;
  for (int k = 0; k < nseg && len; k++) {       // the piece [o, o + len) may straddle two of the caller's buffers
    if (o < base + seg[k].len) {
      const size_t in = o - base, take = seg[k].len - in < len ? seg[k].len - in : len;
      if (team) team->push({seg[k].dst + in, h_stage + o, take});
      else memcpy(seg[k].dst + in, h_stage + o, take);
      o += take; len -= take;
    }
    base += seg[k].len;
  }
}
int copy_back_staged(hipjpeg_ctx *c, const out_segment *seg, int nseg, const unsigned char *d_src, unsigned char *h_stage, size_t bytes) {
  // A small frame (a thumbnail, a 1080p plane set): one copy, one wait, one pass over the host memory —
  // cutting it into pieces only adds launches and event waits that the overlap cannot win back
  // (a YUV frame used to issue up to 18 copies + event waits, a plane at a time).
  if (bytes < ((size_t)2 << 20)) {
    HIP_OK(hipMemcpyAsync(h_stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    scatter(seg, nseg, 0, bytes, h_stage, NULL);
    return EXIT_SUCCESS;
  }
  const bool teamed = c->opt.copy_team >= 0 && bytes >= ((size_t)12 << 20);      // (a 1080p frame's 6 MB: the calling thread alone is quicker)
  if (teamed && !c->team) c->team = new copy_team();
  const int PIECES = teamed ? 8 : 6;
  const size_t piece = ((bytes + PIECES - 1)/PIECES + 4095) & ~(size_t)4095;
  int n = 0;
  for (size_t o = 0; o < bytes; o += piece, n++) {
    const size_t len = bytes - o < piece ? bytes - o : piece;
    if (!c->ev_piece[n]) HIP_OK(hipEventCreateWithFlags(&c->ev_piece[n], hipEventDisableTiming));
    HIP_OK(hipMemcpyAsync(h_stage + o, d_src + o, len, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipEventRecord(c->ev_piece[n], c->stream));
  }
  n = 0;
  for (size_t o = 0; o < bytes; o += piece, n++) {
    const size_t len = bytes - o < piece ? bytes - o : piece;
    const hipError_t e = hipEventSynchronize(c->ev_piece[n]);
    if (e != hipSuccess) {
      if (teamed) c->team->finish();               // (nothing of ours may still be writing into the caller's memory)
      return jga_fail("hipjpeg: HIP error %d (%s) waiting for a piece of the frame", (int)e, hipGetErrorString(e));
    }
    scatter(seg, nseg, o, len, h_stage, teamed ? c->team : NULL);
  }
  if (teamed) c->team->finish();
  return EXIT_SUCCESS;
}

void release_device(hipjpeg_ctx *c) {
  delete c->team;
  c->team = NULL;
  for (int i = 0; i < 8; i++) {
    if (c->ev_piece[i]) (void)hipEventDestroy(c->ev_piece[i]);
    c->ev_piece[i] = NULL;
  }
  for (int i = 0; i < 6; i++) {
    if (c->reg[i].ptr) (void)hipHostUnregister(c->reg[i].ptr);
    c->reg[i].ptr = NULL;
  }
  if (c->hb) jga_huff_destroy(c->hb);
  c->hb = NULL; c->hb_scan = 0;
  if (c->h_coef) (void)hipHostFree(c->h_coef);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->d_coef) (void)hipFree(c->d_coef);
  if (c->d_dc) (void)hipFree(c->d_dc);
  if (c->d_qtab) (void)hipFree(c->d_qtab);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  c->h_coef = NULL; c->h_out = NULL; c->d_coef = NULL; c->d_qtab = NULL; c->d_dc = NULL; c->cap_dc = 0;
  c->d_out = NULL; c->stream = NULL; c->cap_coef = c->cap_out = 0;
}

int ensure_device(hipjpeg_ctx *c, long long coef_shorts, long long out_bytes) {
  if (!c->stream) HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  if (!c->d_qtab) HIP_OK(hipMalloc((void **)&c->d_qtab, 3*64*sizeof(unsigned short)));
  if (coef_shorts > c->cap_coef) {
    if (c->h_coef) (void)hipHostFree(c->h_coef);
    if (c->d_coef) (void)hipFree(c->d_coef);
    c->h_coef = NULL; c->d_coef = NULL; c->cap_coef = 0;
    HIP_OK(hipHostMalloc((void **)&c->h_coef, coef_shorts*sizeof(short), hipHostMallocDefault));
    HIP_OK(hipMalloc((void **)&c->d_coef, coef_shorts*sizeof(short)));
    c->cap_coef = coef_shorts;
  }
  if (out_bytes > c->cap_out) {
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->d_out) (void)hipFree(c->d_out);
    c->h_out = NULL; c->d_out = NULL; c->cap_out = 0;
    HIP_OK(hipHostMalloc((void **)&c->h_out, out_bytes, hipHostMallocDefault));
    HIP_OK(hipMalloc((void **)&c->d_out, out_bytes));
    c->cap_out = out_bytes;
  }
  return EXIT_SUCCESS;
}

jpeg_decode_ctx *hipjpeg_alloc(jpeg_info *info) {
  hipjpeg_ctx *c = (hipjpeg_ctx *)calloc(1, sizeof(hipjpeg_ctx));
  if (c != NULL) {
    const plugin_settings s = current_settings();
    c->buf = info->buf;
    c->size = info->size;
    c->opt.register_buffers = s.register_buffers; c->opt.host_entropy = s.host_entropy;
    c->opt.copy_team = s.copy_team;
  }
  return (jpeg_decode_ctx *)c;
}

int hipjpeg_header(jpeg_decode_ctx *dec, jpeg_header *headers) {
  hipjpeg_ctx *c = (hipjpeg_ctx *)dec;
  int i;
  if (jga_parse_header(c->buf, c->size, &c->header) != EXIT_SUCCESS) {
    return EXIT_FAILURE;
  }
  if (jga_geom_from_header(&c->geom, &c->header) != EXIT_SUCCESS) {
    return EXIT_FAILURE;
  }
  c->have_header = 1;
  *headers = c->header;
  for (i = 0; i < headers->ncomps; i++) {
    // comp[i].quant points INTO its own header (jpeg_wrap.c:313)
    headers->comp[i].quant =
     &headers->quant[c->header.comp[i].quant - c->header.quant];
  }
  return EXIT_SUCCESS;
}

int check_image(const hipjpeg_ctx *c, const image *img) {
  int i;
  if (img->nplanes != c->geom.nplanes || img->width != c->geom.width
   || img->height != c->geom.height) {
    return jga_fail("hipjpeg: image does not match the JPEG headers");
  }
  for (i = 0; i < img->nplanes; i++) {
    if (img->plane[i].width != c->geom.plane[i].hblocks*8
     || img->plane[i].height != c->geom.plane[i].vblocks*8
     || img->plane[i].ystride != img->plane[i].width
     || img->plane[i].xstride != 1) {
      return jga_fail("hipjpeg: image plane %d does not match the JPEG headers", i);
    }
  }
  return EXIT_SUCCESS;
}

int hipjpeg_image(jpeg_decode_ctx *dec, image *img, jpeg_decode_out out) {
  hipjpeg_ctx *c = (hipjpeg_ctx *)dec;
  const jga_geom *g = &c->geom;
  unsigned short qtab[3*64];
  int i;
  if (!c->have_header) {
    // the reference's header pass parks on SOS (xjpeg.c:716-719): same rule
    return jga_fail("hipjpeg: decode_header must precede decode_image");
  }
  if (check_image(c, img) != EXIT_SUCCESS) return EXIT_FAILURE;
  switch (out) {
    case JPEG_DECODE_PACK : {
      long long words = 0, per_plane[3] = {0, 0, 0};
      if (jga_entropy_decode_pack(c->buf, c->size, g, img->coef,
       g->coef_shorts, img->index, &words, per_plane) != EXIT_SUCCESS) {
        return EXIT_FAILURE;
      }
      for (i = 0; i < img->nplanes; i++) img->plane[i].packed = (int)per_plane[i];
      img->packed = (int)words;
      return EXIT_SUCCESS;
    }
    case JPEG_DECODE_QUANT :
    case JPEG_DECODE_DCT : {
      return jga_entropy_decode(c->buf, c->size, g, img->coef,
       out == JPEG_DECODE_DCT);
    }
    case JPEG_DECODE_YUV :
    case JPEG_DECODE_RGB : break;
    default : {
      return jga_fail("Unsupported output '%s' for hipjpeg wrapper.",
       (unsigned)out < JPEG_DECODE_OUT_MAX ? OUT_NAMES[out] : "?");
    }
  }
  // GPU stages: host entropy decode -> pinned -> H2D -> fused kernel -> D2H
  {
    const int rgb = out == JPEG_DECODE_RGB;
    const long long out_bytes = rgb ? g->rgb_bytes : g->yuv_bytes;
    if (ensure_device(c, g->coef_shorts, (out_bytes + 15) & ~15ll) != EXIT_SUCCESS) {
      return EXIT_FAILURE;
    }
    int on_gpu = !c->opt.host_entropy;
    struct call_registration {                                 // the caller's file, registered until this call returns
      void *p = nullptr;
      hipStream_t stream = nullptr;
      ~call_registration() {
        if (!p) return;
        (void)hipStreamSynchronize(stream);                    // (no copy out of it may still be in flight, whatever the way out)
        (void)hipHostUnregister(p);
      }
    } file_of_this_call;
    const bool direct = c->opt.register_buffers == 0;         // (1: registered for the life of the context; -1: staged copies)
    const long long dcstride = (g->coef_shorts/64 + 127) & ~127ll;
    if (on_gpu) {
      jga_geom g2;
      if (!c->hb || c->size + 4096ll > c->hb_scan) {
        if (c->hb) jga_huff_destroy(c->hb);
        c->hb_scan = c->size + c->size/4 + 4096ll;
        c->hb = jga_huff_create(1, c->hb_scan);
        if (!c->hb) { c->hb_scan = 0; return EXIT_FAILURE; }
        jga_huff_set_threads(c->hb, 1);
      }
      if (dcstride > c->cap_dc) {
        if (c->d_dc) (void)hipFree(c->d_dc);
        c->d_dc = NULL; c->cap_dc = 0;
        HIP_OK(hipMalloc((void **)&c->d_dc, (size_t)dcstride*sizeof(short)));
        c->cap_dc = dcstride;
      }
      // (a caller that lets its buffers be registered: a big file is read where it lies and the
      // device cleans the scan up — the host's pass over the entropy-coded bytes takes one core
      // 0.2 ms for a 4K file, the four launches of the device's ~0.1 ms whatever the size: a 4K frame
      // 1.15 -> 1.08 ms, 8K 3.5 -> 2.7, a 1080p frame is better off with the host's)
      // Without a promise from the caller (register_buffers = 0) the file is registered for the length of THIS call
      // (round 5; 45 us for 3 MB).  Until then the upload simply named the caller's memory and the runtime pinned what
      // it touched — read-only, as the source of a copy, and kept that pinning cached: a caller that freed the file
      // and got the same heap memory back for its PIXELS had the copy back fault ("Memory access fault by GPU ...
      // Write access to a read-only page", bench.py's configs leg: a 6 MB 4:4:4 file, then a 6 MB 1080p frame).
      bool file_pinned = c->size >= (3 << 19) && (direct || registered(c, const_cast<unsigned char *>(c->buf), (size_t)c->size));
      if (file_pinned && direct) {
        if (hipHostRegister(const_cast<unsigned char *>(c->buf), (size_t)c->size, hipHostRegisterDefault) == hipSuccess) {
          file_of_this_call.p = const_cast<unsigned char *>(c->buf);
          file_of_this_call.stream = c->stream;
        }
        else {
          (void)hipGetLastError();
          file_pinned = false;                                 // (somebody else's registration, or none to be had: the host reads the file)
        }
      }
      jga_huff_set_device_unstuff(c->hb, file_pinned);
      jga_huff_set_inputs_pinned(c->hb, file_pinned);
      if (jga_huff_prepare(c->hb, &c->buf, &c->size, 1, &g2, c->stream) != EXIT_SUCCESS) {
        if (jga_huff_prepare_verdict(c->hb, 0) != 2) return EXIT_FAILURE;
        on_gpu = 0;                        // tables / frame size outside the device format
      }
      // the decode's first half: everything is queued, nothing waited for — the block decode and the copy back
      // go in behind it, and the frame costs ONE host wait
      else if (jga_huff_decode_split_begin(c->hb, c->d_coef, g->coef_shorts, c->d_dc, dcstride, c->stream) != EXIT_SUCCESS) {
        return EXIT_FAILURE;
      }
    }
    const unsigned short *d_q = c->d_qtab;
    if (on_gpu) d_q = jga_huff_qtabs_device(c->hb);            // (they came up with the file's descriptors)
    else {
      if (jga_entropy_decode(c->buf, c->size, g, c->h_coef, 0) != EXIT_SUCCESS) return EXIT_FAILURE;
      memset(qtab, 0, sizeof(qtab));
      for (i = 0; i < g->nplanes; i++) {
        memcpy(qtab + 64*i, c->header.comp[i].quant->tbl, 64*sizeof(unsigned short));
      }
      HIP_OK(hipMemcpyAsync(c->d_qtab, qtab, sizeof(qtab), hipMemcpyHostToDevice, c->stream));
      HIP_OK(hipStreamSynchronize(c->stream));                 // (qtab is on this stack frame)
      HIP_OK(hipMemcpyAsync(c->d_coef, c->h_coef, g->coef_shorts*sizeof(short),
       hipMemcpyHostToDevice, c->stream));
    }
    // D2H straight into the caller's buffers when the copies may name them, else through
    // the pinned staging buffer (which does its own waiting)
    bool in_place = true;
    if (rgb) in_place = direct || registered(c, img->pixels, (size_t)out_bytes);
    else {
      for (i = 0; i < img->nplanes; i++) {
        in_place = in_place && (direct || registered(c, img->plane[i].data, (size_t)img->plane[i].ystride*img->plane[i].height));
      }
    }
    auto queue_behind = [&]() -> int {
      const short *dcv = on_gpu ? c->d_dc : NULL;
      if ((rgb ? jga_idct_rgb_batch_dc(g, 1, c->d_coef, g->coef_shorts, dcv, dcstride, d_q, 1, c->d_out, c->cap_out, c->stream)
       : jga_idct_yuv_batch_dc(g, 1, c->d_coef, g->coef_shorts, dcv, dcstride, d_q, 1, c->d_out, c->cap_out, c->stream)) != EXIT_SUCCESS) {
        return EXIT_FAILURE;
      }
      if (!in_place) return EXIT_SUCCESS;
      if (rgb) HIP_OK(hipMemcpyAsync(img->pixels, c->d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
      else {
        for (int k = 0; k < img->nplanes; k++) {
          HIP_OK(hipMemcpyAsync(img->plane[k].data, c->d_out + g->plane[k].data_off,
           (size_t)img->plane[k].ystride*img->plane[k].height, hipMemcpyDeviceToHost, c->stream));
        }
      }
      return EXIT_SUCCESS;
    };
    // (a failure from here on must not leave copies into the caller's buffers in flight)
    int rc = queue_behind();
    if (rc == EXIT_SUCCESS && on_gpu) {
      int valid_behind = 0;
      rc = jga_huff_decode_split_end(c->hb, &valid_behind);    // (waits for the stream)
      if (rc == EXIT_SUCCESS && !valid_behind) {
        rc = queue_behind();
        if (rc == EXIT_SUCCESS && hipStreamSynchronize(c->stream) != hipSuccess) rc = jga_fail("hipjpeg: device wait failed");
      }
    }
    else if (rc == EXIT_SUCCESS && hipStreamSynchronize(c->stream) != hipSuccess) rc = jga_fail("hipjpeg: device wait failed");
    if (rc != EXIT_SUCCESS) {
      (void)hipStreamSynchronize(c->stream);
      return EXIT_FAILURE;
    }
    if (!in_place) {
      if (rgb) {
        const out_segment one = {img->pixels, (size_t)out_bytes};
        if (copy_back_staged(c, &one, 1, c->d_out, c->h_out, (size_t)out_bytes) != EXIT_SUCCESS) return EXIT_FAILURE;
      }
      else {
        // (the padded planes lie back to back in d_out, g->plane[i].data_off: ONE staged copy of all of
        // them, scattered into the caller's three buffers as the pieces arrive)
        out_segment planes[NPLANES_MAX];
        for (i = 0; i < img->nplanes; i++) planes[i] = {img->plane[i].data, (size_t)img->plane[i].ystride*img->plane[i].height};
        if (copy_back_staged(c, planes, img->nplanes, c->d_out, c->h_out, (size_t)out_bytes) != EXIT_SUCCESS) return EXIT_FAILURE;
      }
    }
  }
  return EXIT_SUCCESS;
}

void hipjpeg_reset(jpeg_decode_ctx *dec, jpeg_info *info) {
  hipjpeg_ctx *c = (hipjpeg_ctx *)dec;
  c->buf = info->buf;
  c->size = info->size;
  c->have_header = 0;
}

void hipjpeg_free(jpeg_decode_ctx *dec) {
  hipjpeg_ctx *c = (hipjpeg_ctx *)dec;
  if (c) {
    release_device(c);
    free(c);
  }
}

}  // namespace

extern "C" JGA_EXPORT int jga_plugin_configure(const jga_plugin_config *cfg) {
  if (!cfg || cfg->struct_size != (int)sizeof(jga_plugin_config)) {
    return jga_fail("plugin: caller built against another revision of jpeg_gpu_amd.h (jga_plugin_config %d bytes, this "
     "library: %d) - use jga_plugin_config_init()", cfg ? cfg->struct_size : 0, (int)sizeof(jga_plugin_config));
  }
  std::lock_guard<std::mutex> lk(g_settings_mutex);
  g_settings.register_buffers = cfg->register_buffers < 0 ? -1 : cfg->register_buffers > 0 ? 1 : 0;
  g_settings.host_entropy = cfg->host_entropy != 0;
  g_settings.copy_team = cfg->copy_team < 0 ? -1 : 0;
  return EXIT_SUCCESS;
}

extern "C" JGA_EXPORT const jpeg_decode_ctx_vtbl HIPJPEG_DECODE_CTX_VTBL = {
  hipjpeg_alloc,
  hipjpeg_header,
  hipjpeg_image,
  hipjpeg_reset,
  hipjpeg_free
};
