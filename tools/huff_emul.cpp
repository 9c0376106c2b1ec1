// huff_emul.cpp — CPU emulation of the GPU entropy stage (test infrastructure).
// Runs the SAME host+device core (csrc/huff_common.h) and the same pass structure
// as csrc/huff_kernels.hip, one "lane" at a time, so that the synchronisation
// algorithm can be validated on a machine without a GPU (tests/test_huff_emul.py).
// Built into tools/bin/libhuff_emul.so by the test; never part of the product library.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../jpeg_gpu_amd/csrc/huff_prepare.h"

static const int DEZZ_INIT = 0;
static int DEZZ[64];
static void init_dezz() {
  int k = 0;
  for (int s = 0; s < 15; s++) for (int i = 0; i <= s; i++) {
    int r = (s & 1) ? i : s - i, c = s - r;
    if (r < 8 && c < 8) DEZZ[k++] = r*8 + c;
  }
}

struct block_out {
  const hj_image *im; const hj_segment *seg; short *coef; uint32_t b0; short blk[64];
  uint32_t n = 0;                    // complete blocks written so far
  void flush_complete(bool waiting, int slot, bool head) {
    if (waiting) { flush(n, slot, true, head); n++; }
  }
  void flush_partial(int slot, bool head) { flush(n, slot, false, head); }
  void put(int idx, int v) { blk[idx] = (short)v; }
  bool any(bool x) const { return x; }
  bool flush_due(bool waiting, bool) const { return waiting; }
  void flush(uint32_t n, int slot, bool complete, bool head) {
    const uint32_t b = b0 + n;
    short *dst = coef + hj_block_offset(*im, seg->mcu0 + b/(uint32_t)im->nslots, slot);
    for (int q = 0; q < 64; q++) {
      if (complete && head) dst[q] = blk[q];
      else if (blk[q]) dst[q] = blk[q];
      blk[q] = 0;
    }
  }
};

static int g_sub_log2 = HJ_SUB_LOG2_MAX;
// Subsequence length (32, 64 or 128 bytes) of the following decodes.
extern "C" __attribute__((visibility("default")))
int huff_emul_set_sub(int bytes) {
  if (bytes != 32 && bytes != 64 && bytes != 128) return 1;
  g_sub_log2 = bytes == 32 ? 5 : bytes == 64 ? 6 : 7;
  return 0;
}

// The subsequence length a batch of the given size gets (huff_common.h: hj_choose_sub_log2), in bytes.
extern "C" __attribute__((visibility("default")))
int huff_emul_choose_sub(unsigned long long scan_bytes, int nslots, int restart_interval) {
  return 1 << hj_choose_sub_log2(scan_bytes, nslots, restart_interval);
}

// hj_prepare_head's verdict on a file (0 usable, 1 not, 2 = HJ_PREPARE_IRREGULAR: valid, but
// the device format cannot hold it — the host entropy stage takes it).
extern "C" __attribute__((visibility("default")))
int huff_emul_prepare_head(const unsigned char *jpeg, int size) {
  hj_prepared P;
  return hj_prepare_head(jpeg, size, &P);
}

// The host's clean-up of a file's scan (hj_prepare_scan): clean bytes, segment (start, end)
// pairs.  Returns hj_prepare_image's code (0 ok); *scan_len / *nseg are set when it is 0.
extern "C" __attribute__((visibility("default")))
int huff_emul_clean_scan(const unsigned char *jpeg, int size, unsigned char *clean, long long cap,
 unsigned *scan_len, unsigned *segs, int seg_cap, int *nseg, int *scan_off) {
  hj_prepared P;
  P.sub_log2 = g_sub_log2;
  const int rc = hj_prepare_head(jpeg, size, &P);
  if (rc != EXIT_SUCCESS) return rc;
  if (scan_off) *scan_off = size - (int)P.avail;
  P.clean.assign((size_t)P.avail + 16, 0);
  if (hj_prepare_scan(jpeg, size, &P, P.clean.data()) != EXIT_SUCCESS) return 1;
  if ((long long)P.scan_len > cap || (int)P.segs.size() > seg_cap) return 3;
  memcpy(clean, P.clean.data(), P.scan_len);
  *scan_len = P.scan_len;
  *nseg = (int)P.segs.size();
  for (size_t k = 0; k < P.segs.size(); k++) { segs[2*k] = P.segs[k].start; segs[2*k + 1] = P.segs[k].end; }
  return 0;
}

// The packs of the AC tables (hj_tables) must not change what a synchronisation run computes,
// whatever the bits are: run hj_sync_decode over `data` (any bytes, not a scan) from `nstarts`
// start states spread over it, once with the file's tables and once with the packs removed,
// and count the runs that differ (end state, blocks).  Returns that count, or -1.
extern "C" __attribute__((visibility("default")))
int huff_emul_pack_mismatches(const unsigned char *jpeg, int size, const unsigned char *data, int ndata,
 int nstarts, long long *packed_steps) {
  hj_prepared P;
  if (hj_prepare_head(jpeg, size, &P) != EXIT_SUCCESS) return -1;
  hj_tables plain = P.tabs;
  long long packs = 0;
  for (int t = 0; t < 2; t++) {
    for (int j = 0; j < (1 << HJ_FAST_BITS); j++) { packs += (plain.ac[t][j] >> 16) != 0; plain.ac[t][j] &= 0xffffu; }
  }
  if (packed_steps) *packed_steps = packs;
  // the same tables with the 12-bit packs of small batches (hj_ltables_wide, as the kernel stages them in LDS)
  hj_prepare_wide(&P);
  std::vector<hj_ltables_wide> wide(1);
  for (int j = 0; j < (2 << HJ_FAST_BITS); j++) (&wide[0].dc[0][0])[j] = (&P.tabs.dc[0][0])[j];
  memcpy(wide[0].ac, P.wide.data(), sizeof(wide[0].ac));
  memcpy(wide[0].l2, P.tabs.l2, sizeof(wide[0].l2));
  for (int t = 0; t < 2; t++) for (int j = 0; j < (1 << HJ_WIDE_BITS); j++) {
    // the first symbol of a wide entry is the 9-bit table's
    if ((wide[0].ac[t][j] & 0xffffu) != (P.tabs.ac[t][j >> (HJ_WIDE_BITS - HJ_FAST_BITS)] & 0xffffu)) return -1;
  }
  std::vector<unsigned char> buf(data, data + ndata);
  buf.resize((size_t)ndata + 16, 0xFF);
  hj_mem_src src;
  src.s = buf.data();
  int bad = 0;
  for (int i = 0; i < nstarts; i++) {
    const uint64_t p = (uint64_t)i*(uint64_t)(ndata - 160)*8/(uint64_t)nstarts + (uint64_t)(i % 8);
    const int c = i % P.im.nslots, k = (i*7) % 64;
    const uint64_t stop = p + 1024;
    const hj_run a = hj_sync_decode(src, P.im, &P.tabs, hj_pack(p, c, k), stop, (i & 1) != 0);
    const hj_run b = hj_sync_decode(src, P.im, &plain, hj_pack(p, c, k), stop, (i & 1) != 0);
    if (a.end_state != b.end_state || a.nblocks != b.nblocks) bad++;
    const hj_run w = hj_sync_decode(src, P.im, &wide[0], hj_pack(p, c, k), stop, (i & 1) != 0);
    if (w.end_state != b.end_state || w.nblocks != b.nblocks) bad++;
  }
  return bad;
}

static int g_assist_after = 0;
// After this many rounds without settling, the host walk of huff_api.cpp's assist_chains()
// (hj_walk_unsettled) is applied once per further round; 0 = never.  Returns walked
// subsequences of the last decode through huff_emul_walked().
static long long g_walked = 0;
extern "C" __attribute__((visibility("default")))
void huff_emul_set_assist(int rounds) { g_assist_after = rounds; }
extern "C" __attribute__((visibility("default")))
long long huff_emul_walked(void) { return g_walked; }

extern "C" __attribute__((visibility("default")))
int huff_emul_decode(const unsigned char *jpeg, int size, short *coef, long long coef_shorts,
 int jacobi, int *rounds_out, int *nsub_out, long long *runs_out) {
  hj_prepared P;
  (void)DEZZ_INIT;
  P.sub_log2 = g_sub_log2;
  if (hj_prepare_image(jpeg, size, &P) != EXIT_SUCCESS) return 1;
  if (coef_shorts < P.geom.coef_shorts) return 2;
  init_dezz();
  const uint32_t nsub = P.im.nsub;
  std::vector<uint64_t> S(nsub + P.segs.size()), last_in(nsub, ~0ull);
  std::vector<hj_run> R(nsub);
  std::vector<uint32_t> sub_seg(nsub);
  // initial states: true start for lane 0 of each segment, guesses elsewhere
  for (size_t si = 0; si < P.segs.size(); si++) {
    const hj_segment &sg = P.segs[si];
    for (uint32_t i = 0; i < sg.nsub; i++) {
      sub_seg[sg.sub0 + i] = (uint32_t)si;
      const uint32_t byte = sg.start + (i << g_sub_log2);
      S[sg.sub0 + si + i] = hj_pack((uint64_t)byte*8, 0, 0);
    }
    S[sg.sub0 + si + sg.nsub] = 0;
  }
  int rounds = 0;
  long long runs = 0;
  g_walked = 0;
  for (;;) {
    bool ran = false;
    // jacobi: every lane of a round reads the states as they were when the round
    // began (what a GPU launch does at worst); otherwise lanes see earlier lanes' updates
    std::vector<uint64_t> snap;
    if (jacobi) snap = S;
    for (uint32_t g = 0; g < nsub; g++) {                 // "lanes"
      const uint32_t si = sub_seg[g];
      const hj_segment &sg = P.segs[si];
      const uint32_t i = g - sg.sub0;
      const uint64_t start = jacobi ? snap[g + si] : S[g + si];
      if (start == last_in[g]) continue;
      uint32_t stop_byte = sg.start + ((i + 1) << g_sub_log2);
      if (stop_byte > sg.end) stop_byte = sg.end;
      hj_mem_src src; src.s = P.clean.data();
      R[g] = hj_sync_decode(src, P.im, &P.tabs, start, (uint64_t)stop_byte*8, i + 1 >= sg.nsub);
      last_in[g] = start;
      if (i + 1 < sg.nsub) S[g + si + 1] = R[g].end_state;
      ran = true;
      runs++;
    }
    if (!ran) break;
    if (getenv("EMUL_VERBOSE")) {
      long changed = 0, only_c = 0;
      if (jacobi) for (size_t q = 0; q < S.size(); q++) if (S[q] != snap[q]) {
        changed++;
        if ((S[q] >> 16) == (snap[q] >> 16) && (S[q] & 255) == (snap[q] & 255)) only_c++;
      }
      fprintf(stderr, "round %d: states changed %ld (only slot differs: %ld)\n", rounds, changed, only_c);
    }
    rounds++;
    if (rounds > (int)nsub + 4) return 3;
    if (g_assist_after > 0 && rounds >= g_assist_after) {
      g_walked += hj_walk_unsettled(P.im, P.segs.data(), &P.tabs, P.clean.data(), S.data(), last_in.data(),
       g_sub_log2);
    }
  }
  // per-segment exclusive prefix sums + write pass
  for (size_t si = 0; si < P.segs.size(); si++) {
    const hj_segment &sg = P.segs[si];
    uint32_t b = 0;
    int dc[3] = {0, 0, 0};           // DC predictors, carried from lane to lane of the segment
    const uint32_t total = sg.nmcu*(uint32_t)P.im.nslots;
    for (uint32_t i = 0; i < sg.nsub; i++) {
      const uint32_t g = sg.sub0 + i;
      const uint64_t start = S[g + si];
      if (hj_slot(start) != (int)(b % (uint32_t)P.im.nslots) && b < total) return 4;
      block_out bo;
      memset(bo.blk, 0, sizeof(bo.blk));
      bo.im = &P.im; bo.seg = &sg; bo.coef = coef; bo.b0 = b;
      const uint64_t stop = i + 1 < sg.nsub ? hj_pos(S[g + si + 1]) : (uint64_t)sg.end*8;
      hj_mem_src src; src.s = P.clean.data();
      uint8_t dz[64];
      for (int q = 0; q < 64; q++) dz[q] = (uint8_t)DEZZ[q];
      const int err = b < total ? hj_write_decode(src, P.im, &P.tabs, dz, start, stop, total - b,
       dc[0], dc[1], dc[2], bo) : 0;
      if (err) return 5;
      b += R[g].nblocks;
    }
    if (b < total) return 6;
  }
  if (rounds_out) *rounds_out = rounds;
  if (nsub_out) *nsub_out = (int)nsub;
  if (runs_out) *runs_out = runs;
  return 0;
}

// ---- hypothesis pass (round 5 study): a guess of every subsequence's start state from runs that start at the
// subsequence's first byte in every MCU-slot phase, linked by equality of their states at a checkpoint `margin`
// bytes into the next subsequence.  Reports how good the guess is: wrong start states, breaks of the chain, and
// the Jacobi rounds the synchronisation needs from the guess (against the rounds from the plain guess).
extern "C" __attribute__((visibility("default")))
int huff_emul_hypotheses(const unsigned char *jpeg, int size, int margin, int *out /* [9] */) {
  hj_prepared P;
  P.sub_log2 = g_sub_log2;
  if (hj_prepare_image(jpeg, size, &P) != EXIT_SUCCESS) return 1;
  const uint32_t nsub = P.im.nsub;
  const int H = P.im.nslots;
  hj_mem_src src; src.s = P.clean.data();
  std::vector<uint64_t> truth(nsub + P.segs.size()), guess(nsub + P.segs.size());
  std::vector<uint32_t> sub_seg(nsub);
  long wrong = 0, breaks = 0, voted = 0, longest = 0, avoidable = 0;
  for (size_t si = 0; si < P.segs.size(); si++) {
    const hj_segment &sg = P.segs[si];
    // truth: in order
    uint64_t st = hj_pack((uint64_t)sg.start*8, 0, 0);
    for (uint32_t i = 0; i < sg.nsub; i++) {
      sub_seg[sg.sub0 + i] = (uint32_t)si;
      truth[sg.sub0 + si + i] = st;
      uint32_t stop_byte = sg.start + ((i + 1) << g_sub_log2);
      if (stop_byte > sg.end) stop_byte = sg.end;
      st = hj_sync_decode(src, P.im, &P.tabs, st, (uint64_t)stop_byte*8, i + 1 >= sg.nsub).end_state;
    }
    truth[sg.sub0 + si + sg.nsub] = 0;
    // hypotheses
    std::vector<uint64_t> Y((size_t)sg.nsub*H), A((size_t)sg.nsub*H), X((size_t)sg.nsub*H);
    for (uint32_t i = 0; i < sg.nsub; i++) {
      const uint32_t b0 = sg.start + (i << g_sub_log2);
      uint32_t b1 = sg.start + ((i + 1) << g_sub_log2);
      if (b1 > sg.end) b1 = sg.end;
      uint32_t by = b0 + (uint32_t)margin < b1 ? b0 + (uint32_t)margin : b1;
      uint32_t bx = b1 + (uint32_t)margin < sg.end ? b1 + (uint32_t)margin : sg.end;
      for (int h = 0; h < H; h++) {
        uint64_t s0 = hj_pack((uint64_t)b0*8, i == 0 ? 0 : h, 0);
        const uint64_t y = hj_sync_decode(src, P.im, &P.tabs, s0, (uint64_t)by*8, false).end_state;
        const uint64_t a = hj_pos(y) >= (uint64_t)b1*8 ? y : hj_sync_decode(src, P.im, &P.tabs, y, (uint64_t)b1*8, false).end_state;
        const uint64_t x = hj_pos(a) >= (uint64_t)bx*8 ? a : hj_sync_decode(src, P.im, &P.tabs, a, (uint64_t)bx*8, false).end_state;
        Y[(size_t)i*H + h] = y; A[(size_t)i*H + h] = a; X[(size_t)i*H + h] = x;
      }
    }
    // link
    int cur = 0;
    long run_wrong = 0;
    guess[sg.sub0 + si] = truth[sg.sub0 + si];
    for (uint32_t i = 0; i + 1 < sg.nsub; i++) {
      const uint64_t a = A[(size_t)i*H + cur];
      guess[sg.sub0 + si + i + 1] = a;
      if (a != truth[sg.sub0 + si + i + 1]) {
        wrong++; run_wrong++; if (run_wrong > longest) longest = run_wrong;
        bool existed = false;
        for (int h = 0; h < H; h++) existed = existed || A[(size_t)i*H + h] == truth[sg.sub0 + si + i + 1];
        if (existed) avoidable++;
      }
      else run_wrong = 0;
      const uint64_t x = X[(size_t)i*H + cur];
      int next = -1;
      for (int h = 0; h < H && next < 0; h++) if (Y[(size_t)(i + 1)*H + h] == x) next = h;
      if (next < 0) {
        breaks++;
        // a hypothesis of the next subsequence that is on the true parse by ITS checkpoint links on (its state there is
        // some hypothesis's of the subsequence after it); garbage rarely does.  Among those that link on, the most
        // common state wins.
        int best = -1, bestn = 0;
        if (i + 2 < sg.nsub) {
          for (int h = 0; h < H; h++) {
            bool links = false;
            for (int g = 0; g < H && !links; g++) links = X[(size_t)(i + 1)*H + h] == Y[(size_t)(i + 2)*H + g];
            if (!links) continue;
            int n = 0;
            for (int g = 0; g < H; g++) n += X[(size_t)(i + 1)*H + g] == X[(size_t)(i + 1)*H + h];
            if (n > bestn) { bestn = n; best = h; }
          }
        }
        if (best >= 0) { next = best; voted++; }
        else {
          // the phase of the block the next subsequence's first byte lies in
          int ph = hj_slot(a);
          if (hj_k(a) == 0 && hj_pos(a) > (uint64_t)(sg.start + ((i + 1) << g_sub_log2))*8) ph = (ph + H - 1) % H;
          next = ph;
        }
      }
      cur = next;
    }
    guess[sg.sub0 + si + sg.nsub] = 0;
  }
  // Jacobi rounds from the guess and from the plain start
  auto rounds_from = [&](std::vector<uint64_t> S, long long *runs) {
    std::vector<uint64_t> last_in(nsub, ~0ull);
    int rounds = 0;
    *runs = 0;
    for (;;) {
      bool ran = false;
      std::vector<uint64_t> snap = S;
      for (uint32_t g = 0; g < nsub; g++) {
        const uint32_t si = sub_seg[g];
        const hj_segment &sg = P.segs[si];
        const uint32_t i = g - sg.sub0;
        const uint64_t start = snap[g + si];
        if (start == last_in[g]) continue;
        uint32_t stop_byte = sg.start + ((i + 1) << g_sub_log2);
        if (stop_byte > sg.end) stop_byte = sg.end;
        const hj_run r = hj_sync_decode(src, P.im, &P.tabs, start, (uint64_t)stop_byte*8, i + 1 >= sg.nsub);
        last_in[g] = start;
        if (i + 1 < sg.nsub) S[g + si + 1] = r.end_state;
        ran = true;
        (*runs)++;
      }
      if (!ran) break;
      rounds++;
      if (rounds > (int)nsub + 4) break;
    }
    return rounds;
  };
  std::vector<uint64_t> plain(nsub + P.segs.size());
  for (size_t si = 0; si < P.segs.size(); si++) {
    const hj_segment &sg = P.segs[si];
    for (uint32_t i = 0; i < sg.nsub; i++) plain[sg.sub0 + si + i] = hj_pack((uint64_t)(sg.start + (i << g_sub_log2))*8, 0, 0);
  }
  long long runs_g = 0, runs_p = 0;
  const int rg = rounds_from(guess, &runs_g), rp = rounds_from(plain, &runs_p);
  out[0] = (int)nsub; out[1] = (int)wrong; out[2] = (int)breaks; out[3] = (int)voted; out[4] = (int)longest;
  out[5] = rg; out[6] = rp; out[7] = (int)(runs_g*100/(nsub ? nsub : 1)); out[8] = (int)avoidable;
  return 0;
}
