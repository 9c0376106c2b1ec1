// huff_prepare.cpp — see huff_prepare.h.  Replaces, for the GPU entropy stage, the
// table set-up of the reference's host decoder (DHT codeword generation
// src/xjpeg.c:293-336, MCU structure xjpeg_mcu_init 431-446, restart bookkeeping
// 593-629): tables are built once on the host, the scan itself is decoded on the GPU.
#include <stdlib.h>
#include <string.h>
#include <new>
#include <memory>
#include "huff_prepare.h"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

// Copy src[0..n) to dst until the first 0xFF byte (not copied); returns the number of bytes
// before it (n if there is none).  May write up to 31 bytes of junk past the returned count
// (the callers' buffers have that room and overwrite it next).  The scan is mostly FF-free
// runs of a few hundred bytes: a fused compare+copy beats memchr + memcpy per run.
#if defined(__x86_64__)
__attribute__((target("avx2")))
static uint32_t copy_until_ff_avx2(unsigned char *dst, const unsigned char *src, uint32_t n) {
  uint32_t i = 0;
  const __m256i ff = _mm256_set1_epi8((char)0xFF);
  while (i + 32 <= n) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)(src + i));
    const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, ff));
    _mm256_storeu_si256((__m256i *)(dst + i), v);
    if (m) return i + (uint32_t)__builtin_ctz(m);
    i += 32;
  }
  while (i < n && src[i] != 0xFF) { dst[i] = src[i]; i++; }
  return i;
}
#endif
static uint32_t copy_until_ff_plain(unsigned char *dst, const unsigned char *src, uint32_t n) {
  const unsigned char *ff = n ? (const unsigned char *)memchr(src, 0xFF, n) : NULL;
  const uint32_t k = ff ? (uint32_t)(ff - src) : n;
  memcpy(dst, src, k);
  return k;
}
static uint32_t copy_until_ff(unsigned char *dst, const unsigned char *src, uint32_t n) {
#if defined(__x86_64__)
  static const int have_avx2 = __builtin_cpu_supports("avx2");
  if (have_avx2) return copy_until_ff_avx2(dst, src, n);
#endif
  return copy_until_ff_plain(dst, src, n);
}

// Build the two-level lookup of one table into l1[512] (+ level-2 blocks taken from
// *l2_used).  Returns 0, 1 = malformed DHT, 2 = out of level-2 blocks.
static int build_table(hj_tables *T, uint16_t *l1, bool is_dc, int *l2_used, const unsigned char bits[16],
 const unsigned char *vals) {
  unsigned code = 0;
  int k = 0;
  // what a bit pattern that is no code decodes as (see hj_tables)
  const uint16_t nocode = HJ_ENTRY(17, 0, is_dc ? 0 : 63);
  const int l2_first = *l2_used;
  memset(l1, 0, sizeof(uint16_t) << HJ_FAST_BITS);
  for (int len = 1; len <= 16; len++) {
    for (int i = 0; i < bits[len - 1]; i++, k++, code++) {
      if (k >= 256 || code >= (1u << len)) return 1;
      const int sym = vals[k];
      const uint16_t entry = HJ_ENTRY(len, sym & 15, is_dc ? 0 : (sym == 0 ? 63 : sym >> 4));
      if (len <= HJ_FAST_BITS) {
        const unsigned c = code << (HJ_FAST_BITS - len);
        for (unsigned j = 0; j < (1u << (HJ_FAST_BITS - len)); j++) l1[c + j] = entry;
      }
      else {
        const unsigned prefix = code >> (len - HJ_FAST_BITS);
        if (!l1[prefix] || !HJ_IS_ESCAPE(l1[prefix])) {
          if (l1[prefix]) return 1;                       // prefix of a shorter code: not prefix-free
          if (*l2_used >= HJ_L2_BLOCKS) return 2;
          memset(T->l2 + 128*(*l2_used), 0, 256);
          l1[prefix] = HJ_ESCAPE((*l2_used)++);
        }
        uint16_t *blk = T->l2 + 128*HJ_ESCAPE_BLOCK(l1[prefix]);
        const unsigned c = (code & ((1u << (len - HJ_FAST_BITS)) - 1u)) << (16 - len);
        for (unsigned j = 0; j < (1u << (16 - len)); j++) blk[c + j] = entry;
      }
    }
    code <<= 1;
  }
  for (int j = 0; j < (1 << HJ_FAST_BITS); j++) if (!l1[j]) l1[j] = nocode;
  for (int j = 128*l2_first; j < 128*(*l2_used); j++) if (!T->l2[j]) T->l2[j] = nocode;
  return 0;
}

// The pack of every 9-bit pattern (hj_tables): as many whole AC symbols — code and magnitude
// bits — as the pattern holds, at most three, an EOB only as the last; none if fewer than two.
static void build_packs(uint32_t *ac, const uint16_t *l1) {
  static const bool off = jga_tune("JGA_HUFF_PACKS") && atoi(jga_tune("JGA_HUFF_PACKS")) == 0;   // (A/B knob)
  for (unsigned v = 0; v < (1u << HJ_FAST_BITS); v++) {
    int rem = HJ_FAST_BITS, n = 0, bits = 0, adv = 0, prefix = 0;
    unsigned val = v;
    while (n < 3 && rem > 0) {
      const uint16_t e = l1[(val << (HJ_FAST_BITS - rem)) & ((1u << HJ_FAST_BITS) - 1u)];   // zero padded: decides
      if (HJ_IS_ESCAPE(e)) break;                              // nothing unless the code fits the bits left
      const int tot = HJ_E_TOT(e), a = HJ_E_ADV(e);
      if (HJ_E_LEN(e) > 16 || tot > rem || adv > 31) break;    // (no code / too long / prefix field full)
      prefix = adv;
      adv += a;
      bits += tot;
      rem -= tot;
      val &= (1u << rem) - 1u;
      n++;
      if (a == 64) break;                                      // EOB ends the block: the last symbol of a pack
    }
    ac[v] = (uint32_t)l1[v] | (n >= 2 && !off ? HJ_PACK(bits, adv, prefix) << 16 : 0u);
  }
}

// The wide AC table of one slot (hj_wide_ac): for every 12-bit pattern the entry of its first symbol (as the 9-bit
// table holds it) and the pack of the whole symbols the pattern holds — up to three, an EOB only as the last, none
// if fewer than two.  Symbols are looked up in the 9-bit table `ac9` (codes of more than nine bits end a pack).
static void build_wide(uint32_t *wide, const uint32_t *ac9) {
  for (unsigned v = 0; v < (1u << HJ_WIDE_BITS); v++) {
    int rem = HJ_WIDE_BITS, n = 0, bits = 0, adv = 0, prefix = 0;
    unsigned val = v;
    while (n < 3 && rem > 0) {
      const unsigned idx = rem >= HJ_FAST_BITS ? (val >> (rem - HJ_FAST_BITS)) : (val << (HJ_FAST_BITS - rem));   // zero padded: decides
      const uint16_t e = (uint16_t)ac9[idx & ((1u << HJ_FAST_BITS) - 1u)];                                        // nothing unless the code fits
      if (HJ_IS_ESCAPE(e)) break;
      const int tot = HJ_E_TOT(e), a = HJ_E_ADV(e);
      if (HJ_E_LEN(e) > 16 || tot > rem || adv > 31) break;
      prefix = adv;
      adv += a;
      bits += tot;
      rem -= tot;
      val &= (1u << rem) - 1u;
      n++;
      if (a == 64) break;
    }
    wide[v] = (ac9[v >> (HJ_WIDE_BITS - HJ_FAST_BITS)] & 0xffffu) | (n >= 2 ? HJ_PACK(bits, adv, prefix) << 16 : 0u);
  }
}

hj_prepared::~hj_prepared() { free(desc); }

int hj_prepare_head(const unsigned char *jpeg, int size, hj_prepared *out) {
  free(out->desc);
  out->desc = (jga_scan_desc *)malloc(sizeof(jga_scan_desc));
  jga_scan_desc *d = out->desc;
  if (!d) return jga_fail("Out of memory");
  if (jga_scan_describe(jpeg, size, d) != EXIT_SUCCESS
   || jga_geom_from_header(&out->geom, &d->header) != EXIT_SUCCESS) {
    return EXIT_FAILURE;
  }
  const jga_geom &g = out->geom;
  hj_image &im = out->im;
  memset(&im, 0, sizeof(im));
  // The device stage keeps a bit position within one scan and a byte offset within one
  // image's coefficient planes in 32 bits each (hj_reader::p, hj_block_out::offset): frames
  // beyond that (a 40000 x 40000 4:2:0 frame has 4.8 GB of planes) are valid JPEG but not
  // for it — the host entropy stage takes them, like tables outside the lookup format.
  if (g.coef_shorts*2 >= (1ll << 32) || (long long)(size - d->scan_off)*8 >= (1ll << 32)) {
    jga_fail("Frame too large for the GPU entropy stage (%lld bytes of planes, %d of scan)",
     g.coef_shorts*2, size - d->scan_off);
    return HJ_PREPARE_IRREGULAR;
  }
  int slot = 0, l2_used = 0;
  int nslot[2] = {0, 0}, slot_id[2][2] = {{-1, -1}, {-1, -1}};       // [DC / AC]: DHT ids in the two table slots
  // A stream of files from one encoder brings the same DHT segments file after file (every file of the bench, of a
  // camera, of a transcoding farm): the lookup tables of the file before — 10 KB that took ~20 us to build — are
  // kept per thread and taken over when the tables this frame SELECTS are byte for byte the ones they were built
  // from.  What a frame selects, in component order: (DC id, AC id, 16 counts + 256 values of each).
  struct table_memo {
    bool valid = false;
    int nplanes = 0;
    unsigned char sel[3][2];
    unsigned char bits[3][2][16], vals[3][2][256];
    hj_tables tabs;
    uint8_t comp_tbl[3];
  };
  static thread_local std::unique_ptr<table_memo> memo;     // (freed when the thread ends)
  bool memo_hit = memo && memo->valid && memo->nplanes == g.nplanes;
  for (int c = 0; c < g.nplanes && memo_hit; c++) {
    for (int w = 0; w < 2 && memo_hit; w++) {
      const int id = w ? 4 + d->ta[c] : d->td[c];
      memo_hit = memcmp(memo->bits[c][w], d->dht_bits[id], 16) == 0 && memcmp(memo->vals[c][w], d->dht_vals[id], 256) == 0;
    }
  }
  if (memo_hit) memcpy(&out->tabs, &memo->tabs, sizeof(out->tabs));
  else memset(&out->tabs, 0, sizeof(out->tabs));
  for (int c = 0; c < g.nplanes; c++) {
    const jpeg_component &cp = d->header.comp[c];
    im.comp_hs[c] = (uint8_t)cp.hsamp;
    im.comp_vs[c] = (uint8_t)cp.vsamp;
    im.comp_xdec[c] = (uint8_t)g.plane[c].xdec;
    im.comp_coef_off[c] = g.plane[c].coef_off;
    for (int sy = 0; sy < cp.vsamp; sy++) {
      for (int sx = 0; sx < cp.hsamp; sx++) {
        if (slot >= HJ_MAX_SLOTS) {
          // more blocks per MCU than T.81 B.2.3 allows (ten); the reference decodes such files
          // all the same (4x4 luma: src/xjpeg.c:384-391), so does the host entropy stage
          jga_fail("MCU of more than %d blocks: not for the GPU entropy stage", HJ_MAX_SLOTS);
          return HJ_PREPARE_IRREGULAR;
        }
        im.slot_comp[slot] = (uint8_t)c;
        im.slot_sbx[slot] = (uint8_t)sx;
        im.slot_sby[slot] = (uint8_t)sy;
        slot++;
      }
    }
    if (memo_hit) im.comp_tbl[c] = memo->comp_tbl[c];
    else {
      // the device format holds two DC and two AC tables (luma / chroma): components that
      // select the same DHT share a slot
      int rc = 0;
      im.comp_tbl[c] = 0;
      for (int w = 0; w < 2 && !rc; w++) {
        const int id = w ? 4 + d->ta[c] : d->td[c];
        int slot_of = -1;
        // (by CONTENT, not by id: an encoder that writes one DHT per component with the same
        // bits + values under three ids still needs two slots, not three)
        for (int q = 0; q < nslot[w]; q++) {
          const int other = slot_id[w][q];
          if (other == id || (memcmp(d->dht_bits[other], d->dht_bits[id], sizeof(d->dht_bits[id])) == 0
           && memcmp(d->dht_vals[other], d->dht_vals[id], sizeof(d->dht_vals[id])) == 0)) {
            slot_of = q;
          }
        }
        if (slot_of < 0) {
          if (nslot[w] >= 2) { rc = 2; break; }
          slot_of = nslot[w]++;
          slot_id[w][slot_of] = id;
          uint16_t l1[1 << HJ_FAST_BITS];
          rc = build_table(&out->tabs, w ? l1 : out->tabs.dc[slot_of], w == 0, &l2_used, d->dht_bits[id], d->dht_vals[id]);
          if (!rc && w) build_packs(out->tabs.ac[slot_of], l1);
        }
        im.comp_tbl[c] |= (uint8_t)(slot_of << w);
      }
      if (rc) {
        jga_fail(rc == 2 ? "Huffman table too irregular for the GPU entropy stage"
         : "Error invalid DHT.");
        return rc == 2 ? HJ_PREPARE_IRREGULAR : EXIT_FAILURE;
      }
    }
    memcpy(out->qtab + 64*c, cp.quant->tbl, 64*sizeof(unsigned short));
  }
  for (int c = g.nplanes; c < 3; c++) memset(out->qtab + 64*c, 0, 64*sizeof(unsigned short));
  if (!memo_hit) {
    if (!memo) memo.reset(new (std::nothrow) table_memo());    // (one per thread that prepares)
    if (memo) {
      memo->valid = true;
      memo->nplanes = g.nplanes;
      for (int c = 0; c < g.nplanes; c++) {
        for (int w = 0; w < 2; w++) {
          const int id = w ? 4 + d->ta[c] : d->td[c];
          memcpy(memo->bits[c][w], d->dht_bits[id], 16);
          memcpy(memo->vals[c][w], d->dht_vals[id], 256);
        }
        memo->comp_tbl[c] = im.comp_tbl[c];
      }
      memcpy(&memo->tabs, &out->tabs, sizeof(out->tabs));
    }
  }
  im.nslots = slot;
  im.nhmb = g.nhmb;
  im.w0_blocks = g.w0/8;
  out->avail = (uint32_t)(size - d->scan_off);
  return EXIT_SUCCESS;
}

void hj_prepare_wide(hj_prepared *out) {
  // (per thread, beside the 9-bit AC tables they were built from: a stream of lone frames from one encoder builds
  // them once — 8 192 entries, ~40 us)
  struct wide_memo { uint32_t ac9[2][1 << HJ_FAST_BITS]; std::vector<uint32_t> wide; };
  static thread_local std::unique_ptr<wide_memo> memo;
  if (memo && memcmp(memo->ac9, out->tabs.ac, sizeof(memo->ac9)) == 0) { out->wide = memo->wide; return; }
  out->wide.resize(2u << HJ_WIDE_BITS);
  build_wide(out->wide.data(), out->tabs.ac[0]);
  build_wide(out->wide.data() + (1u << HJ_WIDE_BITS), out->tabs.ac[1]);
  if (!memo) memo.reset(new (std::nothrow) wide_memo());
  if (memo) {
    memcpy(memo->ac9, out->tabs.ac, sizeof(memo->ac9));
    memo->wide = out->wide;
  }
}

int hj_prepare_scan(const unsigned char *jpeg, int size, hj_prepared *out, unsigned char *dst) {
  const jga_scan_desc *d = out->desc;
  if (!d) return jga_fail("huff: hj_prepare_head must come first");
  const jga_geom &g = out->geom;
  hj_image &im = out->im;
  // entropy-coded bytes: split at RSTn markers, stop at the first other marker; copy
  // them into a clean stream (stuffed zeros and markers dropped) as we go
  const unsigned char *scan = jpeg + d->scan_off;
  const uint32_t avail = (uint32_t)(size - d->scan_off);
  const uint32_t total_mcus = (uint32_t)g.nhmb*(uint32_t)g.nvmb;
  const uint32_t ri = (uint32_t)d->header.restart_interval;
  out->segs.clear();
  uint32_t pos = 0, mcu0 = 0, nsub = 0, clean_start = 0, w = 0;
  int expect = 0, rc = EXIT_SUCCESS;
  bool done = false;
  while (!done) {
    const uint32_t run = pos < avail ? copy_until_ff(dst + w, scan + pos, avail - pos) : 0;
    const uint32_t at = pos + run;
    const bool ff = at < avail;                                 // scan[at] == 0xFF
    const int marker = (ff && at + 1 < avail) ? scan[at + 1] : 0xD9;   // running off the end == EOI
    w += run;
    if (ff && marker == 0x00) { dst[w++] = 0xFF; pos = at + 2; continue; }   // stuffed zero
    if (ff && marker == 0xFF) { pos = at + 1; continue; }               // fill byte
    // a real marker (or the end of the buffer) closes the current segment
    hj_segment s;
    s.start = clean_start;
    s.end = w;
    s.sub0 = nsub;
    s.nsub = (s.end - s.start + (1u << out->sub_log2) - 1) >> out->sub_log2;
    if (s.nsub == 0) s.nsub = 1;
    s.mcu0 = mcu0;
    s.nmcu = ri ? (total_mcus - mcu0 < ri ? total_mcus - mcu0 : ri) : total_mcus;
    out->segs.push_back(s);
    nsub += s.nsub;
    mcu0 += s.nmcu;
    out->raw_len = at;
    if (marker >= 0xD0 && marker <= 0xD7 && ri && mcu0 < total_mcus) {
      if (marker != 0xD0 + (expect & 7)) { rc = jga_fail("Error invalid RST counter in marker."); break; }
      expect++;
      pos = at + 2;
      clean_start = s.end;
    }
    else done = true;
  }
  out->scan_len = w;
  memset(dst + w, 0xFF, 16);
  free(out->desc);
  out->desc = nullptr;
  if (rc != EXIT_SUCCESS) return rc;
  if (mcu0 != total_mcus) return jga_fail("Error, entropy data ended early.");
  im.nsub = nsub;
  im.nseg = (uint32_t)out->segs.size();
  im.scan_len = out->scan_len;
  return EXIT_SUCCESS;
}

int hj_prepare_image(const unsigned char *jpeg, int size, hj_prepared *out) {
  if (hj_prepare_head(jpeg, size, out) != EXIT_SUCCESS) return EXIT_FAILURE;
  out->clean.assign((size_t)out->avail + 16, 0);
  if (hj_prepare_scan(jpeg, size, out, out->clean.data()) != EXIT_SUCCESS) return EXIT_FAILURE;
  out->clean.resize((size_t)out->scan_len + 16);
  return EXIT_SUCCESS;
}

int hj_walk_unsettled(const hj_image &im, const hj_segment *segs, const hj_tables *tabs,
 const unsigned char *clean, uint64_t *S, const uint64_t *last_in, int sub_log2) {
  hj_mem_src src;
  src.s = clean;
  int walked = 0;
  for (uint32_t si = 0; si < im.nseg; si++) {
    const hj_segment &sg = segs[si];
    uint64_t *Ss = S + sg.sub0 + si;                             // nsub + 1 entries
    const uint64_t *Ls = last_in + sg.sub0;
    for (uint32_t k = 0; k < sg.nsub; k++) {
      if (Ss[k] == Ls[k]) continue;                              // ran from its current state
      for (;;) {                                                 // walk on from lane k
        uint32_t stop = sg.start + ((k + 1) << sub_log2);
        if (stop > sg.end) stop = sg.end;
        const hj_run r = hj_sync_decode(src, im, tabs, Ss[k], (uint64_t)stop*8, k + 1 >= sg.nsub);
        walked++;
        if (k + 1 >= sg.nsub) break;
        k++;
        if (Ss[k] == r.end_state && Ls[k] == r.end_state) break; // that lane ran from here
        Ss[k] = r.end_state;
        if (Ls[k] == r.end_state) break;                         // (its output stands as well)
      }
    }
  }
  return walked;
}
