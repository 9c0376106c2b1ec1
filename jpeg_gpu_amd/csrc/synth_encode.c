/* synth_encode.c — deterministic baseline-JPEG writer for synthetic inputs.
 *
 * The reference ships no JPEG fixtures and the GPU box has no
 * network, so tests and bench.py synthesise their inputs with this encoder:
 * SOF0 baseline, 8-bit, 1 or 3 components, sampling factors {1,2,4}, ITU-T T.81
 * Annex K quantisation tables scaled the IJG way, Annex K Huffman tables,
 * optional DRI.  It is an input generator, not part of the decode path and not
 * the oracle; it lives in its own library (libjga_synth.so).
 *
 * Three entry points:
 *   jgs_encode_synthetic : SURVEY.md §8(d) recipe — per channel
 *       127 + 80 sin(x/(37+11c)) cos(y/(53+7c)) + noise(sigma 12), seed-driven
 *   jgs_encode_pixels    : caller's grey / RGB pixels
 *   jgs_encode_levels    : caller's QUANTISED coefficient levels, given in the
 *       packed coefficient-plane layout the decoder produces (SURVEY.md
 *       Appendix B) — decode(encode(levels)) == levels is the round-trip
 *       property the entropy tests use.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JGS_API __attribute__((visibility("default")))

/* ---- ITU-T T.81 Annex K tables (standard-defined) ---------------------- */
static const unsigned char K1_LUMA_Q[64] = {
  16, 11, 10, 16, 24, 40, 51, 61,   12, 12, 14, 19, 26, 58, 60, 55,
  14, 13, 16, 24, 40, 57, 69, 56,   14, 17, 22, 29, 51, 87, 80, 62,
  18, 22, 37, 56, 68,109,103, 77,   24, 35, 55, 64, 81,104,113, 92,
  49, 64, 78, 87,103,121,120,101,   72, 92, 95, 98,112,100,103, 99
};
static const unsigned char K2_CHROMA_Q[64] = {
  17, 18, 24, 47, 99, 99, 99, 99,   18, 21, 26, 66, 99, 99, 99, 99,
  24, 26, 56, 99, 99, 99, 99, 99,   47, 66, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99,   99, 99, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99,   99, 99, 99, 99, 99, 99, 99, 99
};
static const unsigned char K3_DC_LUMA_BITS[16] = {0,1,5,1,1,1,1,1,1,0,0,0,0,0,0,0};
static const unsigned char K4_DC_CHROMA_BITS[16] = {0,3,1,1,1,1,1,1,1,1,1,0,0,0,0,0};
static const unsigned char K_DC_VALS[12] = {0,1,2,3,4,5,6,7,8,9,10,11};
static const unsigned char K5_AC_LUMA_BITS[16] = {0,2,1,3,3,2,4,3,5,5,4,4,0,0,1,0x7d};
static const unsigned char K5_AC_LUMA_VALS[162] = {
  0x01,0x02,0x03,0x00,0x04,0x11,0x05,0x12,0x21,0x31,0x41,0x06,0x13,0x51,0x61,0x07,
  0x22,0x71,0x14,0x32,0x81,0x91,0xa1,0x08,0x23,0x42,0xb1,0xc1,0x15,0x52,0xd1,0xf0,
  0x24,0x33,0x62,0x72,0x82,0x09,0x0a,0x16,0x17,0x18,0x19,0x1a,0x25,0x26,0x27,0x28,
  0x29,0x2a,0x34,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,
  0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,
  0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x83,0x84,0x85,0x86,0x87,0x88,0x89,
  0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,
  0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,
  0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe1,0xe2,
  0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf1,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,
  0xf9,0xfa
};
static const unsigned char K6_AC_CHROMA_BITS[16] = {0,2,1,2,4,4,3,4,7,5,4,4,0,1,2,0x77};
static const unsigned char K6_AC_CHROMA_VALS[162] = {
  0x00,0x01,0x02,0x03,0x11,0x04,0x05,0x21,0x31,0x06,0x12,0x41,0x51,0x07,0x61,0x71,
  0x13,0x22,0x32,0x81,0x08,0x14,0x42,0x91,0xa1,0xb1,0xc1,0x09,0x23,0x33,0x52,0xf0,
  0x15,0x62,0x72,0xd1,0x0a,0x16,0x24,0x34,0xe1,0x25,0xf1,0x17,0x18,0x19,0x1a,0x26,
  0x27,0x28,0x29,0x2a,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,
  0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,
  0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x82,0x83,0x84,0x85,0x86,0x87,
  0x88,0x89,0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,
  0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,
  0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,
  0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,
  0xf9,0xfa
};

/* zig-zag scan order (T.81 Figure A.6), generated */
static int ZZ[64];
static void init_zz(void) {
  int k = 0, s, i;
  if (ZZ[63] == 63) return;
  for (s = 0; s < 15; s++) {
    for (i = 0; i <= s; i++) {
      int r = (s & 1) ? i : s - i, c = s - r;
      if (r < 8 && c < 8) ZZ[k++] = r*8 + c;
    }
  }
}

/* ---- byte / bit writer -------------------------------------------------- */
typedef struct writer {
  unsigned char *out;
  long cap, n;
  uint64_t acc;
  int nacc;
  int overflow;
} writer;

static void put8(writer *w, int b) {
  if (w->n < w->cap) w->out[w->n] = (unsigned char)b;
  else w->overflow = 1;
  w->n++;
}
static void put16(writer *w, int v) { put8(w, v >> 8); put8(w, v & 0xFF); }
static void put_bits(writer *w, unsigned code, int len) {
  if (!len) return;
  w->acc = (w->acc << len) | (code & ((1u << len) - 1));
  w->nacc += len;
  while (w->nacc >= 8) {
    int b = (int)((w->acc >> (w->nacc - 8)) & 0xFF);
    put8(w, b);
    if (b == 0xFF) put8(w, 0);          /* byte stuffing, T.81 F.1.2.3 */
    w->nacc -= 8;
  }
}
static void flush_bits(writer *w) {      /* pad with 1 bits to a byte boundary */
  if (w->nacc) put_bits(w, (1u << (8 - w->nacc)) - 1, 8 - w->nacc);
  w->acc = 0;
  w->nacc = 0;
}

typedef struct hufftab {
  unsigned short code[256];
  unsigned char len[256];
} hufftab;

static void build_huff(hufftab *h, const unsigned char bits[16],
 const unsigned char *vals) {
  unsigned code = 0;
  int k = 0, i, j;
  memset(h, 0, sizeof(*h));
  for (i = 0; i < 16; i++) {
    for (j = 0; j < bits[i]; j++) {
      h->code[vals[k]] = (unsigned short)code;
      h->len[vals[k]] = (unsigned char)(i + 1);
      k++;
      code++;
    }
    code <<= 1;
  }
}

static void put_dht(writer *w, int tc_th, const unsigned char bits[16],
 const unsigned char *vals) {
  int n = 0, i;
  for (i = 0; i < 16; i++) n += bits[i];
  put16(w, 0xFFC4);
  put16(w, 2 + 1 + 16 + n);
  put8(w, tc_th);
  for (i = 0; i < 16; i++) put8(w, bits[i]);
  for (i = 0; i < n; i++) put8(w, vals[i]);
}

static int bit_category(int v) {
  int n = 0;
  if (v < 0) v = -v;
  while (v) { n++; v >>= 1; }
  return n;
}

/* Encode one block of quantised levels (natural order).  Returns 0, or 1 if a
 * value cannot be coded with the baseline tables. */
static int put_block(writer *w, const short lv[64], int *dc_pred,
 const hufftab *dc, const hufftab *ac) {
  int diff = lv[0] - *dc_pred, cat, k, run = 0;
  /* DC differences are coded modulo 2^16 like the decoder's int16 predictor */
  diff = (short)diff;
  *dc_pred = lv[0];
  cat = bit_category(diff);
  if (cat > 11) return 1;
  put_bits(w, dc->code[cat], dc->len[cat]);
  put_bits(w, (unsigned)(diff < 0 ? diff - 1 : diff), cat);
  for (k = 1; k < 64; k++) {
    int v = lv[ZZ[k]];
    if (v == 0) { run++; continue; }
    while (run > 15) { put_bits(w, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
    cat = bit_category(v);
    if (cat > 10) return 1;
    put_bits(w, ac->code[(run << 4) | cat], ac->len[(run << 4) | cat]);
    put_bits(w, (unsigned)(v < 0 ? v - 1 : v), cat);
    run = 0;
  }
  if (run) put_bits(w, ac->code[0], ac->len[0]);
  return 0;
}

/* ---- frame description -------------------------------------------------- */
typedef struct frame {
  int width, height, ncomps;
  int hs[3], vs[3];
  int hmax, vmax, nhmb, nvmb;
  int hblocks[3], vblocks[3], xdec[3], ydec[3], cstride[3];
  long long coef_off[3], coef_shorts;
} frame;

static int ilog(unsigned v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

/* Chroma sampling factors of the frames made from here on (Cb h, v, Cr h, v): 1,1,1,1 unless a
 * test asks for a file whose chroma planes are decimated differently, or whose luma is not the
 * finest plane.  Process-wide and not thread-safe: set, encode, reset (synth.py does). */
static int g_chroma[4] = {1, 1, 1, 1};
JGS_API int jgs_set_chroma_factors(int cbh, int cbv, int crh, int crv) {
  const int v[4] = {cbh, cbv, crh, crv};
  int i;
  for (i = 0; i < 4; i++) if (v[i] != 1 && v[i] != 2 && v[i] != 4) return 1;
  memcpy(g_chroma, v, sizeof(v));
  return 0;
}

static int frame_init(frame *f, int width, int height, int ncomps, int hs,
 int vs) {
  int i;
  long long off = 0;
  memset(f, 0, sizeof(*f));
  if (width < 1 || height < 1 || width > 65535 || height > 65535) return 1;
  if (ncomps != 1 && ncomps != 3) return 1;
  if ((hs != 1 && hs != 2 && hs != 4) || (vs != 1 && vs != 2 && vs != 4)) return 1;
  f->width = width; f->height = height; f->ncomps = ncomps;
  f->hs[0] = ncomps == 1 ? 1 : hs; f->vs[0] = ncomps == 1 ? 1 : vs;
  f->hs[1] = g_chroma[0]; f->vs[1] = g_chroma[1];
  f->hs[2] = g_chroma[2]; f->vs[2] = g_chroma[3];
  f->hmax = f->hs[0]; f->vmax = f->vs[0];
  for (i = 1; i < ncomps; i++) {
    if (f->hs[i] > f->hmax) f->hmax = f->hs[i];
    if (f->vs[i] > f->vmax) f->vmax = f->vs[i];
  }
  f->nhmb = (width + 8*f->hmax - 1)/(8*f->hmax);
  f->nvmb = (height + 8*f->vmax - 1)/(8*f->vmax);
  for (i = 0; i < ncomps; i++) {
    f->hblocks[i] = f->nhmb*f->hs[i];
    f->vblocks[i] = f->nvmb*f->vs[i];
    f->xdec[i] = ilog(f->hmax) - ilog(f->hs[i]);
    f->ydec[i] = ilog(f->vmax) - ilog(f->vs[i]);
    f->cstride[i] = (f->vblocks[i] + (1 << f->xdec[i]) - 1) >> f->xdec[i];
    f->coef_off[i] = off;
    off += ((long long)f->hblocks[i] << (f->xdec[i] + 6))*f->cstride[i];
  }
  f->coef_shorts = off;
  return 0;
}

static long long block_off(const frame *f, int p, int bx, int by) {
  long long rs = (long long)f->hblocks[0] << 6;
  return f->coef_off[p] + rs*(by >> f->xdec[p])
   + (rs >> f->xdec[p])*(by & ((1 << f->xdec[p]) - 1)) + ((long long)bx << 6);
}

/* flags */
#define JGS_DQT16      1   /* write 16-bit (Pq=1) quantisation tables */
#define JGS_NO_JFIF    2   /* omit the APP0 segment */
#define JGS_SPLIT_DHT  4   /* one DHT segment per table instead of one for all */
#define JGS_FLAT_AC    8   /* AC tables with all 162 symbols on 10-bit codes: valid, wasteful, and
                            * 81 distinct 9-bit prefixes of long codes (the Annex K tables have 5) */
static const unsigned char FLAT_AC_BITS[16] = {0,0,0,0,0,0,0,0,0,162,0,0,0,0,0,0};
#define JGS_SWAP_AC   16   /* luma coded with the Annex K CHROMA AC table and chroma with the luma one: valid,
                            * regular, and not the tables every other file of a batch brings */
#define AC_LUMA_BITS(flags) (((flags) & JGS_FLAT_AC) ? FLAT_AC_BITS : ((flags) & JGS_SWAP_AC) ? K6_AC_CHROMA_BITS : K5_AC_LUMA_BITS)
#define AC_CHROMA_BITS(flags) (((flags) & JGS_FLAT_AC) ? FLAT_AC_BITS : ((flags) & JGS_SWAP_AC) ? K5_AC_LUMA_BITS : K6_AC_CHROMA_BITS)
#define AC_LUMA_VALS(flags) (((flags) & JGS_SWAP_AC) ? K6_AC_CHROMA_VALS : K5_AC_LUMA_VALS)
#define AC_CHROMA_VALS(flags) (((flags) & JGS_SWAP_AC) ? K5_AC_LUMA_VALS : K6_AC_CHROMA_VALS)

static void put_headers(writer *w, const frame *f,
 const unsigned short q[3][64], int nq, int restart_interval, int flags) {
  int i, k;
  init_zz();
  put16(w, 0xFFD8);
  if (!(flags & JGS_NO_JFIF)) {
    static const unsigned char app0[14] =
     {'J','F','I','F',0, 1,1, 0, 0,1, 0,1, 0,0};
    put16(w, 0xFFE0);
    put16(w, 16);
    for (i = 0; i < 14; i++) put8(w, app0[i]);
  }
  for (i = 0; i < nq; i++) {
    int p16 = (flags & JGS_DQT16) != 0;
    put16(w, 0xFFDB);
    put16(w, 2 + 1 + (p16 ? 128 : 64));
    put8(w, (p16 << 4) | i);
    for (k = 0; k < 64; k++) {
      if (p16) put16(w, q[i][ZZ[k]]);
      else put8(w, q[i][ZZ[k]]);
    }
  }
  put16(w, 0xFFC0);
  put16(w, 8 + 3*f->ncomps);
  put8(w, 8);
  put16(w, f->height);
  put16(w, f->width);
  put8(w, f->ncomps);
  for (i = 0; i < f->ncomps; i++) {
    put8(w, i + 1);
    put8(w, (f->hs[i] << 4) | f->vs[i]);
    put8(w, nq == 1 ? 0 : (i == 0 ? 0 : (nq == 2 ? 1 : i)));
  }
  if (flags & JGS_SPLIT_DHT) {
    put_dht(w, 0x00, K3_DC_LUMA_BITS, K_DC_VALS);
    put_dht(w, 0x10, AC_LUMA_BITS(flags), AC_LUMA_VALS(flags));
    if (f->ncomps == 3) {
      put_dht(w, 0x01, K4_DC_CHROMA_BITS, K_DC_VALS);
      put_dht(w, 0x11, AC_CHROMA_BITS(flags), AC_CHROMA_VALS(flags));
    }
  }
  else {
    int n = 2 + (1 + 16 + 12) + (1 + 16 + 162);
    if (f->ncomps == 3) n += (1 + 16 + 12) + (1 + 16 + 162);
    put16(w, 0xFFC4);
    put16(w, n);
    put8(w, 0x00);
    for (i = 0; i < 16; i++) put8(w, K3_DC_LUMA_BITS[i]);
    for (i = 0; i < 12; i++) put8(w, K_DC_VALS[i]);
    put8(w, 0x10);
    for (i = 0; i < 16; i++) put8(w, AC_LUMA_BITS(flags)[i]);
    for (i = 0; i < 162; i++) put8(w, AC_LUMA_VALS(flags)[i]);
    if (f->ncomps == 3) {
      put8(w, 0x01);
      for (i = 0; i < 16; i++) put8(w, K4_DC_CHROMA_BITS[i]);
      for (i = 0; i < 12; i++) put8(w, K_DC_VALS[i]);
      put8(w, 0x11);
      for (i = 0; i < 16; i++) put8(w, AC_CHROMA_BITS(flags)[i]);
      for (i = 0; i < 162; i++) put8(w, AC_CHROMA_VALS(flags)[i]);
    }
  }
  if (restart_interval) {
    put16(w, 0xFFDD);
    put16(w, 4);
    put16(w, restart_interval);
  }
  put16(w, 0xFFDA);
  put16(w, 6 + 2*f->ncomps);
  put8(w, f->ncomps);
  for (i = 0; i < f->ncomps; i++) {
    put8(w, i + 1);
    put8(w, i == 0 ? 0x00 : 0x11);
  }
  put8(w, 0);
  put8(w, 63);
  put8(w, 0);
}

/* Entropy-code a whole packed coefficient buffer of quantised levels. */
static long encode_levels(const frame *f, const short *levels,
 const unsigned short q[3][64], int nq, int restart_interval, int flags,
 unsigned char *out, long cap) {
  writer w;
  hufftab dcl, acl, dcc, acc;
  int dc_pred[3] = {0, 0, 0};
  int mbx, mby, i, sbx, sby, mcu = 0, rst = 0;
  long total = (long)f->nhmb*f->nvmb;
  memset(&w, 0, sizeof(w));
  w.out = out;
  w.cap = cap;
  build_huff(&dcl, K3_DC_LUMA_BITS, K_DC_VALS);
  build_huff(&acl, AC_LUMA_BITS(flags), AC_LUMA_VALS(flags));
  build_huff(&dcc, K4_DC_CHROMA_BITS, K_DC_VALS);
  build_huff(&acc, AC_CHROMA_BITS(flags), AC_CHROMA_VALS(flags));
  put_headers(&w, f, q, nq, restart_interval, flags);
  for (mby = 0; mby < f->nvmb; mby++) {
    for (mbx = 0; mbx < f->nhmb; mbx++) {
      for (i = 0; i < f->ncomps; i++) {
        for (sby = 0; sby < f->vs[i]; sby++) {
          for (sbx = 0; sbx < f->hs[i]; sbx++) {
            const short *lv = levels
             + block_off(f, i, mbx*f->hs[i] + sbx, mby*f->vs[i] + sby);
            if (put_block(&w, lv, &dc_pred[i], i ? &dcc : &dcl,
             i ? &acc : &acl)) {
              return -2;
            }
          }
        }
      }
      mcu++;
      if (restart_interval && mcu % restart_interval == 0 && mcu < total) {
        flush_bits(&w);
        put16(&w, 0xFFD0 + (rst & 7));
        rst++;
        dc_pred[0] = dc_pred[1] = dc_pred[2] = 0;
      }
    }
  }
  flush_bits(&w);
  put16(&w, 0xFFD9);
  return w.overflow ? -1 : w.n;
}

/* IJG quality scaling of an Annex K table (jcparam.c semantics, restated). */
static void scale_qtable(unsigned short out[64], const unsigned char base[64],
 int quality) {
  int scale, k;
  if (quality < 1) quality = 1;
  if (quality > 100) quality = 100;
  scale = quality < 50 ? 5000/quality : 200 - 2*quality;
  for (k = 0; k < 64; k++) {
    long v = ((long)base[k]*scale + 50)/100;
    if (v < 1) v = 1;
    if (v > 255) v = 255;
    out[k] = (unsigned short)v;
  }
}

JGS_API void jgs_quality_tables(int quality, unsigned short q[3][64]) {
  scale_qtable(q[0], K1_LUMA_Q, quality);
  scale_qtable(q[1], K2_CHROMA_Q, quality);
  memcpy(q[2], q[1], sizeof(q[1]));
}

/* Size of the packed coefficient buffer (shorts) for a frame; also returns
 * per-plane geometry through optional arrays of 3 ints / long longs. */
JGS_API long long jgs_coef_shorts(int width, int height, int ncomps, int hs,
 int vs, int *hblocks, int *vblocks, long long *coef_off) {
  frame f;
  int i;
  if (frame_init(&f, width, height, ncomps, hs, vs)) return -1;
  for (i = 0; i < ncomps; i++) {
    if (hblocks) hblocks[i] = f.hblocks[i];
    if (vblocks) vblocks[i] = f.vblocks[i];
    if (coef_off) coef_off[i] = f.coef_off[i];
  }
  return f.coef_shorts;
}

/* Entropy-code caller-supplied quantised levels (packed Appendix-B layout).
 * restart_interval: MCUs per interval, 0 = none, -1 = one MCU row.
 * Returns bytes written, -1 if `cap` is too small, -2 if a level is not
 * representable in baseline (|AC| > 1023 or |DC diff| > 2047), -3 bad args. */
JGS_API long jgs_encode_levels(const short *levels, int width, int height,
 int ncomps, int hs, int vs, const unsigned short q[3][64],
 int restart_interval, int flags, unsigned char *out, long cap) {
  frame f;
  if (frame_init(&f, width, height, ncomps, hs, vs)) return -3;
  if (restart_interval < 0) restart_interval = f.nhmb;
  if (restart_interval > 65535) return -3;
  return encode_levels(&f, levels, q, ncomps == 1 ? 1 : 3, restart_interval,
   flags, out, cap);
}

/* ---- pixels -> levels --------------------------------------------------- */

static float COS8[8][8];   /* COS8[u][x] = c(u)/2 * cos((2x+1)u pi/16) */
static void init_cos(void) {
  int u, x;
  if (COS8[0][0] != 0.0f) return;
  for (u = 0; u < 8; u++) {
    for (x = 0; x < 8; x++) {
      COS8[u][x] = (float)((u ? 0.5 : 0.35355339059327373)
       *cos((2*x + 1)*u*3.14159265358979323846/16.0));
    }
  }
}

static void fdct_quant(short out[64], const float px[64],
 const unsigned short q[64]) {
  float tmp[64];
  int u, v, x, y;
  for (y = 0; y < 8; y++) {
    for (u = 0; u < 8; u++) {
      float s = 0;
      for (x = 0; x < 8; x++) s += px[y*8 + x]*COS8[u][x];
      tmp[y*8 + u] = s;
    }
  }
  for (v = 0; v < 8; v++) {
    for (u = 0; u < 8; u++) {
      float s = 0, r;
      for (y = 0; y < 8; y++) s += tmp[y*8 + u]*COS8[v][y];
      r = s/(float)q[v*8 + u];
      out[v*8 + u] = (short)(r < 0 ? -(int)(-r + 0.5f) : (int)(r + 0.5f));
    }
  }
}

/* planes: ncomps full-resolution float planes (Y, Cb, Cr or grey), padded to
 * the MCU grid by edge replication; chroma is box-averaged down. */
static void planes_to_levels(const frame *f, float *const planes[3], int pw,
 const unsigned short q[3][64], short *levels) {
  int p, bx, by, x, y, i, j;
  init_cos();
  for (p = 0; p < f->ncomps; p++) {
    int sx = 1 << f->xdec[p], sy = 1 << f->ydec[p];
    float norm = 1.0f/(float)(sx*sy);
    for (by = 0; by < f->vblocks[p]; by++) {
      for (bx = 0; bx < f->hblocks[p]; bx++) {
        float px[64];
        for (y = 0; y < 8; y++) {
          for (x = 0; x < 8; x++) {
            float s = 0;
            int X = (bx*8 + x)*sx, Y = (by*8 + y)*sy;
            for (j = 0; j < sy; j++) {
              for (i = 0; i < sx; i++) s += planes[p][(long)(Y + j)*pw + X + i];
            }
            px[y*8 + x] = s*norm - 128.0f;
          }
        }
        fdct_quant(levels + block_off(f, p, bx, by), px, q[p]);
      }
    }
  }
}

static long encode_planes(const frame *f, float *const planes[3], int pw,
 int quality, int restart_interval, int flags, unsigned char *out, long cap) {
  unsigned short q[3][64];
  short *levels;
  long n;
  jgs_quality_tables(quality, q);
  levels = (short *)calloc((size_t)f->coef_shorts, sizeof(short));
  if (!levels) return -3;
  planes_to_levels(f, planes, pw, q, levels);
  n = encode_levels(f, levels, q, f->ncomps == 1 ? 1 : 2, restart_interval,
   flags, out, cap);
  free(levels);
  return n;
}

/* Encode caller pixels: grey (ncomps 1, width*height bytes) or interleaved RGB
 * (ncomps 3).  hs/vs = luma sampling factors (2,2 = 4:2:0; 2,1 = 4:2:2;
 * 1,1 = 4:4:4; 1,2 = 4:4:0; 4,1 = 4:1:1). */
JGS_API long jgs_encode_pixels(const unsigned char *pixels, int width,
 int height, int ncomps, int hs, int vs, int quality, int restart_interval,
 int flags, unsigned char *out, long cap) {
  frame f;
  float *planes[3] = {0, 0, 0};
  int pw, ph, x, y, p;
  long n;
  if (frame_init(&f, width, height, ncomps, hs, vs)) return -3;
  if (restart_interval < 0) restart_interval = f.nhmb;
  if (restart_interval > 65535) return -3;
  pw = f.nhmb*f.hmax*8;
  ph = f.nvmb*f.vmax*8;
  for (p = 0; p < ncomps; p++) {
    planes[p] = (float *)malloc(sizeof(float)*(size_t)pw*ph);
    if (!planes[p]) return -3;
  }
  for (y = 0; y < ph; y++) {
    int sy = y < height ? y : height - 1;
    for (x = 0; x < pw; x++) {
      int sx = x < width ? x : width - 1;
      const unsigned char *s = pixels + ((long)sy*width + sx)*ncomps;
      if (ncomps == 1) planes[0][(long)y*pw + x] = s[0];
      else {
        float r = s[0], g = s[1], b = s[2];
        planes[0][(long)y*pw + x] = 0.299f*r + 0.587f*g + 0.114f*b;
        planes[1][(long)y*pw + x] = -0.168736f*r - 0.331264f*g + 0.5f*b + 128.0f;
        planes[2][(long)y*pw + x] = 0.5f*r - 0.418688f*g - 0.081312f*b + 128.0f;
      }
    }
  }
  n = encode_planes(&f, planes, pw, quality, restart_interval, flags, out, cap);
  for (p = 0; p < ncomps; p++) free(planes[p]);
  return n;
}

/* xorshift64* PRNG — deterministic on every platform */
static uint64_t rng_next(uint64_t *s) {
  uint64_t x = *s;
  x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
  *s = x;
  return x*0x2545F4914F6CDD1DULL;
}

/* Fill `pixels` (width*height*ncomps) with the SURVEY.md §8(d) recipe.  Noise
 * is the sum of four uniform bytes (Irwin-Hall), scaled to sigma 12. */
JGS_API void jgs_synthetic_pixels(unsigned char *pixels, int width, int height,
 int ncomps, unsigned seed) {
  uint64_t s = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)seed*0xD1B54A32D192ED03ULL + 1);
  float *sx = (float *)malloc(sizeof(float)*(size_t)width*3);
  float *cy = (float *)malloc(sizeof(float)*(size_t)height*3);
  const float nscale = 12.0f/147.80f;   /* sqrt(4*(256^2-1)/12) = 147.80 */
  int x, y, c;
  for (c = 0; c < 3; c++) {
    for (x = 0; x < width; x++) sx[c*width + x] = (float)sin(x/(37.0 + 11.0*c));
    for (y = 0; y < height; y++) cy[c*height + y] = (float)cos(y/(53.0 + 7.0*c));
  }
  for (y = 0; y < height; y++) {
    for (x = 0; x < width; x++) {
      for (c = 0; c < ncomps; c++) {
        uint32_t r = (uint32_t)(rng_next(&s) >> 32);
        int sum = (int)(r & 255) + (int)((r >> 8) & 255) + (int)((r >> 16) & 255)
         + (int)(r >> 24);
        float v = 127.0f + 80.0f*sx[c*width + x]*cy[c*height + y]
         + ((float)sum - 510.0f)*nscale;
        pixels[((long)y*width + x)*ncomps + c] =
         (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : (int)(v + 0.5f));
      }
    }
  }
  free(sx);
  free(cy);
}

JGS_API long jgs_encode_synthetic(int width, int height, int ncomps, int hs,
 int vs, int quality, int restart_interval, unsigned seed, int flags,
 unsigned char *out, long cap) {
  unsigned char *px = (unsigned char *)malloc((size_t)width*height*ncomps);
  long n;
  if (!px) return -3;
  jgs_synthetic_pixels(px, width, height, ncomps, seed);
  n = jgs_encode_pixels(px, width, height, ncomps, hs, vs, quality,
   restart_interval, flags, out, cap);
  free(px);
  return n;
}
