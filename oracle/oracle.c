/* oracle.c — CPU restatement of the reference's JPEG block-decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (jpeg_gpu_amd/, include/)
 * may include, link, call or execute this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the reported CPU baseline.
 *
 * Every function restates one piece of negge/jpeg_gpu's CPU path and cites the
 * reference file:line it follows (paths relative to the reference root).  The
 * restatement is PINNED by tests/test_oracle_pins.py: bit-for-bit against the
 * reference's own sources compiled into oracle/_ref/ (see oracle/Makefile;
 * test_oracle_equals_reference_*), against committed vectors that were
 * produced by that compiled reference (tests/golden/, generator
 * tests/golden/make_golden.py; test_golden_*, test_pack_consumer_*), against
 * SURVEY Appendix C known answers, and by the reference's own IEEE-1180 unit
 * test procedure (test/dct.c:229-261; test_ieee1180_accuracy).
 *
 * Exception: the upsample + YCbCr->RGB stage has NO CPU code and
 * no test in the reference (only GLSL, res/unyuv.fs.glsl) — for that stage
 * orc_planes_to_rgb() below IS the definition and its
 * parity is "unpinned by the reference".
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off (never -ffast-math, never -march
 * flags that enable FMA): the reference builds -std=c89 -O2 (Makefile:20-21),
 * i.e. IEEE binary32 operation-for-operation, no contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------- */
/* Scaled float IDCT — src/dct.c                                           */
/* ---------------------------------------------------------------------- */

/* Per-axis scale factors, src/dct.c:89-98.  Written from their closed forms
 * S[0] = 1/(2*sqrt(2)), S[k] = cos(k*pi/16)/2 (doc/dct8.pdf); each rounds to
 * the same binary32 as the reference literal (bit patterns checked in
 * tests/test_oracle_pins.py). */
static const float ORC_S[8] = {
  (float)0.35355339059327373,  /* 3eb504f3 */
  (float)0.49039264020161522,  /* 3efb14be */
  (float)0.46193976625564337,  /* 3eec835e */
  (float)0.41573480615127262,  /* 3ed4db31 */
  (float)0.35355339059327373,  /* 3eb504f3 */
  (float)0.27778511650980114,  /* 3e8e39da */
  (float)0.19134171618254492,  /* 3e43ef15 */
  (float)0.097545161008064166  /* 3dc7c5c2 */
};
/* Rotation constants used inside the 1-D kernel, src/dct.c:51,62-65:
 * sqrt(2), 2cos(pi/8), 2(cos(pi/8)-sin(pi/8)), 2(cos(pi/8)+sin(pi/8)). */
#define ORC_C1 ((float)1.4142135623730951)  /* 3fb504f3 */
#define ORC_C2 ((float)1.8477590650225735)  /* 3fec835e */
#define ORC_C3 ((float)1.0823922002923938)  /* 3f8a8bd4 */
#define ORC_C4 ((float)2.6131259297527532)  /* 40273d75 */

ORC_API void orc_constants(float out[12]) {
  int i;
  for (i = 0; i < 8; i++) out[i] = ORC_S[i];
  out[8] = ORC_C1; out[9] = ORC_C2; out[10] = ORC_C3; out[11] = ORC_C4;
}

/* 1-D 8-point scaled IDCT: embedded 4-point DCT-II on the even inputs,
 * 4-point DST-IV on the odd inputs, 8 output butterflies.  Follows
 * src/dct.c:21-87 operation by operation (5 multiplies, 29 adds); the
 * association of every expression is the reference's.  Output stride `xs`. */
static void orc_idct8(float *x, int xs, const float y[8]) {
  /* even half, src/dct.c:47-55 */
  float e0 = y[0] + y[4];
  float e1 = y[0] - y[4];
  float e3 = y[2] + y[6];
  float e2 = (y[2] - y[6])*ORC_C1 - e3;
  float a0 = e0 + e3;
  float a3 = e0 - e3;
  float a1 = e1 + e2;
  float a2 = e1 - e2;
  /* odd half, src/dct.c:57-69 */
  float p5 = y[5] + y[3];
  float p6 = y[5] - y[3];
  float p7 = y[1] + y[7];
  float p4 = y[1] - y[7];
  float o7 = p7 + p5;
  float o5 = (p7 - p5)*ORC_C1;
  float o8 = (p4 + p6)*ORC_C2;
  float o4 = o8 - p4*ORC_C3;
  float o6 = o8 - p6*ORC_C4;
  float b7 = o7;
  float b6 = b7 - o6;
  float b5 = b6 + o5;
  float b4 = b5 - o4;
  /* butterflies, src/dct.c:71-86 */
  x[0*xs] = a0 + b7;
  x[1*xs] = a1 - b6;
  x[2*xs] = a2 + b5;
  x[3*xs] = a3 - b4;
  x[4*xs] = a3 + b4;
  x[5*xs] = a2 - b5;
  x[6*xs] = a1 + b6;
  x[7*xs] = a0 - b7;
}

/* 2-D: src/dct.c:100-121.  Scale with two roundings (107-108), row pass
 * stored transposed (111), +0.5 on the first entry of each column vector
 * (113), column pass (114), floor -> short (118).  In-place safe like the
 * reference (x == y allowed). */
ORC_API void orc_idct8x8(short *x, int xstride, const short *y, int ystride) {
  float t[64];
  float z[64];
  int i, j;
  for (j = 0; j < 8; j++) {
    for (i = 0; i < 8; i++) {
      t[j*8 + i] = ((float)y[j*ystride + i]*ORC_S[j])*ORC_S[i];
    }
  }
  for (i = 0; i < 8; i++) orc_idct8(z + i, 8, t + 8*i);
  for (i = 0; i < 8; i++) {
    /* the reference adds the double constant 0.5 to a float (dct.c:113); both
       operands and the sum are exact in double, so this equals a float add */
    z[8*i] = (float)((double)z[8*i] + 0.5);
    orc_idct8(t + i, 8, z + 8*i);
  }
  for (j = 0; j < 8; j++) {
    for (i = 0; i < 8; i++) {
      /* (short)floor(): x86-64 converts through int32 and keeps the low 16
         bits; made explicit so the oracle does not depend on UB */
      x[j*xstride + i] = (short)(int32_t)floor((double)t[j*8 + i]);
    }
  }
}

/* Batch helper for tests: n contiguous 64-short blocks. */
ORC_API void orc_idct8x8_blocks(short *out, const short *in, long n) {
  long b;
  for (b = 0; b < n; b++) orc_idct8x8(out + 64*b, 8, in + 64*b, 8);
}

/* ---------------------------------------------------------------------- */
/* Geometry — src/image.c:24-97, src/xjpeg.c:400-407, 556-561              */
/* ---------------------------------------------------------------------- */

typedef struct orc_plane {
  int hsamp, vsamp;
  int hblocks, vblocks;       /* MCU padded */
  int xdec, ydec;
  int width, height;          /* hblocks*8, vblocks*8 */
  int cstride;
  int id;                     /* component identifier Ci (SOF) */
  int tq;                     /* quant table index */
  int td, ta;                 /* huffman table indices (from SOS) */
  long long coef_off;         /* shorts */
} orc_plane;

typedef struct orc_frame {
  int width, height, ncomps, bits;
  int nhmb, nvmb;
  int restart_interval;
  long long coef_shorts;
  orc_plane plane[3];
  unsigned short quant[4][64];   /* natural order */
  int quant_valid[4];
  int quant_bits[4];
} orc_frame;

/* src/internal.c:46-67 (bit length of v). */
static int orc_ilog(unsigned v) {
  int n = 0;
  while (v) { n++; v >>= 1; }
  return n;
}

/* image_init, src/image.c:34-70 and 86-95. */
static void orc_frame_layout(orc_frame *f) {
  int hmax = 0, vmax = 0, i;
  long long off = 0;
  for (i = 0; i < f->ncomps; i++) {
    if (f->plane[i].hsamp > hmax) hmax = f->plane[i].hsamp;
    if (f->plane[i].vsamp > vmax) vmax = f->plane[i].vsamp;
  }
  /* src/xjpeg.c:400-407 */
  f->nhmb = (f->width + hmax*8 - 1)/(hmax*8);
  f->nvmb = (f->height + vmax*8 - 1)/(vmax*8);
  for (i = 0; i < f->ncomps; i++) {
    orc_plane *p = &f->plane[i];
    p->hblocks = f->nhmb*p->hsamp;          /* src/jpeg_wrap.c:304-305 */
    p->vblocks = f->nvmb*p->vsamp;
    p->width = p->hblocks << 3;
    p->height = p->vblocks << 3;
    p->xdec = orc_ilog(hmax) - orc_ilog(p->hsamp);
    p->ydec = orc_ilog(vmax) - orc_ilog(p->vsamp);
    p->cstride = (p->vblocks + ((1 << p->xdec) - 1)) >> p->xdec;
  }
  for (i = 0; i < f->ncomps; i++) {
    orc_plane *p = &f->plane[i];
    p->coef_off = off;
    /* src/image.c:91-92: (width << (xdec+3))*cstride, with width the PLANE's
       width; equals luma row size when hblocks<<xdec == luma hblocks */
    off += ((long long)p->width << (p->xdec + 3))*p->cstride;
  }
  f->coef_shorts = off;
}

/* src/xjpeg.c:556-561 */
static long long orc_block_off(const orc_frame *f, int pi, int bx, int by) {
  const orc_plane *p = &f->plane[pi];
  long long rs = (long long)f->plane[0].width << 3;
  return p->coef_off + rs*(by >> p->xdec)
   + (rs >> p->xdec)*(by & ((1 << p->xdec) - 1)) + ((long long)bx << 6);
}

/* ---------------------------------------------------------------------- */
/* Entropy decode — src/xjpeg.c                                            */
/* ---------------------------------------------------------------------- */

typedef struct orc_huff {
  int valid;
  int nbits[16];
  int nsymbs;
  unsigned short codeword[256];
  unsigned char symbol[256];
  int lookup[256];
  int index[16];
  int maxcode[16];
} orc_huff;

typedef struct orc_dec {
  const unsigned char *buf;
  long pos, size;
  uint64_t bitbuf;
  int bits;
  int marker;
  const char *error;
  orc_huff dc[4], ac[4];
  orc_frame f;
  int frame_valid, scan_valid, soi, eoi;
  int scan_ncomps;
  int scan_plane[3];
} orc_dec;

/* ITU-T T.81 Figure A.6 zig-zag scan, generated (the reference keeps its
 * inverse as a literal table, src/xjpeg.c:44-53). */
static int ORC_DEZIGZAG[64];
static void orc_init_zigzag(void) {
  int k = 0, s, i;
  if (ORC_DEZIGZAG[63] == 63) return;
  for (s = 0; s < 15; s++) {
    for (i = 0; i <= s; i++) {
      int r = (s & 1) ? i : s - i;
      int c = s - r;
      if (r < 8 && c < 8) ORC_DEZIGZAG[k++] = r*8 + c;
    }
  }
}

ORC_API void orc_dezigzag(int out[64]) {
  orc_init_zigzag();
  memcpy(out, ORC_DEZIGZAG, sizeof(ORC_DEZIGZAG));
}

#define ORC_FAIL(d, msg) do { (d)->error = (msg); return; } while (0)

static int orc_u8(orc_dec *d) {
  if (d->pos >= d->size) { d->error = "read past end of file"; return 0; }
  return d->buf[d->pos++];
}
static int orc_u16(orc_dec *d) {
  int hi = orc_u8(d);
  return (hi << 8) | orc_u8(d);
}

/* XJPEG_FILL_BYTE, src/xjpeg.c:113-127: append one byte; FF 00 unstuffs; on a
 * real marker remember it, rewind to the FF and feed zero bits. */
static void orc_fill_byte(orc_dec *d) {
  int byte;
  if (d->pos >= d->size) {          /* bounds check the reference lacks */
    d->bits += 8;
    d->bitbuf <<= 8;
    return;
  }
  byte = d->buf[d->pos++];
  d->bits += 8;
  d->bitbuf = (d->bitbuf << 8) | (uint64_t)byte;
  if (byte == 0xFF) {
    int next = d->pos < d->size ? d->buf[d->pos] : 0xD9;
    d->pos++;
    if (next != 0x00) {
      d->marker = next;
      d->pos -= 2;
      d->bitbuf &= ~(uint64_t)0xFF;
    }
  }
}
/* XJPEG_FILL_BITS, src/xjpeg.c:129-140 */
static void orc_fill_bits(orc_dec *d) {
  if (d->bits <= 16) {
    int k;
    for (k = 0; k < 6; k++) orc_fill_byte(d);
  }
}
static int orc_get_bits(orc_dec *d, int n) {
  int v;
  if (n == 0) return 0;
  v = (int)((d->bitbuf >> (d->bits - n)) & ((1u << n) - 1));
  d->bits -= n;
  return v;
}
/* XJPEG_DECODE_HUFF, src/xjpeg.c:163-187: 8-bit LUT, then canonical walk. */
static int orc_huff_symbol(orc_dec *d, const orc_huff *h) {
  int value, lookup, bits, symbol;
  orc_fill_bits(d);
  value = (int)((d->bitbuf >> (d->bits - 8)) & 0xFF);
  lookup = h->lookup[value];
  bits = lookup >> 8;
  symbol = lookup & 0xFF;
  d->bits -= bits;
  if (bits > 8) {
    value = (int)((d->bitbuf >> d->bits) & ((1u << bits) - 1));
    while (value > h->maxcode[bits - 1]) {
      if (bits >= 16) { d->error = "invalid huffman code"; return 0; }
      value = (value << 1) | orc_get_bits(d, 1);
      bits++;
    }
    symbol = h->symbol[(value + h->index[bits - 1]) & 0xFF];
  }
  return symbol;
}
/* XJPEG_DECODE_VLC + XJPEG_HUFF_EXTEND, src/xjpeg.c:189-205.  len==0 yields
 * 0 (the reference's macro shifts by -1 there; x86 gives 0, Appendix E). */
static int orc_vlc(orc_dec *d, const orc_huff *h, int *symbol) {
  int len, v;
  *symbol = orc_huff_symbol(d, h);
  len = *symbol & 0xF;
  orc_fill_bits(d);
  v = orc_get_bits(d, len);
  if (len && v < (1 << (len - 1))) v += (int)((~0u << len) + 1);
  return v;
}

/* src/xjpeg.c:219-256 */
static void orc_dqt(orc_dec *d) {
  int len = orc_u16(d) - 2;
  orc_init_zigzag();
  while (len >= 65 && !d->error) {
    int b = orc_u8(d), pq = b >> 4, tq = b & 7, i;
    if (pq > 1 || tq > 3) ORC_FAIL(d, "bad DQT");
    d->f.quant_valid[tq] = 1;
    d->f.quant_bits[tq] = pq ? 16 : 8;
    for (i = 0; i < 64; i++) {
      d->f.quant[tq][ORC_DEZIGZAG[i]] =
       (unsigned short)(pq ? orc_u16(d) : orc_u8(d));
    }
    len -= 65 + 64*pq;
  }
  if (len != 0) ORC_FAIL(d, "DQT length");
}

/* src/xjpeg.c:258-345 */
static void orc_dht(orc_dec *d) {
  int len = orc_u16(d) - 2;
  while (len >= 17 && !d->error) {
    int b = orc_u8(d), tc = b >> 4, th = b & 7, i, j, k, l;
    unsigned code;
    orc_huff *h;
    if (tc > 1 || th > 3) ORC_FAIL(d, "bad DHT");
    h = tc ? &d->ac[th] : &d->dc[th];
    h->valid = 1;
    h->nsymbs = 0;
    for (i = 0; i < 16; i++) {
      h->nbits[i] = orc_u8(d);
      h->nsymbs += h->nbits[i];
    }
    len -= 17;
    if (h->nsymbs > 256 || h->nsymbs > len) ORC_FAIL(d, "DHT symbols");
    /* canonical codewords, 293-310 */
    k = 0;
    code = 0;
    for (i = 0; i < 16; i++) {
      for (j = 0; j < h->nbits[i]; j++) {
        h->codeword[k] = (unsigned short)code;
        h->symbol[k] = (unsigned char)orc_u8(d);
        k++;
        code++;
      }
      len -= h->nbits[i];
      if (code > (1u << (i + 1))) ORC_FAIL(d, "invalid DHT");
      code <<= 1;
    }
    /* 8-bit lookup, 312-325: entry = (length << 8) | symbol, 9<<8 = miss */
    for (i = 0; i < 256; i++) h->lookup[i] = 9 << 8;
    k = 0;
    for (i = 1; i <= 8; i++) {
      for (j = 0; j < h->nbits[i - 1]; j++) {
        unsigned c = (unsigned)h->codeword[k] << (8 - i);
        for (l = 0; l < 1 << (8 - i); l++) {
          h->lookup[(c + l) & 0xFF] = (i << 8) | h->symbol[k];
        }
        k++;
      }
    }
    /* maxcode / index per length, 328-336 */
    k = 0;
    for (i = 0; i < 16; i++) {
      h->maxcode[i] = -1;
      h->index[i] = 0;
      if (h->nbits[i]) {
        h->index[i] = k - h->codeword[k];
        k += h->nbits[i];
        h->maxcode[i] = h->codeword[k - 1];
      }
    }
  }
  if (len != 0) ORC_FAIL(d, "DHT length");
}

/* src/xjpeg.c:350-410 */
static void orc_sof(orc_dec *d) {
  int len = orc_u16(d) - 2, i;
  if (len < 9 || d->frame_valid) ORC_FAIL(d, "bad SOF");
  d->frame_valid = 1;
  d->f.bits = orc_u8(d);
  d->f.height = orc_u16(d);
  d->f.width = orc_u16(d);
  d->f.ncomps = orc_u8(d);
  len -= 6;
  if (d->f.width == 0 || d->f.height == 0) ORC_FAIL(d, "SOF size");
  if (d->f.ncomps != 1 && d->f.ncomps != 3) ORC_FAIL(d, "SOF ncomps");
  if (d->f.bits != 8) ORC_FAIL(d, "SOF precision");
  if (len != 3*d->f.ncomps) ORC_FAIL(d, "SOF length");
  for (i = 0; i < d->f.ncomps; i++) {
    orc_plane *p = &d->f.plane[i];
    int b;
    p->id = orc_u8(d);
    b = orc_u8(d);
    p->hsamp = b >> 4;
    p->vsamp = b & 7;
    p->tq = orc_u8(d);
    if (p->hsamp < 1 || p->hsamp > 4 || p->hsamp == 3
     || p->vsamp < 1 || p->vsamp > 4 || p->vsamp == 3 || p->tq > 3) {
      ORC_FAIL(d, "SOF sampling");
    }
    if (!d->f.quant_valid[p->tq]) ORC_FAIL(d, "SOF quant table");
  }
  orc_frame_layout(&d->f);
}

/* src/xjpeg.c:412-420 */
static void orc_dri(orc_dec *d) {
  int len = orc_u16(d);
  if (len != 4) ORC_FAIL(d, "bad DRI");
  d->f.restart_interval = orc_u16(d);
}

/* out modes mirror xjpeg_decode_out (src/xjpeg.h:137-143) */
enum { ORC_QUANT = 1, ORC_DCT = 2, ORC_YUV = 3 };

/* GLJ_CLAMP255, src/internal.h:36-37 */
static unsigned char orc_clamp255(int x) {
  return (unsigned char)(x < 0 ? 0 : x > 255 ? 255 : x);
}

/* xjpeg_decode_scan, src/xjpeg.c:449-632 */
static void orc_scan(orc_dec *d, int out, short *coef, unsigned char **planes) {
  orc_frame *f = &d->f;
  short dc_pred[3] = {0, 0, 0};
  int mcu_counter = f->restart_interval;
  int rst_counter = 0;
  int mbx, mby, i;
  orc_init_zigzag();
  for (mby = 0; mby < f->nvmb; mby++) {
    for (mbx = 0; mbx < f->nhmb; mbx++) {
      for (i = 0; i < d->scan_ncomps; i++) {
        /* like the reference (xjpeg.c:437-443, Appendix E) component i of
           the scan is assumed to be component i of the frame */
        orc_plane *p = &f->plane[i];
        const orc_huff *hdc = &d->dc[p->td];
        const orc_huff *hac = &d->ac[p->ta];
        const unsigned short *q = f->quant[p->tq];
        int sbx, sby;
        for (sby = 0; sby < p->vsamp; sby++) {
          for (sbx = 0; sbx < p->hsamp; sbx++) {
            short block[64];
            int symbol, j = 0;
            short value;
            int by = mby*p->vsamp + sby;
            int bx = mbx*p->hsamp + sbx;
            memset(block, 0, sizeof(block));
            value = (short)orc_vlc(d, hdc, &symbol);
            dc_pred[i] = (short)(dc_pred[i] + value);       /* 480 */
            if (out == ORC_QUANT) block[0] = dc_pred[i];    /* 498-499 */
            else block[0] = (short)(dc_pred[i]*q[0]);       /* 501-503 */
            do {
              value = (short)orc_vlc(d, hac, &symbol);
              if (d->error) return;
              if (!symbol) break;                           /* EOB 530-539 */
              j += (symbol >> 4) + 1;                       /* 508 */
              if (j > 63) ORC_FAIL(d, "coefficient index outside block");
              if (out == ORC_QUANT) block[ORC_DEZIGZAG[j]] = value;
              else {
                block[ORC_DEZIGZAG[j]] =
                 (short)(value*q[ORC_DEZIGZAG[j]]);         /* 524-527 */
              }
            }
            while (j < 63);
            if (out == ORC_QUANT || out == ORC_DCT) {
              memcpy(coef + orc_block_off(f, i, bx, by), block,
               sizeof(block));                              /* 550-563 */
            }
            else {
              /* 565-584 */
              unsigned char *row = planes[i] + (long)by*8*p->width + bx*8;
              int k, c;
              orc_idct8x8(block, 8, block, 8);
              for (k = 0; k < 8; k++) {
                for (c = 0; c < 8; c++) row[c] = orc_clamp255(block[k*8 + c] + 128);
                row += p->width;
              }
            }
          }
        }
      }
      /* restart handling, 593-629 */
      mcu_counter--;
      if (f->restart_interval && mcu_counter == 0) {
        int m;
        if (d->pos + 1 >= d->size || d->buf[d->pos] != 0xFF) {
          ORC_FAIL(d, "expected marker at restart");
        }
        m = d->buf[d->pos + 1];
        d->pos += 2;
        if (m >= 0xD0 && m <= 0xD7) {
          if ((m & 7) != (rst_counter & 7)) ORC_FAIL(d, "RST out of order");
          d->marker = 0;
          d->bits = 0;
          mcu_counter = f->restart_interval;
          rst_counter++;
          dc_pred[0] = dc_pred[1] = dc_pred[2] = 0;
        }
        else if (m == 0xD9) {
          d->marker = m;
          return;
        }
        else ORC_FAIL(d, "unknown marker in scan");
      }
    }
  }
}

/* xjpeg_decode_sos, src/xjpeg.c:634-695 */
static void orc_sos(orc_dec *d, int out, short *coef, unsigned char **planes) {
  int len = orc_u16(d) - 2, i, j, b;
  if (len < 6 || d->scan_valid || !d->frame_valid) ORC_FAIL(d, "bad SOS");
  d->scan_valid = 1;
  d->scan_ncomps = orc_u8(d);
  if (d->scan_ncomps != d->f.ncomps) ORC_FAIL(d, "SOS ncomps");
  for (i = 0; i < d->scan_ncomps; i++) {
    int id = orc_u8(d), found = -1;
    for (j = 0; j < d->f.ncomps; j++) if (d->f.plane[j].id == id) found = j;
    if (found < 0) ORC_FAIL(d, "SOS component");
    /* table selectors are recorded per SCAN index and applied to frame
       component i (xjpeg.c:437-443): only SOS order == SOF order is sane */
    if (found != i) ORC_FAIL(d, "SOS order differs from SOF order");
    d->scan_plane[i] = found;
    b = orc_u8(d);
    d->f.plane[i].td = b >> 4;
    d->f.plane[i].ta = b & 7;
    if (d->f.plane[i].td > 3 || d->f.plane[i].ta > 3
     || !d->dc[d->f.plane[i].td].valid || !d->ac[d->f.plane[i].ta].valid) {
      ORC_FAIL(d, "SOS huffman table");
    }
  }
  if (orc_u8(d) != 0 || orc_u8(d) != 63 || orc_u8(d) != 0) {
    ORC_FAIL(d, "SOS spectral selection (baseline only)");
  }
  if (d->error) return;
  d->bits = 0;
  d->bitbuf = 0;
  orc_scan(d, out, coef, planes);
}

/* xjpeg_decode, src/xjpeg.c:704-763 */
static void orc_run(orc_dec *d, int headers_only, int out, short *coef,
 unsigned char **planes) {
  while (!d->error && !d->eoi) {
    int marker = d->marker;
    d->marker = 0;
    if (marker == 0) {
      if (d->pos + 2 > d->size) ORC_FAIL(d, "underflow reading marker");
      if (d->buf[d->pos] != 0xFF) ORC_FAIL(d, "invalid JPEG syntax");
      marker = d->buf[d->pos + 1];
      d->pos += 2;
    }
    if (headers_only && marker == 0xDA) {
      d->marker = marker;
      return;
    }
    switch (marker) {
      case 0xD8 : d->soi = 1; break;
      case 0xD9 : d->eoi = 1; break;
      case 0xDB : orc_dqt(d); break;
      case 0xC4 : orc_dht(d); break;
      case 0xC0 : orc_sof(d); break;
      case 0xDD : orc_dri(d); break;
      case 0xDA : orc_sos(d, out, coef, planes); break;
      default : {
        int len = orc_u16(d);
        if (len < 2 || d->pos + len - 2 > d->size) ORC_FAIL(d, "bad segment");
        d->pos += len - 2;
      }
    }
  }
}

static void orc_dec_init(orc_dec *d, const unsigned char *buf, long size) {
  memset(d, 0, sizeof(*d));
  d->buf = buf;
  d->size = size;
  if (size < 4 || buf[0] != 0xFF || buf[1] != 0xD8 || buf[2] != 0xFF) {
    d->error = "not a JPEG (invalid SOI marker)";
  }
}

/* Flat, ctypes-friendly description of a parsed frame. */
typedef struct orc_info {
  int width, height, ncomps, restart_interval;
  int nhmb, nvmb;
  long long coef_shorts;
  int hsamp[3], vsamp[3], hblocks[3], vblocks[3], xdec[3], ydec[3];
  int cstride[3], tq[3];
  long long coef_off[3];
  unsigned short quant[3][64];    /* per PLANE, natural order */
} orc_info;

static void orc_export_info(const orc_frame *f, orc_info *o) {
  int i;
  memset(o, 0, sizeof(*o));
  o->width = f->width; o->height = f->height; o->ncomps = f->ncomps;
  o->restart_interval = f->restart_interval;
  o->nhmb = f->nhmb; o->nvmb = f->nvmb;
  o->coef_shorts = f->coef_shorts;
  for (i = 0; i < f->ncomps; i++) {
    const orc_plane *p = &f->plane[i];
    o->hsamp[i] = p->hsamp; o->vsamp[i] = p->vsamp;
    o->hblocks[i] = p->hblocks; o->vblocks[i] = p->vblocks;
    o->xdec[i] = p->xdec; o->ydec[i] = p->ydec;
    o->cstride[i] = p->cstride; o->tq[i] = p->tq;
    o->coef_off[i] = p->coef_off;
    memcpy(o->quant[i], f->quant[p->tq], sizeof(o->quant[i]));
  }
}

/* Header pass (xjpeg_decode_header, src/xjpeg.c:765-767 + the copy-out of
 * src/jpeg_wrap.c:263-319).  Returns 0 on success. */
ORC_API int orc_parse(const unsigned char *buf, long size, orc_info *info,
 const char **err) {
  orc_dec d;
  orc_dec_init(&d, buf, size);
  orc_run(&d, 1, 0, NULL, NULL);
  if (!d.error && !d.frame_valid) d.error = "no SOF0 frame header";
  if (err) *err = d.error;
  if (d.error) return 1;
  orc_export_info(&d.f, info);
  return 0;
}

/* Full decode to a stage.  out = ORC_QUANT / ORC_DCT fill `coef`
 * (info.coef_shorts shorts, caller-zeroed); ORC_YUV fills planes[i]
 * (width*height of plane i). */
ORC_API int orc_decode(const unsigned char *buf, long size, int out,
 short *coef, unsigned char *p0, unsigned char *p1, unsigned char *p2,
 orc_info *info, const char **err) {
  orc_dec d;
  unsigned char *planes[3];
  planes[0] = p0; planes[1] = p1; planes[2] = p2;
  orc_dec_init(&d, buf, size);
  orc_run(&d, 0, out, coef, planes);
  if (!d.error && !d.scan_valid) d.error = "no scan";
  if (err) *err = d.error;
  if (info && d.frame_valid) orc_export_info(&d.f, info);
  return d.error ? 1 : 0;
}

/* ---------------------------------------------------------------------- */
/* Device-stage restatement on coefficient planes                          */
/* ---------------------------------------------------------------------- */

/* What the GPU stage must compute from a QUANT-stage buffer: dequantise
 * (src/xjpeg.c:501-503, 524-527: int multiply truncated to int16), IDCT
 * (src/dct.c:100-121), level shift + clamp + store (src/xjpeg.c:565-584).
 * `dequant` = 0 treats coef as DCT-stage input (already multiplied). */
ORC_API void orc_coef_to_planes(const orc_info *o, const short *coef,
 int dequant, unsigned char *p0, unsigned char *p1, unsigned char *p2) {
  unsigned char *planes[3];
  int i, bx, by, k, c;
  long long rs = (long long)o->hblocks[0] << 6;   /* (W0<<3) shorts */
  planes[0] = p0; planes[1] = p1; planes[2] = p2;
  for (i = 0; i < o->ncomps; i++) {
    int width = o->hblocks[i]*8;
    for (by = 0; by < o->vblocks[i]; by++) {
      for (bx = 0; bx < o->hblocks[i]; bx++) {
        const short *src = coef + o->coef_off[i] + rs*(by >> o->xdec[i])
         + (rs >> o->xdec[i])*(by & ((1 << o->xdec[i]) - 1)) + ((long long)bx << 6);
        short block[64];
        unsigned char *row = planes[i] + (long)by*8*width + bx*8;
        for (k = 0; k < 64; k++) {
          block[k] = dequant ? (short)((int)src[k]*(int)o->quant[i][k]) : src[k];
        }
        orc_idct8x8(block, 8, block, 8);
        for (k = 0; k < 8; k++) {
          for (c = 0; c < 8; c++) row[c] = orc_clamp255(block[k*8 + c] + 128);
          row += width;
        }
      }
    }
  }
}

/* PACK consumer: restatement of the expansion loop of the reference's first
 * PACK pass, res/horz_pack_yuv.fs.glsl:94-127 — block zeroed (110), word 0 =
 * DC with 12-bit sign extension (112), then per word: 0 ends the block
 * (117-119), otherwise `j += run + 1` (121, 124) and the sign-extended level
 * lands at DE_ZIG_ZAG[j] (125); the loop runs while j < 63 (114).  Producer:
 * src/xjpeg.c:484-496, 513-519, 531-535.  The shader indexes out of bounds
 * when a run passes coefficient 63 and reads past the buffer on a truncated
 * stream (undefined in GLSL); this restatement DEFINES both as "the block ends
 * there".  starts[b] = index of word 0 of block b; out = nblocks x 64 shorts,
 * natural order. */
ORC_API void orc_unpack_blocks(const unsigned short *pack, long long nwords,
 const int *starts, long nblocks, short *out) {
  long b;
  orc_init_zigzag();
  for (b = 0; b < nblocks; b++) {
    short *blk = out + b*64;
    long long i = starts[b];
    int j = 0;
    unsigned p;
    memset(blk, 0, 64*sizeof(short));
    if (i < 0 || i >= nwords) continue;
    p = pack[i++];
    blk[0] = (short)((p & 0xfff) | ((p & 0x800) ? ~0xfff : 0));
    while (j < 63 && i < nwords) {
      p = pack[i++];
      if (p == 0) break;
      j += (int)((p >> 12) & 0xf) + 1;
      if (j > 63) break;
      blk[ORC_DEZIGZAG[j]] = (short)((p & 0xfff) | ((p & 0x800) ? ~0xfff : 0));
    }
  }
}

/* Upsample + YCbCr->RGB: restatement of res/unyuv.fs.glsl:12-16, 29-31,
 * 39-41, 48-49 (nearest replication s>>xdec, t>>ydec; JFIF float matrix in
 * GLSL mat3*vec3 column order) and res/ungrey.fs.glsl:18, per SURVEY.md
 * Appendix A.5.  The reference has no CPU code for this stage (F4): this
 * function is the definition.  Output: true-size, row pitch width*ncomps
 * (the img->pixels convention of src/jpeg_wrap.c:215-219): 3 B/px colour,
 * 1 B/px grey. */
static unsigned char orc_unorm8(float c) {
  float m = c < 0.0f ? 0.0f : c;
  m = m > 255.0f ? 255.0f : m;
  return (unsigned char)(int)(m + 0.5f);
}

ORC_API void orc_planes_to_rgb(const orc_info *o, const unsigned char *p0,
 const unsigned char *p1, const unsigned char *p2, unsigned char *rgb) {
  int x, y;
  int w0 = o->hblocks[0]*8;
  if (o->ncomps == 1) {
    for (y = 0; y < o->height; y++) {
      memcpy(rgb + (long)y*o->width, p0 + (long)y*w0, o->width);
    }
    return;
  }
  {
    int w1 = o->hblocks[1]*8, w2 = o->hblocks[2]*8;
    for (y = 0; y < o->height; y++) {
      unsigned char *dst = rgb + (long)y*o->width*3;
      for (x = 0; x < o->width; x++) {
        float Y = (float)p0[(long)y*w0 + x];
        float u = (float)p1[(long)(y >> o->ydec[1])*w1 + (x >> o->xdec[1])] - 128.0f;
        float v = (float)p2[(long)(y >> o->ydec[2])*w2 + (x >> o->xdec[2])] - 128.0f;
        float r = Y + 1.402f*v;
        float g = (Y + (-0.34414f)*u) + (-0.71414f)*v;
        float b = Y + 1.772f*u;
        dst[3*x + 0] = orc_unorm8(r);
        dst[3*x + 1] = orc_unorm8(g);
        dst[3*x + 2] = orc_unorm8(b);
      }
    }
  }
}

/* Whole CPU path on one file: entropy decode with in-loop IDCT (the
 * reference's YUV stage) + the RGB stage above.  This is what bench.py times
 * as cpu_baseline kind "port".  `scratch` must hold the three padded planes. */
ORC_API int orc_decode_rgb(const unsigned char *buf, long size,
 unsigned char *scratch, unsigned char *rgb, orc_info *info) {
  orc_info local;
  const char *err = NULL;
  unsigned char *p0, *p1, *p2;
  if (orc_parse(buf, size, &local, &err)) return 1;
  p0 = scratch;
  p1 = p0 + (long)local.hblocks[0]*local.vblocks[0]*64;
  p2 = local.ncomps == 3 ? p1 + (long)local.hblocks[1]*local.vblocks[1]*64 : p1;
  if (orc_decode(buf, size, ORC_YUV, NULL, p0, p1, p2, &local, &err)) return 1;
  orc_planes_to_rgb(&local, p0, p1, p2, rgb);
  if (info) *info = local;
  return 0;
}
