/* entropy.c — host entropy stage: baseline JPEG bytes -> packed coefficient
 * planes.  north_star keeps this stage on the host ("the branchy Huffman
 * entropy decode ... stays on the host in C"); it feeds the HIP kernels.
 *
 * Replaces, with the same outputs on every conforming baseline stream:
 *   marker loop ............ reference src/xjpeg.c:704-763
 *   DQT / DHT / SOF0 / DRI .. src/xjpeg.c:219-256, 258-345, 350-410, 412-420
 *   SOS + scan loop ........ src/xjpeg.c:634-695, 449-632
 *   header copy-out ........ src/jpeg_wrap.c:263-319
 * Output contract: 64 de-zigzagged int16 per block
 * at jga_block_offset(); QUANT stage = raw levels with the DC predictor
 * accumulated in int16 (xjpeg.c:480, 498-499); DCT stage = level*q truncated
 * to int16 (xjpeg.c:501-503, 524-527); PACK = RLE words (xjpeg.c:484-496,
 * 513-519, 531-535).
 *
 * This is NOT the reference's decoder restated (that lives in oracle/): it is
 * a bounds-checked decoder with a 64-bit bit window refilled 8 bytes at a
 * time, a 10-bit Huffman lookup and a combined code+magnitude AC lookup.
 * Unlike the reference's default build (no validation) it never
 * reads or writes out of bounds on malformed input.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "jga_internal.h"

#define FAST_BITS 10

typedef struct htab {
  int valid;
  uint32_t fast[1 << FAST_BITS];     /* tot | len << 8 | symbol << 16 (tot = len + the symbol's magnitude bits), 0 = longer code */
  uint32_t maxcode[18];              /* left-aligned 16-bit exclusive bounds */
  int32_t delta[17];                 /* symbol index = code + delta[len] */
  uint8_t sym[256];
  int nsym;
} htab;

typedef struct comp_info {
  int id, hs, vs, tq, td, ta;
} comp_info;

typedef struct parser {
  const uint8_t *buf;
  long size, pos;
  int width, height, bits, ncomps;
  int restart_interval;
  int frame_valid, scan_valid;
  comp_info comp[3];
  jpeg_quant quant[NQUANT_MAX];
  htab dc[4], ac[4];
  uint8_t dht_bits[8][16];        /* raw DHT (tc*4+th), for the GPU entropy stage */
  uint8_t dht_vals[8][256];
} parser;

/* zig-zag position -> natural index (T.81 Fig. A.6); a constant, so the parsing threads of
 * the pipeline and of jga_huff_prepare share it without initialisation order */
static const uint8_t DEZZ[64] = {
   0,  1,  8, 16,  9,  2,  3, 10,
  17, 24, 32, 25, 18, 11,  4,  5,
  12, 19, 26, 33, 40, 48, 41, 34,
  27, 20, 13,  6,  7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36,
  29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46,
  53, 60, 61, 54, 47, 55, 62, 63
};

/* ---- segment parsing ---------------------------------------------------- */

static int need(parser *ps, long n) { return ps->pos + n <= ps->size; }
static int rd8(parser *ps) { return ps->buf[ps->pos++]; }
static int rd16(parser *ps) {
  int v = (ps->buf[ps->pos] << 8) | ps->buf[ps->pos + 1];
  ps->pos += 2;
  return v;
}

static int parse_dqt(parser *ps, long end) {
  while (ps->pos < end) {
    int b, pq, tq, k;
    b = rd8(ps);
    pq = b >> 4;
    tq = b & 15;
    if (pq > 1 || tq > 3) return jga_fail("Error DQT expected Pq 0..1, Tq 0..3.");
    if (ps->pos + 64*(pq + 1) > end) return jga_fail("Error decoding DQT, truncated.");
    ps->quant[tq].valid = 1;
    ps->quant[tq].bits = pq ? 16 : 8;
    for (k = 0; k < 64; k++) {
      ps->quant[tq].tbl[DEZZ[k]] = (unsigned short)(pq ? rd16(ps) : rd8(ps));
    }
  }
  return EXIT_SUCCESS;
}

static int build_htab(htab *h, const uint8_t counts[16], const uint8_t *syms,
 int is_ac) {
  unsigned code = 0;
  int k = 0, len, i;
  uint8_t size[257];
  uint16_t codes[256];
  memset(h, 0, sizeof(*h));
  for (len = 1; len <= 16; len++) {
    for (i = 0; i < counts[len - 1]; i++) {
      if (k >= 256) return 1;
      size[k] = (uint8_t)len;
      codes[k] = (uint16_t)code;
      h->sym[k] = syms[k];
      k++;
      code++;
    }
    if (code > (1u << len)) return 1;         /* over-subscribed */
    h->delta[len] = k - (int)code;            /* index of code c = c + delta */
    h->maxcode[len] = code << (16 - len);
    code <<= 1;
  }
  h->maxcode[17] = 0xFFFFFFFFu;
  h->nsym = k;
  for (i = 0; i < k; i++) {
    int s = size[i];
    if (s <= FAST_BITS) {
      unsigned c = (unsigned)codes[i] << (FAST_BITS - s);
      unsigned m = 1u << (FAST_BITS - s), j;
      /* what the symbol takes from the stream in all: its code and, behind it, the
         magnitude bits its low nibble announces (DC: the category; AC: SSSS) */
      for (j = 0; j < m; j++) {
        h->fast[c + j] = (uint32_t)(s + (h->sym[i] & 15)) | ((uint32_t)s << 8) | ((uint32_t)h->sym[i] << 16);
      }
    }
  }
  h->valid = 1;
  return 0;
}

static int parse_dht(parser *ps, long end) {
  while (ps->pos < end) {
    int b, tc, th, n = 0, i;
    const uint8_t *counts;
    if (ps->pos + 17 > end) return jga_fail("Error decoding DHT, truncated.");
    b = rd8(ps);
    tc = b >> 4;
    th = b & 15;
    if (tc > 1 || th > 3) return jga_fail("Error DHT expected Tc 0..1, Th 0..3.");
    counts = ps->buf + ps->pos;
    ps->pos += 16;
    for (i = 0; i < 16; i++) n += counts[i];
    if (n > 256 || ps->pos + n > end) {
      return jga_fail("Error DHT needs more bytes than available.");
    }
    if (build_htab(tc ? &ps->ac[th] : &ps->dc[th], counts, ps->buf + ps->pos, tc)) {
      return jga_fail("Error invalid DHT.");
    }
    memcpy(ps->dht_bits[tc*4 + th], counts, 16);
    memset(ps->dht_vals[tc*4 + th], 0, 256);
    memcpy(ps->dht_vals[tc*4 + th], ps->buf + ps->pos, (size_t)n);
    if (!tc) {
      for (i = 0; i < n; i++) {
        if (ps->buf[ps->pos + i] > 15) return jga_fail("Error invalid DC symbol.");
      }
    }
    ps->pos += n;
  }
  return EXIT_SUCCESS;
}

static int parse_sof0(parser *ps, long end) {
  int i;
  if (ps->frame_valid) return jga_fail("Error multiple SOF not supported.");
  if (end - ps->pos < 6) return jga_fail("Error SOF needs at least 9 bytes");
  ps->bits = rd8(ps);
  ps->height = rd16(ps);
  ps->width = rd16(ps);
  ps->ncomps = rd8(ps);
  if (ps->bits != 8) return jga_fail("Unsupported sample precision %i", ps->bits);
  if (!ps->height) return jga_fail("Error SOF has invalid height.");
  if (!ps->width) return jga_fail("Error SOF has invalid width.");
  if (ps->ncomps != 1 && ps->ncomps != 3) {
    return jga_fail("Unsupported number of components %i", ps->ncomps);
  }
  if (end - ps->pos != 3*ps->ncomps) {
    return jga_fail("Error decoding SOF, wrong length.");
  }
  for (i = 0; i < ps->ncomps; i++) {
    comp_info *c = &ps->comp[i];
    int b;
    c->id = rd8(ps);
    b = rd8(ps);
    c->hs = b >> 4;
    c->vs = b & 15;
    c->tq = rd8(ps);
    if (c->hs != 1 && c->hs != 2 && c->hs != 4) {
      return jga_fail("Unsupported horizontal sampling.");
    }
    if (c->vs != 1 && c->vs != 2 && c->vs != 4) {
      return jga_fail("Unsupported vertical sampling.");
    }
    if (c->tq > 3) return jga_fail("Error SOF expected Tq value 0 to 3.");
  }
  ps->frame_valid = 1;
  return EXIT_SUCCESS;
}

static int parse_sos(parser *ps, long end) {
  int n, i, j;
  if (!ps->frame_valid) return jga_fail("Error SOS before SOF.");
  if (ps->scan_valid) return jga_fail("Error multiple SOS not supported.");
  if (end - ps->pos < 1) return jga_fail("Error SOS needs at least 6 bytes");
  n = rd8(ps);
  if (n != ps->ncomps) {
    return jga_fail("Error only single-scan files with all components supported");
  }
  if (end - ps->pos != 2*n + 3) return jga_fail("Error decoding SOS, wrong length.");
  for (i = 0; i < n; i++) {
    int id = rd8(ps), b = rd8(ps);
    j = i;
    /* scan order must equal frame order (the reference assumes it,
       src/xjpeg.c:437-443; interleaving order is defined by the frame) */
    if (ps->comp[j].id != id) {
      return jga_fail("Error SOS component order differs from SOF.");
    }
    ps->comp[j].td = b >> 4;
    ps->comp[j].ta = b & 15;
    if (ps->comp[j].td > 3 || !ps->dc[ps->comp[j].td].valid) {
      return jga_fail("Error SOS component references invalid DC entropy table.");
    }
    if (ps->comp[j].ta > 3 || !ps->ac[ps->comp[j].ta].valid) {
      return jga_fail("Error SOS component references invalid AC entropy table.");
    }
  }
  if (rd8(ps) != 0) return jga_fail("Error SOS expected Ss value 0.");
  if (rd8(ps) != 63) return jga_fail("Error SOS expected Se value 63.");
  if (rd8(ps) != 0) return jga_fail("Error SOS expected Ah/Al value 0.");
  for (i = 0; i < n; i++) {
    if (!ps->quant[ps->comp[i].tq].valid) {
      return jga_fail("Invalid quantization table for components %i", i);
    }
  }
  ps->scan_valid = 1;
  return EXIT_SUCCESS;
}

/* Walk the marker segments up to and including SOS.  On success ps->pos is the
 * first byte of entropy-coded data. */
static int parse_to_scan(parser *ps, const uint8_t *buf, long size) {
  memset(ps, 0, sizeof(*ps));
  ps->buf = buf;
  ps->size = size;
  if (size < 4 || buf[0] != 0xFF || buf[1] != 0xD8) {
    return jga_fail("Error, not a JPEG (invalid SOI marker).");
  }
  ps->pos = 2;
  for (;;) {
    int marker, rc = EXIT_SUCCESS;
    long len, end;
    if (!need(ps, 2)) return jga_fail("Error underflow reading marker.");
    if (ps->buf[ps->pos] != 0xFF) return jga_fail("Error, invalid JPEG syntax.");
    while (ps->pos < ps->size && ps->buf[ps->pos] == 0xFF) ps->pos++;  /* fill */
    if (!need(ps, 1)) return jga_fail("Error underflow reading marker.");
    marker = rd8(ps);
    if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) {
      continue;                                   /* standalone markers */
    }
    if (marker == 0xD9) return jga_fail("Error, EOI before any scan.");
    if (!need(ps, 2)) return jga_fail("Error reading past the end of file.");
    len = rd16(ps);
    end = ps->pos + len - 2;
    if (len < 2 || end > ps->size) return jga_fail("Error skipping past the end of file.");
    switch (marker) {
      case 0xDB : rc = parse_dqt(ps, end); break;
      case 0xC4 : rc = parse_dht(ps, end); break;
      case 0xC0 : rc = parse_sof0(ps, end); break;
      case 0xDD : {
        if (len != 4) return jga_fail("Error decoding DRI, unprocessed bytes.");
        ps->restart_interval = rd16(ps);
        break;
      }
      case 0xDA : {
        rc = parse_sos(ps, end);
        if (rc == EXIT_SUCCESS) ps->pos = end;
        return rc;
      }
      case 0xC1 : case 0xC2 : case 0xC3 : case 0xC5 : case 0xC6 : case 0xC7 :
      case 0xC9 : case 0xCA : case 0xCB : case 0xCD : case 0xCE : case 0xCF : {
        return jga_fail("Unsupported JPEG process (SOF%i); baseline only.", marker & 15);
      }
      default : break;                            /* APPn, COM, ... skipped */
    }
    if (rc != EXIT_SUCCESS) return rc;
    ps->pos = end;
  }
}

static void fill_header(const parser *ps, jpeg_header *h) {
  int i, hmax = 0, vmax = 0, nhmb, nvmb;
  memset(h, 0, sizeof(*h));
  h->bits = ps->bits;
  h->width = ps->width;
  h->height = ps->height;
  h->ncomps = ps->ncomps;
  h->restart_interval = ps->restart_interval;
  for (i = 0; i < NQUANT_MAX; i++) h->quant[i] = ps->quant[i];
  for (i = 0; i < ps->ncomps; i++) {
    if (ps->comp[i].hs > hmax) hmax = ps->comp[i].hs;
    if (ps->comp[i].vs > vmax) vmax = ps->comp[i].vs;
  }
  nhmb = (ps->width + 8*hmax - 1)/(8*hmax);
  nvmb = (ps->height + 8*vmax - 1)/(8*vmax);
  for (i = 0; i < ps->ncomps; i++) {
    h->comp[i].hsamp = ps->comp[i].hs;
    h->comp[i].vsamp = ps->comp[i].vs;
    h->comp[i].hblocks = nhmb*ps->comp[i].hs;
    h->comp[i].vblocks = nvmb*ps->comp[i].vs;
    h->comp[i].quant = &h->quant[ps->comp[i].tq];
  }
  if (ps->ncomps == 1) h->subsamp = JPEG_SUBSAMP_MONO;
  else {
    h->subsamp = (jpeg_subsamp)jga_subsamp_of(
     jga_ilog(ps->comp[0].hs) - jga_ilog(ps->comp[1].hs),
     jga_ilog(ps->comp[0].vs) - jga_ilog(ps->comp[1].vs), 3);
  }
}

JGA_EXPORT int jga_parse_header(const unsigned char *buf, int size,
 jpeg_header *header) {
  parser *ps = (parser *)malloc(sizeof(parser));
  int rc;
  if (!ps) return jga_fail("Out of memory");
  rc = parse_to_scan(ps, buf, size);
  if (rc == EXIT_SUCCESS) fill_header(ps, header);
  free(ps);
  return rc;
}

int jga_scan_describe(const unsigned char *buf, int size, jga_scan_desc *d) {
  parser *ps = (parser *)malloc(sizeof(parser));
  int rc, i;
  if (!ps) return jga_fail("Out of memory");
  rc = parse_to_scan(ps, buf, size);
  if (rc == EXIT_SUCCESS) {
    memset(d, 0, sizeof(*d));
    fill_header(ps, &d->header);
    d->scan_off = (int)ps->pos;
    for (i = 0; i < ps->ncomps; i++) {
      d->td[i] = ps->comp[i].td;
      d->ta[i] = ps->comp[i].ta;
    }
    for (i = 0; i < 8; i++) {
      d->dht_valid[i] = i < 4 ? ps->dc[i].valid : ps->ac[i - 4].valid;
      memcpy(d->dht_bits[i], ps->dht_bits[i], 16);
      memcpy(d->dht_vals[i], ps->dht_vals[i], 256);
    }
  }
  free(ps);
  return rc;
}

/* ---- bit reader --------------------------------------------------------- */

/* The scan is read one restart interval at a time from a CLEAN copy: the bytes up to the next
 * marker with the stuffed zeros (FF 00) and fill bytes (FF FF ..) taken out (load_interval), 280
 * zero bytes behind them.  The per-symbol path then refills its window without looking at what
 * it loads — no "is there an FF among the next eight bytes" test, no marker state — and running
 * off the data is found by arithmetic afterwards (bits consumed against bits there were).  The
 * copy costs a pass of memchr + memcpy over runs of ~256 bytes, 2 % of what the decode takes. */
typedef struct bitreader {
  uint64_t bits;       /* MSB-aligned window; the top nbits are accounted for, those below are
                          look-ahead (the same stream bits the next refill ORs in again) */
  int nbits;
  const uint8_t *p;    /* next byte of the clean interval to load */
  const uint8_t *base, *lim;   /* the clean interval; zeros behind it */
  /* where the interval came from */
  const uint8_t *raw, *raw_end;      /* raw: the marker that ended it (or raw_end) */
  int marker;          /* that marker's second byte, 0 = the buffer ended without one */
  uint8_t *clean;      /* the copy's buffer (raw_end - raw0 + CLEAN_PAD bytes) */
} bitreader;
#define CLEAN_PAD 280  /* a block takes at most 64 x 31 bits = 248 bytes, a refill reads 8 */

static inline __attribute__((always_inline)) uint64_t load_be64(const uint8_t *p) {
  uint64_t w;
  memcpy(&w, p, 8);
  return __builtin_bswap64(w);
}

/* Bits of the interval consumed so far exceed the bits it has: the decode ran into the padding. */
static inline __attribute__((always_inline)) int ran_past_end(const bitreader *br) {
  return 8*(br->p - br->base) - br->nbits > 8*(br->lim - br->base);
}

/* Clean copy of the entropy-coded bytes from `from` up to the next marker.  The rules are T.81
 * B.1.1.5 / F.1.2.3 as the reference's reader applies them (src/xjpeg.c:151-186): FF 00 is a data
 * byte FF, FF FF.. is fill before a marker (or before more data), FF xx anything else is a
 * marker — the interval ends ON its FF.  A lone FF as the buffer's last byte counts as the end
 * of the image. */
static void load_interval(bitreader *br, const uint8_t *from) {
  const uint8_t *p = from, *end = br->raw_end;
  uint8_t *o = br->clean;
  br->marker = 0;
  while (p < end) {
    const uint8_t *q = (const uint8_t *)memchr(p, 0xFF, (size_t)(end - p));
    if (!q) q = end;
    memcpy(o, p, (size_t)(q - p));
    o += q - p;
    p = q;
    if (p >= end) break;
    if (p + 1 >= end) { br->marker = 0xD9; break; }          /* FF, then nothing */
    if (p[1] == 0x00) { *o++ = 0xFF; p += 2; }
    else if (p[1] == 0xFF) p++;
    else { br->marker = p[1]; break; }
  }
  memset(o, 0, CLEAN_PAD);
  br->raw = p;
  br->base = br->p = br->clean;
  br->lim = o;
  br->bits = 0;
  br->nbits = 0;
}

/* After this at least 56 bits are accounted for (real or padding); at most 63. */
static inline __attribute__((always_inline)) void refill(bitreader *br) {
  br->bits |= load_be64(br->p) >> br->nbits;
  br->p += (63 - br->nbits) >> 3;
  br->nbits |= 56;
}

#define PEEK(br, n) ((unsigned)((br)->bits >> (64 - (n))))
#define SKIP(br, n) ((br)->bits <<= (n), (br)->nbits -= (n))

/* Decode one Huffman symbol; needs >= 16 bits in the window. */
static inline __attribute__((always_inline)) int huff_symbol(bitreader *br, const htab *h) {
  unsigned e = h->fast[PEEK(br, FAST_BITS)];
  unsigned code;
  int len;
  if (e) {
    SKIP(br, (e >> 8) & 255);
    return (int)(e >> 16);
  }
  code = PEEK(br, 16);
  len = FAST_BITS + 1;
  while (code >= h->maxcode[len]) len++;
  if (len > 16) return -1;
  SKIP(br, len);
  return h->sym[((int)(code >> (16 - len)) + h->delta[len]) & 255];
}

/* ---- scan --------------------------------------------------------------- */

typedef struct scan_out {
  int stage;
  short *coef;
  short *pack;
  long long pack_cap, nwords;
  long long plane_words[3];
  int *index;
} scan_out;

/* Sign extension of an s-bit magnitude (T.81 F.2.2.1 EXTEND), without a branch: values whose
 * top bit is clear are negative, v - (2^s - 1). */
static inline __attribute__((always_inline)) int extend_bits(unsigned v, int s) {
  return (int)v + (int)(((v >> (s - 1)) - 1u) & (unsigned)(1 - (1 << s)));
}

/* ---- two AC symbols per look-up ---------------------------------------------------------------
 * At ~5 bits per symbol (code + magnitude) the ten bits the code table is indexed with usually
 * hold TWO whole symbols — 66 % of the consecutive pairs of the bench's q90 frames, 54 % at q50.
 * The pair table gives, for the same ten bits, everything both leave behind:
 *     bits  0.. 7  tot   bits the symbols take (0: the first one does not fit — one-symbol path)
 *     bits  8..15  need  how far the zig-zag index moves (every run + 1), + 1 for an EOB: the step
 *                        is taken iff k + need <= 64 — an EOB is a symbol of THIS block only
 *                        while k < 64 (behind coefficient 63 the next bits are a DC code)
 *     bits 16..23  at1   where the first value goes relative to k   (64: nowhere — ZRL, EOB)
 *     bits 24..31  at2   the second one's                            (64: none)
 *     bits 32..47  v1, bits 48..62 v2 (15 bits, sign extended)       bit 63: the last symbol is an EOB
 * built by decoding the ten bits symbol by symbol (build_pairs), so a step through it is, for
 * any input, what two steps of the one-symbol path would have done — as long as the block has
 * room for both (k + need <= 64: in a valid stream nearly always; otherwise the one-symbol path goes
 * on and reports what it finds where it finds it).  Values are stored without a test: DEZZX
 * maps every position >= 64 to 0, the DC coefficient's place, which is written last.  (A block
 * put together in a local buffer with a spare slot and copied out at its end measured SLOWER on
 * the GPU box's EPYC, 4.6 -> 6.6 ms per q50 frame: 16-byte loads of a line that 2-byte stores
 * have just written; a select between the block and a scratch short became a branch.) */
#define PAIR_EOB ((uint64_t)1 << 63)
static const uint8_t DEZZX[136] = {
   0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
   0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
   0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
   0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
   0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
   0,  0,  0,  0,  0,  0,  0,  0
};

/* The same for the PACK stage (xjpeg.c:484-496, 513-519): what the symbols leave behind there is
 * one RLE word each, run << 12 | value & 0xfff, an EOB the word 0:
 *     bits 0..7 tot, 8..15 need (as above), 16..23 words (1 or 2), 32..47 word 1, 48..63 word 2
 * (one word: it is word 2 as well, so that both can be stored without a test). */
static void build_pack_pairs(const htab *ac, uint64_t *pair) {
  unsigned idx;
  for (idx = 0; idx < (1u << FAST_BITS); idx++) {
    const unsigned e1 = ac->fast[idx];
    const int tot1 = (int)(e1 & 255), rs1 = (int)(e1 >> 16), r1 = rs1 >> 4, s1 = rs1 & 15;
    int v1 = 0, need, tot = tot1, words = 1, eob = 0;
    unsigned w1, w2;
    pair[idx] = 0;
    if (!e1 || tot1 > FAST_BITS) continue;
    if (s1) v1 = extend_bits((idx >> (FAST_BITS - tot1)) & ((1u << s1) - 1u), s1);
    if (rs1 == 0) { eob = 1; need = 1; w1 = w2 = 0; }
    else {
      const unsigned idx2 = (idx << tot1) & ((1u << FAST_BITS) - 1u);
      const unsigned e2 = ac->fast[idx2];
      const int tot2 = (int)(e2 & 255), rs2 = (int)(e2 >> 16), r2 = rs2 >> 4, s2 = rs2 & 15;
      need = r1 + 1;
      w1 = w2 = (unsigned)((r1 << 12) | (v1 & 0xfff)) & 0xffffu;
      if (e2 && tot1 + tot2 <= FAST_BITS) {
        tot += tot2;
        words = 2;
        if (rs2 == 0) { eob = 1; need++; w2 = 0; }
        else {
          const int v2 = s2 ? extend_bits((idx2 >> (FAST_BITS - tot2)) & ((1u << s2) - 1u), s2) : 0;
          need += r2 + 1;
          w2 = (unsigned)((r2 << 12) | (v2 & 0xfff)) & 0xffffu;
        }
      }
    }
    pair[idx] = (uint64_t)tot | (uint64_t)need << 8 | (uint64_t)words << 16 | (uint64_t)eob << 24
     | (uint64_t)w1 << 32 | (uint64_t)w2 << 48;
  }
}

static void build_pairs(const htab *ac, uint64_t *pair) {
  unsigned idx;
  for (idx = 0; idx < (1u << FAST_BITS); idx++) {
    const unsigned e1 = ac->fast[idx];
    const int tot1 = (int)(e1 & 255), len1 = (int)((e1 >> 8) & 255);
    const int rs1 = (int)(e1 >> 16), r1 = rs1 >> 4, s1 = rs1 & 15;
    uint64_t x;
    int v1 = 0, v2 = 0, adv, at1 = 64, at2 = 64, tot = tot1, eob = 0;
    pair[idx] = 0;
    if (!e1 || tot1 > FAST_BITS) continue;             /* a longer code, or magnitude bits beyond the index */
    if (s1) v1 = extend_bits((idx >> (FAST_BITS - tot1)) & ((1u << s1) - 1u), s1);
    (void)len1;
    if (rs1 == 0) { eob = 1; adv = 0; }
    else {
      const unsigned idx2 = (idx << tot1) & ((1u << FAST_BITS) - 1u);     /* what follows, zeros behind it */
      const unsigned e2 = ac->fast[idx2];
      const int tot2 = (int)(e2 & 255), rs2 = (int)(e2 >> 16), r2 = rs2 >> 4, s2 = rs2 & 15;
      adv = r1 + 1;
      if (s1) at1 = r1;
      if (e2 && tot1 + tot2 <= FAST_BITS) {              /* the second symbol lies inside the index too */
        tot += tot2;
        if (rs2 == 0) eob = 1;
        else {
          if (s2) {
            v2 = extend_bits((idx2 >> (FAST_BITS - tot2)) & ((1u << s2) - 1u), s2);
            at2 = adv + r2;
          }
          adv += r2 + 1;
        }
      }
    }
    x = (uint64_t)tot | (uint64_t)(adv + eob) << 8 | (uint64_t)at1 << 16 | (uint64_t)at2 << 24
     | (uint64_t)(uint16_t)v1 << 32 | (uint64_t)((uint16_t)v2 & 0x7fffu) << 48;
    pair[idx] = eob ? x | PAIR_EOB : x;
  }
}

/* One block.  `blk` receives 64 natural-order shorts (QUANT/DCT stages).
 * One path per symbol whatever its length: a table look-up for the code (codes of up to
 * FAST_BITS bits — nearly all — in one step), then the magnitude bits straight from the
 * window.  The branches left are the ones that go the same way almost every time (window
 * low, code longer than FAST_BITS, end of block): the earlier version chose between a
 * combined code+value table and this path on every symbol, a coin toss on busy images. */
static inline __attribute__((always_inline)) int decode_block(bitreader *br, const htab *dc, const htab *ac,
 const uint64_t *pair, const unsigned short *q, short *pred, short *blk, scan_out *so, int stage) {
  int s, k;
  /* (a block reads at most 248 + 8 bytes past this: inside the padding) */
  if (br->p > br->lim + 8) return jga_fail("Error, entropy data ended early.");
  refill(br);                                      /* >= 56 bits: DC code + magnitude <= 31 */
  s = huff_symbol(br, dc);
  if (s < 0 || s > 15) return jga_fail("Error invalid DC code.");
  if (s) {
    *pred = (short)(*pred + extend_bits(PEEK(br, s), s));
    SKIP(br, s);
  }
  if (stage == JGA_STAGE_PACK) {
    if (so->nwords + 64 > so->pack_cap) return jga_fail("Error PACK buffer too small.");   /* a block is at most 1 + 63 words */
    so->pack[so->nwords++] = (short)(*pred & 0xfff);
  }
  else memset(blk, 0, 64*sizeof(short));
  /* Two steps per refill: it leaves >= 56 bits, a step through a table takes at most
     FAST_BITS + 15 = 25 and any symbol at most 31 — so the second step has its 25 whatever the
     first was, and a long code (rare) refills for itself. */
#define AC_STEP() do { \
    unsigned e; \
    int rs, r, v; \
    if (stage == JGA_STAGE_PACK) { \
      const uint64_t x = pair[PEEK(br, FAST_BITS)]; \
      const int need = (int)((x >> 8) & 255); \
      if (__builtin_expect((x & 255) != 0 && k + need <= 64, 1)) { \
        const int words = (int)((x >> 16) & 255); \
        so->pack[so->nwords] = (short)(x >> 32); \
        so->pack[so->nwords + words - 1] = (short)(x >> 48); \
        so->nwords += words; \
        SKIP(br, x & 255); \
        if (x & (1u << 24)) goto block_done; \
        k += need; \
        break; \
      } \
    } \
    else { \
      const uint64_t x = pair[PEEK(br, FAST_BITS)]; \
      const int need = (int)((x >> 8) & 255); \
      if (__builtin_expect((x & 255) != 0 && k + need <= 64, 1)) { \
        const int n1 = DEZZX[k + (int)((x >> 16) & 255)], n2 = DEZZX[k + (int)((x >> 24) & 255)]; \
        const int v1 = (int16_t)(x >> 32), v2 = (int16_t)((int64_t)(x << 1) >> 49); \
        blk[n1] = stage == JGA_STAGE_DCT ? (short)((short)v1*q[n1]) : (short)v1; \
        blk[n2] = stage == JGA_STAGE_DCT ? (short)((short)v2*q[n2]) : (short)v2; \
        SKIP(br, x & 255); \
        if (x & PAIR_EOB) goto block_done; \
        k += need; \
        break; \
      } \
    } \
    e = ac->fast[PEEK(br, FAST_BITS)]; \
    if (__builtin_expect(e != 0, 1)) { \
      /* code and magnitude leave the window in ONE shift (the next look-up waits for nothing \
         else); the value is read from the bits as they were */ \
      const uint64_t w = br->bits << ((e >> 8) & 255); \
      rs = (int)(e >> 16); \
      r = rs >> 4; \
      s = rs & 15; \
      SKIP(br, e & 255); \
      /* (EXTEND from the window itself: its top bit is the magnitude's first, 1 = positive) */ \
      v = s ? (int)(w >> (64 - s)) + ((int)~((int64_t)w >> 63) & (1 - (1 << s))) : 0; \
    } \
    else { \
      refill(br); \
      rs = huff_symbol(br, ac); \
      if (rs < 0) return jga_fail("Error invalid AC code."); \
      r = rs >> 4; \
      s = rs & 15; \
      v = 0; \
      if (s) { \
        v = extend_bits(PEEK(br, s), s); \
        SKIP(br, s); \
      } \
    } \
    if (__builtin_expect(s == 0, 0)) { \
      if (rs == 0) {                               /* EOB */ \
        if (stage == JGA_STAGE_PACK) so->pack[so->nwords++] = 0; \
        goto block_done; \
      } \
      v = 0;                                       /* ZRL (or any run without a value): r + 1 zeros, xjpeg.c:507-508 */ \
    } \
    k += r; \
    if (k > 63) return jga_fail("Error indexing outside block."); \
    if (stage == JGA_STAGE_PACK) { \
      so->pack[so->nwords++] = (short)((r << 12) | (v & 0xfff)); \
    } \
    else if (s) { \
      const int n = DEZZ[k]; \
      blk[n] = stage == JGA_STAGE_DCT ? (short)((short)v*q[n]) : (short)v; \
    } \
    k++; \
  } while (0)
  for (k = 1; k < 64;) {
    refill(br);
    AC_STEP();
    if (k >= 64) break;
    AC_STEP();
  }
block_done:
#undef AC_STEP
  /* (last: the pair path's stores that go nowhere went here) */
  if (stage != JGA_STAGE_PACK) blk[0] = stage == JGA_STAGE_DCT ? (short)(*pred*q[0]) : *pred;
  return EXIT_SUCCESS;
}

static int next_restart(bitreader *br, int expect) {
  if (ran_past_end(br)) return jga_fail("Error, entropy data ended early.");
  /* (bytes of the interval the decode did not need are tolerated, as stray bytes before the marker) */
  if (!br->marker) return jga_fail("Error, expected to find marker.");
  if (br->marker != 0xD0 + (expect & 7)) {
    if (br->marker >= 0xD0 && br->marker <= 0xD7) {
      return jga_fail("Error invalid RST counter in marker.");
    }
    return jga_fail("Error, unknown marker found in scan.");
  }
  load_interval(br, br->raw + 2);
  return EXIT_SUCCESS;
}

static int check_geom(const parser *ps, const jga_geom *g) {
  jpeg_header h;
  jga_geom mine;
  fill_header(ps, &h);
  if (jga_geom_from_header(&mine, &h) != EXIT_SUCCESS) return EXIT_FAILURE;
  if (mine.width != g->width || mine.height != g->height
   || mine.nplanes != g->nplanes || mine.coef_shorts != g->coef_shorts
   || mine.w0 != g->w0 || mine.nhmb != g->nhmb || mine.nvmb != g->nvmb) {
    return jga_fail("Error, image geometry does not match the JPEG headers.");
  }
  return EXIT_SUCCESS;
}

#define MCU_SLOTS_MAX 48          /* 4 x 4 blocks of each of three components */
typedef struct mcu_slot {
  const htab *dc, *ac;
  const uint64_t *pair;            /* the AC table's pair table (QUANT / DCT stages) */
  const unsigned short *q;         /* 65 entries, the last one 0 */
  int comp, sbx, sby, step;
  long long at;                    /* offset of the slot's block in the planes (PACK: in the index) */
} mcu_slot;

static inline __attribute__((always_inline)) int decode_scan(parser *ps, const jga_geom *g, scan_out *so,
 const int stage, uint8_t *clean) {
  bitreader br;
  short pred[3] = {0, 0, 0};
  short scratch[64];                               /* (PACK: decode_block's unused block) */
  unsigned short qx[3][72];
  uint64_t pairs[4][1 << FAST_BITS];               /* built for the AC tables the scan uses */
  int pairs_built[4] = {0, 0, 0, 0};
  int mbx, mby, i, sbx, sby, nslots = 0;
  long mcus = 0, total = (long)g->nhmb*g->nvmb;
  int rst = 0, to_restart;
  long long index_base[3];
  mcu_slot slots[MCU_SLOTS_MAX];
  memset(&br, 0, sizeof(br));
  br.clean = clean;
  br.raw_end = ps->buf + ps->size;
  load_interval(&br, ps->buf + ps->pos);
  index_base[0] = 0;
  for (i = 1; i < 3; i++) {
    index_base[i] = index_base[i - 1]
     + (long long)(g->plane[i - 1].hblocks << g->plane[i - 1].xdec)
     *g->plane[i - 1].cstride;
  }
  /* the blocks of an MCU in scan order (xjpeg.c:461-472), each with its tables; where a slot's
     block lies is worked out once per row of MCUs and stepped from MCU to MCU */
  for (i = 0; i < ps->ncomps; i++) {
    const comp_info *c = &ps->comp[i];
    memcpy(qx[i], ps->quant[c->tq].tbl, 64*sizeof(unsigned short));
    memset(qx[i] + 64, 0, 8*sizeof(unsigned short));
    if (!pairs_built[c->ta]) {
      if (stage == JGA_STAGE_PACK) build_pack_pairs(&ps->ac[c->ta], pairs[c->ta]);
      else build_pairs(&ps->ac[c->ta], pairs[c->ta]);
      pairs_built[c->ta] = 1;
    }
    for (sby = 0; sby < c->vs; sby++) {
      for (sbx = 0; sbx < c->hs; sbx++) {
        if (nslots >= MCU_SLOTS_MAX) return jga_fail("Unsupported sampling: more than %d blocks per MCU", MCU_SLOTS_MAX);
        slots[nslots].dc = &ps->dc[c->td];
        slots[nslots].ac = &ps->ac[c->ta];
        slots[nslots].pair = pairs[c->ta];
        slots[nslots].q = qx[i];
        slots[nslots].comp = i;
        slots[nslots].sbx = sbx;
        slots[nslots].sby = sby;
        nslots++;
      }
    }
  }
  to_restart = ps->restart_interval;
  for (mby = 0; mby < g->nvmb; mby++) {
    for (i = 0; i < nslots; i++) {
      const comp_info *c = &ps->comp[slots[i].comp];
      const int by = mby*c->vs + slots[i].sby;
      slots[i].at = stage == JGA_STAGE_PACK
       ? index_base[slots[i].comp] + (long long)by*g->plane[slots[i].comp].hblocks + slots[i].sbx
       : jga_block_offset(g, slots[i].comp, slots[i].sbx, by);
      slots[i].step = stage == JGA_STAGE_PACK ? c->hs : c->hs*64;
    }
    for (mbx = 0; mbx < g->nhmb; mbx++) {
      for (i = 0; i < nslots; i++) {
        mcu_slot *sl = &slots[i];
        short *blk = scratch;
        const long long w0 = so->nwords;
        if (stage == JGA_STAGE_PACK) so->index[sl->at] = (int)so->nwords;
        else blk = so->coef + sl->at;
        if (decode_block(&br, sl->dc, sl->ac, sl->pair, sl->q, &pred[sl->comp], blk, so, stage) != EXIT_SUCCESS) {
          return EXIT_FAILURE;
        }
        sl->at += sl->step;
        if (stage == JGA_STAGE_PACK) so->plane_words[sl->comp] += so->nwords - w0;
      }
      mcus++;
      if (ps->restart_interval && --to_restart == 0 && mcus < total) {
        if (next_restart(&br, rst++) != EXIT_SUCCESS) return EXIT_FAILURE;
        pred[0] = pred[1] = pred[2] = 0;
        to_restart = ps->restart_interval;
      }
    }
  }
  if (ran_past_end(&br)) return jga_fail("Error, entropy data ended early.");
  return EXIT_SUCCESS;
}

/* The scan loop once per stage (a literal stage: the per-symbol path holds no stage test) and once
 * more per stage for CPUs with BMI2 — shifts by a register that need not be %cl, which the
 * per-symbol path is made of (a third fewer instructions). */
#define SCAN_VARIANT(name, attr, stage) \
  attr static int name(parser *ps, const jga_geom *g, scan_out *so, uint8_t *clean) { \
    return decode_scan(ps, g, so, stage, clean); \
  }
SCAN_VARIANT(scan_pack, , JGA_STAGE_PACK)
SCAN_VARIANT(scan_quant, , JGA_STAGE_QUANT)
SCAN_VARIANT(scan_dct, , JGA_STAGE_DCT)
#if defined(__x86_64__) && !defined(JGA_ENTROPY_NO_BMI2)   /* (NO_BMI2: the differential fuzz runs the plain variants on a CPU that has BMI2) */
#define JGA_ENTROPY_BMI2 (1)
SCAN_VARIANT(scan_pack_bmi2, __attribute__((target("bmi,bmi2"))), JGA_STAGE_PACK)
SCAN_VARIANT(scan_quant_bmi2, __attribute__((target("bmi,bmi2"))), JGA_STAGE_QUANT)
SCAN_VARIANT(scan_dct_bmi2, __attribute__((target("bmi,bmi2"))), JGA_STAGE_DCT)
#endif

static int run_decode(const unsigned char *buf, int size, const jga_geom *g,
 scan_out *so) {
  parser *ps = (parser *)malloc(sizeof(parser));
  int rc;
  if (!ps) return jga_fail("Out of memory");
  rc = parse_to_scan(ps, buf, size);
  if (rc == EXIT_SUCCESS) rc = check_geom(ps, g);
  if (rc == EXIT_SUCCESS) {
    /* room for the clean copy of the longest interval there can be: everything behind SOS */
    uint8_t *clean = (uint8_t *)malloc((size_t)(ps->size - ps->pos) + CLEAN_PAD);
    if (!clean) rc = jga_fail("Out of memory");
#ifdef JGA_ENTROPY_BMI2
    else if (__builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi")) switch (so->stage) {
      case JGA_STAGE_PACK : rc = scan_pack_bmi2(ps, g, so, clean); break;
      case JGA_STAGE_QUANT : rc = scan_quant_bmi2(ps, g, so, clean); break;
      default : rc = scan_dct_bmi2(ps, g, so, clean); break;
    }
#endif
    else switch (so->stage) {
      case JGA_STAGE_PACK : rc = scan_pack(ps, g, so, clean); break;
      case JGA_STAGE_QUANT : rc = scan_quant(ps, g, so, clean); break;
      default : rc = scan_dct(ps, g, so, clean); break;
    }
    free(clean);
  }
  free(ps);
  return rc;
}

JGA_EXPORT int jga_entropy_decode(const unsigned char *buf, int size,
 const jga_geom *g, short *coef, int dequant) {
  scan_out so;
  memset(&so, 0, sizeof(so));
  so.stage = dequant ? JGA_STAGE_DCT : JGA_STAGE_QUANT;
  so.coef = coef;
  return run_decode(buf, size, g, &so);
}

JGA_EXPORT int jga_entropy_decode_pack(const unsigned char *buf, int size,
 const jga_geom *g, short *pack, long long pack_cap, int *index,
 long long *nwords, long long *plane_words) {
  scan_out so;
  int rc;
  memset(&so, 0, sizeof(so));
  so.stage = JGA_STAGE_PACK;
  so.pack = pack;
  so.pack_cap = pack_cap;
  so.index = index;
  rc = run_decode(buf, size, g, &so);
  if (nwords) *nwords = so.nwords;
  if (plane_words) memcpy(plane_words, so.plane_words, sizeof(so.plane_words));
  return rc;
}
