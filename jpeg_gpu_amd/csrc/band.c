/* band.c — one frame cut into bands of MCU rows that decode independently ("an 8K
 * image could be band-split across GPUs by MCU rows").
 *
 * What makes it possible is the restart interval (reference src/xjpeg.c:593-629, DRI parse
 * 412-420): at every RSTn the entropy coder is byte aligned and the DC predictors are back at 0
 * (xjpeg.c:612-618), so the MCUs from one marker on depend on nothing before it.  A band is a run of
 * whole MCU rows that starts at such a marker.  jga_band_plan() finds them (one pass over the
 * entropy-coded bytes looking for FF Dn), jga_band_file() writes band b as a baseline JPEG FILE of
 * its own: the frame's marker segments as they stand, the SOF0 height replaced by the band's rows,
 * the band's entropy-coded bytes with the RSTn counters renumbered from 0 (decoders, this one
 * included, check them: entropy.c next_restart, xjpeg.c:600-611), EOI.  Any decode entry point of
 * this library — and the reference's — then decodes the band like any other file; its pixels are
 * rows [y0, y0 + rows) of the frame, bit for bit (the block decode works block by block and the
 * chroma upsample replicates samples inside an MCU: nothing reads across an MCU row).
 *
 * Host code: no device work here.  One process per GPU takes band `rank` of `world`
 * (jpeg_gpu_amd/shard.py: band_of_rank); no collective — the bands stay where they were decoded. */
#include <stdlib.h>
#include <string.h>
#include "jga_internal.h"

typedef struct band_walk {
  long sof_height_off;               /* offset of SOF0's 16-bit height */
  long scan0;                        /* first entropy-coded byte */
  int width, height, hmax, vmax, ri;
} band_walk;

/* Marker segments up to SOS — only what the cut needs; jga_parse_header() has validated the rest. */
static int walk(const unsigned char *f, long size, band_walk *w) {
  long pos = 2;
  memset(w, 0, sizeof(*w));
  for (;;) {
    int m;
    long len;
    if (pos + 4 > size || f[pos] != 0xFF) return jga_fail("Error, invalid JPEG syntax.");
    while (pos < size && f[pos] == 0xFF) pos++;
    if (pos >= size) return jga_fail("Error underflow reading marker.");
    m = f[pos++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (pos + 2 > size) return jga_fail("Error reading past the end of file.");
    len = (f[pos] << 8) | f[pos + 1];
    if (len < 2 || pos + len > size) return jga_fail("Error skipping past the end of file.");
    if (m == 0xC0) {
      int i, n;
      if (len < 8) return jga_fail("Error SOF needs at least 9 bytes");
      w->sof_height_off = pos + 3;
      w->height = (f[pos + 3] << 8) | f[pos + 4];
      w->width = (f[pos + 5] << 8) | f[pos + 6];
      n = f[pos + 7];
      if (len != 8 + 3*n) return jga_fail("Error decoding SOF, wrong length.");
      for (i = 0; i < n; i++) {
        const int b = f[pos + 8 + 3*i + 1];
        if ((b >> 4) > w->hmax) w->hmax = b >> 4;
        if ((b & 15) > w->vmax) w->vmax = b & 15;
      }
    }
    else if (m == 0xDD && len == 4) w->ri = (f[pos + 2] << 8) | f[pos + 3];
    pos += len;
    if (m == 0xDA) { w->scan0 = pos; return EXIT_SUCCESS; }
  }
}

/* Next position >= pos (< size - 1) of an FF that is followed by something other than 00 — a marker
 * (or fill) inside entropy-coded data, where one byte in 256 is a stuffed FF 00 — or -1.  The
 * stuffed ones are dropped 32 bytes at a time: the pass over an 8K frame's 12 MB is bound by
 * reading them (0.5 ms), not by 48 000 calls of memchr (1.7 ms). */
#if defined(__x86_64__)
__attribute__((target("avx2"))) static long next_marker_avx2(const unsigned char *f, long pos, long size) {
  typedef long long v4 __attribute__((vector_size(32), aligned(1)));
  typedef char v32 __attribute__((vector_size(32)));
  while (pos + 33 <= size) {
    v4 a, b;
    v32 hit;
    unsigned mask;
    __builtin_memcpy(&a, f + pos, 32);
    __builtin_memcpy(&b, f + pos + 1, 32);
    hit = ((v32)a == (v32)(__builtin_ia32_pcmpeqb256((v32)a, (v32)a))) & ((v32)b != (v32){0});
    mask = (unsigned)__builtin_ia32_pmovmskb256(hit);
    if (mask) return pos + __builtin_ctz(mask);
    pos += 32;
  }
  for (; pos + 1 < size; pos++) if (f[pos] == 0xFF && f[pos + 1] != 0x00) return pos;
  return -1;
}
#endif
static long next_marker(const unsigned char *f, long pos, long size) {
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2")) return next_marker_avx2(f, pos, size);
#endif
  while (pos + 1 < size) {
    const unsigned char *p = (const unsigned char *)memchr(f + pos, 0xFF, (size_t)(size - 1 - pos));
    if (!p) return -1;
    pos = (long)(p - f);
    if (f[pos + 1] != 0x00) return pos;
    pos += 2;
  }
  return -1;
}

static long gcd_l(long a, long b) { while (b) { const long t = a % b; a = b; b = t; } return a; }

JGA_EXPORT int jga_band_plan(const unsigned char *file, long size, int count, jga_band *bands) {
  jpeg_header h;
  band_walk w;
  long nhmb, nvmb, step, groups, nint, i, pos, found;
  long *at;                          /* file offset of the first byte of interval i*per_group... see below */
  int b, made;
  if (!file || !bands || count < 1) return -jga_fail("jga_band_plan: bad arguments");
  if (size > 0x7fffffffL) return -jga_fail("jga_band_plan: file too large");
  if (jga_parse_header(file, (int)size, &h) != EXIT_SUCCESS) return -EXIT_FAILURE;
  if (walk(file, size, &w) != EXIT_SUCCESS) return -EXIT_FAILURE;
  if (w.hmax < 1 || w.vmax < 1) return -jga_fail("jga_band_plan: no frame header");
  nhmb = (w.width + 8*w.hmax - 1)/(8*w.hmax);
  nvmb = (w.height + 8*w.vmax - 1)/(8*w.vmax);
  /* MCU rows at which an interval begins: row*nhmb a multiple of the interval */
  if (w.ri <= 0) { step = nvmb; }                       /* no restart markers: the frame is one band */
  else step = w.ri/gcd_l(w.ri, nhmb);
  groups = (nvmb + step - 1)/step;                      /* runs of MCU rows that can stand alone */
  made = count < groups ? count : (int)groups;
  /* file offsets of the first byte of every group of rows: one pass over the scan */
  at = (long *)malloc((size_t)(groups + 1)*sizeof(long));
  if (!at) return -jga_fail("Out of memory");
  at[0] = w.scan0;
  nint = w.ri > 0 ? (nhmb*nvmb + w.ri - 1)/w.ri : 1;     /* intervals of the frame */
  found = 0;                                            /* markers seen = intervals ended */
  pos = w.scan0;
  {
    const long per_group = w.ri > 0 ? step*nhmb/w.ri : 1;   /* intervals per group of rows */
    long end = size;
    while ((pos = next_marker(file, pos, size)) >= 0) {
      const int m = file[pos + 1];
      if (m == 0xFF) { pos += 1; continue; }            /* fill */
      if (m >= 0xD0 && m <= 0xD7) {
        found++;
        if (found % per_group == 0 && found/per_group <= groups) at[found/per_group] = pos + 2;
        pos += 2;
        continue;
      }
      end = pos;                                        /* EOI or anything else: the scan ends here */
      break;
    }
    if (found + 1 < nint) { free(at); return -jga_fail("jga_band_plan: %ld restart markers, the frame needs %ld", found, nint - 1); }
    at[groups] = end;
    for (i = 1; i < groups; i++) {
      if (at[i] <= at[i - 1]) { free(at); return -jga_fail("jga_band_plan: restart markers out of place"); }
    }
    /* groups -> bands, sizes differing by at most one group */
    {
      long g0 = 0;
      for (b = 0; b < made; b++) {
        const long n = groups/made + (b < groups % made ? 1 : 0);
        jga_band *d = bands + b;
        long r0 = g0*step, r1 = (g0 + n)*step;
        if (r1 > nvmb) r1 = nvmb;
        memset(d, 0, sizeof(*d));
        d->index = b;
        d->count = made;
        d->mcu_row0 = (int)r0;
        d->mcu_rows = (int)(r1 - r0);
        d->y0 = (int)(r0*8*w.vmax);
        d->rows = (int)((r1 == nvmb ? w.height : r1*8*w.vmax) - d->y0);
        d->scan_off = at[g0];
        /* up to, not including, the marker that ends its last interval (or the scan's end) */
        d->scan_bytes = (g0 + n == groups ? at[groups] : at[g0 + n] - 2) - at[g0];
        d->first_interval = (int)(g0*per_group);
        g0 += n;
      }
    }
  }
  free(at);
  return made;
}

JGA_EXPORT long jga_band_file(const unsigned char *file, long size, const jga_band *band,
 unsigned char *out, long cap) {
  band_walk w;
  long need, o, pos, end;
  int next = 0;
  if (!file || !band) return -jga_fail("jga_band_file: bad arguments");
  if (size > 0x7fffffffL) return -jga_fail("jga_band_file: file too large");
  /* (the same gate as jga_band_plan: a band is only ever cut out of a file the header parser accepts) */
  {
    jpeg_header h;
    if (jga_parse_header(file, (int)size, &h) != EXIT_SUCCESS) return -EXIT_FAILURE;
  }
  if (walk(file, size, &w) != EXIT_SUCCESS) return -EXIT_FAILURE;
  /* no SOF0 seen: sof_height_off would still be 0 and the height would land on the SOI bytes */
  if (w.hmax < 1 || w.vmax < 1 || w.sof_height_off < 4) return -jga_fail("jga_band_file: no frame header");
  if (band->scan_off < w.scan0 || band->scan_bytes < 0 || band->scan_off + band->scan_bytes > size
   || band->rows < 1 || band->rows > 65535) return -jga_fail("jga_band_file: band outside the file");
  /* a band of THIS file begins at the scan's first byte or right behind an RSTn, and is no taller than the frame
   * (a jga_band planned on another file can pass the range test above by accident) */
  if (band->scan_off > w.scan0 && !(file[band->scan_off - 2] == 0xFF && file[band->scan_off - 1] >= 0xD0
   && file[band->scan_off - 1] <= 0xD7)) return -jga_fail("jga_band_file: the band does not start at a restart marker of this file");
  if (band->y0 < 0 || (long)band->y0 + band->rows > w.height) return -jga_fail("jga_band_file: band outside the frame");
  need = w.scan0 + band->scan_bytes + 2;
  if (!out) return need;
  if (cap < need) return -jga_fail("jga_band_file: %ld bytes needed, %ld given", need, cap);
  memcpy(out, file, (size_t)w.scan0);
  out[w.sof_height_off] = (unsigned char)(band->rows >> 8);
  out[w.sof_height_off + 1] = (unsigned char)(band->rows & 255);
  o = w.scan0;
  memcpy(out + o, file + band->scan_off, (size_t)band->scan_bytes);
  /* RSTn counters run from 0 in every file */
  if (band->first_interval & 7) {
    pos = o;
    end = o + band->scan_bytes;
    while ((pos = next_marker(out, pos, end)) >= 0) {
      if (out[pos + 1] >= 0xD0 && out[pos + 1] <= 0xD7) { out[pos + 1] = (unsigned char)(0xD0 + (next++ & 7)); pos += 2; }
      else pos += 1;
    }
  }
  o += band->scan_bytes;
  out[o++] = 0xFF;
  out[o++] = 0xD9;
  return o;
}
