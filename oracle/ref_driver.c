/* ref_driver.c — thin driver over the REFERENCE's own sources.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as oracle.c).  This file contains no
 * reference code: it is compiled together with the reference's C files where
 * they lie under /root/reference/src (see oracle/Makefile) into
 * oracle/_ref/libjpeggpu_ref.so, and only calls their public functions:
 *   glj_real_idct8x8 ............ src/dct.h
 *   xjpeg_init / xjpeg_decode_header / xjpeg_decode_image ... src/xjpeg.h
 *   image_init / image_clear .... src/image.h
 * The reference's plugin glue (src/jpeg_wrap.c) needs <jpeglib.h>, absent in
 * this image, so the header copy-out it performs (jpeg_wrap.c:263-319) is
 * re-stated here — that is the only logic in this file.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dct.h"
#include "image.h"
#include "internal.h"
#include "xjpeg.h"

#define REF_API __attribute__((visibility("default")))

typedef struct ref_info {
  int width, height, ncomps, restart_interval;
  int nhmb, nvmb;
  long long coef_shorts;
  int hsamp[3], vsamp[3], hblocks[3], vblocks[3], xdec[3], ydec[3];
  int cstride[3], tq[3];
  long long coef_off[3];
  unsigned short quant[3][64];
} ref_info;

REF_API void ref_idct8x8_blocks(short *out, const short *in, long n) {
  long b;
  for (b = 0; b < n; b++) {
    memcpy(out + 64*b, in + 64*b, 64*sizeof(short));
    /* in place, as the reference calls it (src/xjpeg.c:569) */
    glj_real_idct8x8(out + 64*b, 8, out + 64*b, 8);
  }
}

static int fill_header(xjpeg_decode_ctx *ctx, jpeg_header *h) {
  int i;
  if (ctx->error || !ctx->frame.valid) return 1;
  if (ctx->frame.ncomps != 1 && ctx->frame.ncomps != 3) return 1;
  memset(h, 0, sizeof(*h));
  h->width = ctx->frame.width;
  h->height = ctx->frame.height;
  h->bits = ctx->frame.bits;
  h->ncomps = ctx->frame.ncomps;
  h->restart_interval = ctx->restart_interval;
  for (i = 0; i < NQUANT_MAX; i++) {
    h->quant[i].valid = ctx->quant[i].valid;
    h->quant[i].bits = ctx->quant[i].bits;
    memcpy(h->quant[i].tbl, ctx->quant[i].tbl, sizeof(h->quant[i].tbl));
  }
  for (i = 0; i < h->ncomps; i++) {
    h->comp[i].hsamp = ctx->frame.comp[i].hsamp;
    h->comp[i].vsamp = ctx->frame.comp[i].vsamp;
    h->comp[i].hblocks = ctx->frame.nhmb*ctx->frame.comp[i].hsamp;
    h->comp[i].vblocks = ctx->frame.nvmb*ctx->frame.comp[i].vsamp;
    h->comp[i].quant = &h->quant[ctx->frame.comp[i].tq];
  }
  return 0;
}

static void export_info(const xjpeg_decode_ctx *ctx, const jpeg_header *h,
 const image *img, ref_info *o) {
  int i;
  long long total = 0;
  memset(o, 0, sizeof(*o));
  o->width = h->width; o->height = h->height; o->ncomps = h->ncomps;
  o->restart_interval = h->restart_interval;
  o->nhmb = ctx->frame.nhmb; o->nvmb = ctx->frame.nvmb;
  for (i = 0; i < h->ncomps; i++) {
    const image_plane *p = &img->plane[i];
    o->hsamp[i] = h->comp[i].hsamp; o->vsamp[i] = h->comp[i].vsamp;
    o->hblocks[i] = h->comp[i].hblocks; o->vblocks[i] = h->comp[i].vblocks;
    o->xdec[i] = p->xdec; o->ydec[i] = p->ydec;
    o->cstride[i] = p->cstride; o->tq[i] = ctx->frame.comp[i].tq;
    o->coef_off[i] = p->coef - img->coef;
    memcpy(o->quant[i], h->comp[i].quant->tbl, sizeof(o->quant[i]));
    total += ((long long)p->width << (p->xdec + 3))*p->cstride;
  }
  o->coef_shorts = total;
}

/* out: 0 pack, 1 quant, 2 dct, 3 yuv (xjpeg_decode_out, src/xjpeg.h:137-143).
 * Any of coef / p0..p2 / index may be NULL.  `nwords` receives the number of
 * PACK words written (sum of plane->packed). */
REF_API int ref_decode(const unsigned char *buf, int size, int out,
 short *coef, unsigned char *p0, unsigned char *p1, unsigned char *p2,
 int *index, long long *nwords, ref_info *info) {
  xjpeg_decode_ctx ctx;
  jpeg_header h;
  image img;
  unsigned char *planes[3];
  int i;
  planes[0] = p0; planes[1] = p1; planes[2] = p2;
  xjpeg_init(&ctx, buf, size);
  xjpeg_decode_header(&ctx);
  if (fill_header(&ctx, &h)) return 1;
  if (image_init(&img, &h) != EXIT_SUCCESS) return 1;
  image_zero(&img);
  if (out >= 0) xjpeg_decode_image(&ctx, &img, (xjpeg_decode_out)out);
  if (info) export_info(&ctx, &h, &img, info);
  if (out >= 0 && !ctx.error) {
    long long total = 0, blocks = 0, words = 0;
    for (i = 0; i < img.nplanes; i++) {
      image_plane *p = &img.plane[i];
      total += ((long long)p->width << (p->xdec + 3))*p->cstride;
      blocks += (long long)((p->width >> 3) << p->xdec)*p->cstride;
      words += p->packed;
      if (planes[i]) memcpy(planes[i], p->data, (size_t)p->ystride*p->height);
    }
    if (coef) memcpy(coef, img.coef, (size_t)total*sizeof(short));
    if (index) memcpy(index, img.index, (size_t)blocks*sizeof(int));
    if (nwords) *nwords = words;
  }
  i = ctx.error != NULL;
  image_clear(&img);
  return i;
}

/* The reference's steady-state loop for `-i xjpeg -o yuv` (src/jpeg_gpu.c:1215-1237 with
 * XJPEG_DECODE_CTX_VTBL, src/jpeg_wrap.c:321-358): buffers made once, then per frame
 * decode_reset (= xjpeg_init) -> decode_header (xjpeg_decode_header + the header copy-out)
 * -> decode_image(YUV).  bench.py's cpu_baseline leg runs one of these per host core. */
REF_API int ref_frames_yuv(const unsigned char *buf, int size, int frames) {
  xjpeg_decode_ctx ctx;
  jpeg_header h;
  image img;
  int f, bad = 0;
  xjpeg_init(&ctx, buf, size);
  xjpeg_decode_header(&ctx);
  if (fill_header(&ctx, &h)) return 1;
  if (image_init(&img, &h) != EXIT_SUCCESS) return 1;
  image_zero(&img);
  for (f = 0; f < frames && !bad; f++) {
    xjpeg_init(&ctx, buf, size);
    xjpeg_decode_header(&ctx);
    bad = fill_header(&ctx, &h);
    if (!bad) {
      xjpeg_decode_image(&ctx, &img, XJPEG_DECODE_YUV);
      bad = ctx.error != NULL;
    }
  }
  image_clear(&img);
  return bad;
}

/* image_init on a synthetic header (test/image.c:21-55 style). */
REF_API int ref_layout(int width, int height, int ncomps, const int *hsamp,
 const int *vsamp, ref_info *o) {
  jpeg_header h;
  image img;
  int i, hmax = 0, vmax = 0, nhmb, nvmb;
  long long total = 0;
  memset(&h, 0, sizeof(h));
  h.bits = 8; h.width = width; h.height = height; h.ncomps = ncomps;
  for (i = 0; i < ncomps; i++) {
    if (hsamp[i] > hmax) hmax = hsamp[i];
    if (vsamp[i] > vmax) vmax = vsamp[i];
  }
  nhmb = (width + hmax*8 - 1)/(hmax*8);
  nvmb = (height + vmax*8 - 1)/(vmax*8);
  for (i = 0; i < ncomps; i++) {
    h.comp[i].hsamp = hsamp[i]; h.comp[i].vsamp = vsamp[i];
    h.comp[i].hblocks = nhmb*hsamp[i]; h.comp[i].vblocks = nvmb*vsamp[i];
    h.comp[i].quant = &h.quant[0];
  }
  if (image_init(&img, &h) != EXIT_SUCCESS) return 1;
  memset(o, 0, sizeof(*o));
  o->width = width; o->height = height; o->ncomps = ncomps;
  o->nhmb = nhmb; o->nvmb = nvmb;
  for (i = 0; i < ncomps; i++) {
    const image_plane *p = &img.plane[i];
    o->hsamp[i] = hsamp[i]; o->vsamp[i] = vsamp[i];
    o->hblocks[i] = h.comp[i].hblocks; o->vblocks[i] = h.comp[i].vblocks;
    o->xdec[i] = p->xdec; o->ydec[i] = p->ydec; o->cstride[i] = p->cstride;
    o->coef_off[i] = p->coef - img.coef;
    total += ((long long)p->width << (p->xdec + 3))*p->cstride;
  }
  o->coef_shorts = total;
  image_clear(&img);
  return 0;
}
