/* libjpeg_vtbl.c — LIBJPEG_DECODE_CTX_VTBL: the comparison backend.
 *
 * The reference ships two plugin instances: its own decoder and a wrapper around the
 * platform's libjpeg (src/jpeg_wrap.c:56-252, selected with `-i libjpeg`,
 * src/jpeg_gpu.c:545-557).  This is the second one for this build: the CPU path that
 * bench.py times beside the MI355X path ("the xjpeg/libjpeg-turbo CPU path on the same
 * box's host cores") and that the harness offers as `-i libjpeg`.  It is NOT on the hot
 * path and its arithmetic is third-party (libjpeg-turbo's integer IDCT differs from
 * src/dct.c by +-1 on a few percent of samples): parity for its YUV/RGB
 * stages is *unpinned*; its QUANT stage is plain Huffman decoding and equals the oracle's
 * planes exactly (tests/test_libjpeg_backend.py).
 *
 * Differences from the reference's wrapper, on purpose:
 *  - libjpeg is bound at run time (dlopen of libjpeg.so.8 + hand-declared ABI,
 *    csrc/libjpeg8_abi.h): the image has no <jpeglib.h>.  decode_alloc() returns NULL with
 *    a message when the library is not there; nothing else in the product depends on it.
 *  - library errors come back as EXIT_FAILURE (setjmp trap) instead of exit().
 *  - QUANT puts every block where the coefficient-plane layout wants it
 *    (jga_block_offset, src/xjpeg.c:556-561) instead of back to back, so frames whose width
 *    is not a multiple of the MCU have the same planes as the XJPEG/HIPJPEG backends.
 * Same as the reference's: stages QUANT, YUV, RGB only (jpeg_wrap.c:134-228); raw planes
 * and pixels without fancy upsampling, ISLOW transform (jpeg_wrap.c:166-174, 199-207). */
#include <dlfcn.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "jga_internal.h"
#include "libjpeg8_abi.h"

static struct {
  void *handle;
  lj8_std_error_fn std_error;
  lj8_create_decompress_fn create;
  lj8_mem_src_fn mem_src;
  lj8_read_header_fn read_header;
  lj8_read_coefficients_fn read_coefficients;
  lj8_start_decompress_fn start;
  lj8_read_raw_data_fn read_raw;
  lj8_read_scanlines_fn read_scanlines;
  lj8_finish_decompress_fn finish;
  lj8_destroy_decompress_fn destroy;
  int ok;
} LJ;
static pthread_once_t LJ_ONCE = PTHREAD_ONCE_INIT;

static void lj_bind(void) {
  static const char *const NAMES[] = { "libjpeg.so.8", "libjpeg.so", NULL };
  const char *env = getenv("JGA_LIBJPEG");
  int i;
  if (env && *env) LJ.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  for (i = 0; !LJ.handle && NAMES[i]; i++) LJ.handle = dlopen(NAMES[i], RTLD_NOW | RTLD_LOCAL);
  if (!LJ.handle) return;
#define BIND(field, type, sym) LJ.field = (type)dlsym(LJ.handle, sym)
  BIND(std_error, lj8_std_error_fn, "jpeg_std_error");
  BIND(create, lj8_create_decompress_fn, "jpeg_CreateDecompress");
  BIND(mem_src, lj8_mem_src_fn, "jpeg_mem_src");
  BIND(read_header, lj8_read_header_fn, "jpeg_read_header");
  BIND(read_coefficients, lj8_read_coefficients_fn, "jpeg_read_coefficients");
  BIND(start, lj8_start_decompress_fn, "jpeg_start_decompress");
  BIND(read_raw, lj8_read_raw_data_fn, "jpeg_read_raw_data");
  BIND(read_scanlines, lj8_read_scanlines_fn, "jpeg_read_scanlines");
  BIND(finish, lj8_finish_decompress_fn, "jpeg_finish_decompress");
  BIND(destroy, lj8_destroy_decompress_fn, "jpeg_destroy_decompress");
#undef BIND
  LJ.ok = LJ.std_error && LJ.create && LJ.mem_src && LJ.read_header && LJ.read_coefficients
   && LJ.start && LJ.read_raw && LJ.read_scanlines && LJ.finish && LJ.destroy;
}

typedef struct lj_ctx {
  lj8_decompress cinfo;
  lj8_error_mgr jerr;
  jmp_buf trap;
  int live;                          /* cinfo holds a created decompressor */
  char message[256];
} lj_ctx;

/* libjpeg's default error_exit() ends the process; ours unwinds to the vtable entry. */
static void lj_on_error(struct lj8_common *c) {
  lj8_decompress *cinfo = (lj8_decompress *)c;
  lj_ctx *ctx = (lj_ctx *)cinfo->client_data;
  char text[200];
  text[0] = 0;
  (*cinfo->err->format_message)(c, text);
  snprintf(ctx->message, sizeof(ctx->message), "libjpeg: %s", text);
  longjmp(ctx->trap, 1);
}
static void lj_on_message(struct lj8_common *c, int level) { (void)c; (void)level; }

static int lj_open(lj_ctx *ctx, const jpeg_info *info) {
  if (setjmp(ctx->trap)) return jga_fail("%s", ctx->message);
  ctx->cinfo.err = (*LJ.std_error)(&ctx->jerr);
  ctx->jerr.error_exit = lj_on_error;
  ctx->jerr.emit_message = lj_on_message;    /* corrupt-data warnings are not errors here either */
  ctx->cinfo.client_data = ctx;
  (*LJ.create)(&ctx->cinfo, LJ8_LIB_VERSION, sizeof(ctx->cinfo));
  ctx->cinfo.client_data = ctx;              /* create() clears the record */
  ctx->live = 1;
  (*LJ.mem_src)(&ctx->cinfo, info->buf, (unsigned long)info->size);
  return EXIT_SUCCESS;
}

static void lj_close(lj_ctx *ctx) {
  if (!ctx->live) return;
  ctx->live = 0;
  if (setjmp(ctx->trap)) return;
  (*LJ.destroy)(&ctx->cinfo);
}

static jpeg_decode_ctx *lj_alloc(jpeg_info *info) {
  lj_ctx *ctx;
  pthread_once(&LJ_ONCE, lj_bind);
  if (!LJ.ok) {
    jga_fail("libjpeg backend: %s", LJ.handle ? "libjpeg.so.8 lacks an entry point"
     : "libjpeg.so.8 not found (set JGA_LIBJPEG to its path)");
    return NULL;
  }
  ctx = (lj_ctx *)calloc(1, sizeof(*ctx));
  if (!ctx) return NULL;
  if (lj_open(ctx, info) != EXIT_SUCCESS) {
    lj_close(ctx);
    free(ctx);
    return NULL;
  }
  return (jpeg_decode_ctx *)ctx;
}

/* What src/jpeg_wrap.c:74-132 copies out of the library's record. */
static int lj_header(jpeg_decode_ctx *dec, jpeg_header *h) {
  lj_ctx *ctx = (lj_ctx *)dec;
  lj8_decompress *ci = &ctx->cinfo;
  int i, k, mcu_w, mcu_h;
  if (!ctx->live) return jga_fail("libjpeg backend: decoder is closed");
  if (setjmp(ctx->trap)) return jga_fail("%s", ctx->message);
  if ((*LJ.read_header)(ci, 1) != LJ8_HEADER_OK) return jga_fail("Error reading jpeg headers");
  memset(h, 0, sizeof(*h));
  h->width = (int)ci->image_width;
  h->height = (int)ci->image_height;
  h->bits = ci->data_precision;
  h->ncomps = ci->num_components;
  h->restart_interval = (int)ci->restart_interval;
  if (h->ncomps != 1 && h->ncomps != 3) {
    return jga_fail("Unsupported number of components %i", h->ncomps);
  }
  for (i = 0; i < NQUANT_MAX; i++) {
    const lj8_quant_tbl *t = ci->quant_tbl_ptrs[i];
    if (!t) continue;
    h->quant[i].valid = 1;
    h->quant[i].bits = 8;
    for (k = 0; k < 64; k++) {
      h->quant[i].tbl[k] = t->quantval[k];
      if (t->quantval[k] > 255) h->quant[i].bits = 16;
    }
  }
  mcu_w = ci->max_h_samp*8;
  mcu_h = ci->max_v_samp*8;
  for (i = 0; i < h->ncomps; i++) {
    const lj8_component *c = &ci->comp_info[i];
    if (c->quant_tbl_no < 0 || c->quant_tbl_no >= NQUANT_MAX || !ci->quant_tbl_ptrs[c->quant_tbl_no]) {
      return jga_fail("Missing quantization table for component %i", i);
    }
    h->comp[i].hsamp = c->h_samp;
    h->comp[i].vsamp = c->v_samp;
    h->comp[i].hblocks = (h->width + mcu_w - 1)/mcu_w*c->h_samp;
    h->comp[i].vblocks = (h->height + mcu_h - 1)/mcu_h*c->v_samp;
    h->comp[i].quant = &h->quant[c->quant_tbl_no];
  }
  if (h->ncomps == 1) h->subsamp = JPEG_SUBSAMP_MONO;
  else {
    h->subsamp = (jpeg_subsamp)jga_subsamp_of(
     jga_ilog(h->comp[0].hsamp) - jga_ilog(h->comp[1].hsamp),
     jga_ilog(h->comp[0].vsamp) - jga_ilog(h->comp[1].vsamp), 3);
  }
  return EXIT_SUCCESS;
}

static int lj_quant(lj_ctx *ctx, image *img) {
  lj8_decompress *ci = &ctx->cinfo;
  lj8_virt_blocks_ptr *arrays = (*LJ.read_coefficients)(ci);
  int p;
  if (!arrays) return jga_fail("libjpeg backend: no coefficient arrays");
  for (p = 0; p < img->nplanes && p < ci->num_components; p++) {
    const lj8_component *c = &ci->comp_info[p];
    const image_plane *pl = &img->plane[p];
    /* the library's arrays are padded to whole MCUs for an interleaved scan */
    const unsigned aw = (c->width_in_blocks + c->h_samp - 1)/c->h_samp*c->h_samp;
    const unsigned ah = (c->height_in_blocks + c->v_samp - 1)/c->v_samp*c->v_samp;
    const unsigned hb = pl->width >> 3, vb = pl->height >> 3;
    const unsigned nx = hb < aw ? hb : aw;
    unsigned by;
    for (by = 0; by < vb && by < ah; by++) {
      lj8_block_rows rows = (*ci->mem->access_virt_barray)((struct lj8_common *)ci, arrays[p],
       by, 1, 0);
      /* a plane is `cstride` luma-width rows, each holding 1<<xdec of its block rows back to
       * back (src/xjpeg.c:556-561): block row `by` therefore starts at by*hb blocks */
      memcpy(pl->coef + ((size_t)by*hb << 6), rows[0], (size_t)nx*sizeof(lj8_block));
    }
  }
  return EXIT_SUCCESS;
}

static int lj_planes(lj_ctx *ctx, image *img) {
  lj8_decompress *ci = &ctx->cinfo;
  lj8_row rowp[3][4*8];
  lj8_rows comp[3];
  int p, lines;
  ci->raw_data_out = 1;
  ci->do_fancy_upsampling = 0;
  ci->dct_method = LJ8_DCT_ISLOW;
  (*LJ.start)(ci);
  lines = ci->max_v_samp*8;                  /* one iMCU row per call */
  if (lines > 32) return jga_fail("libjpeg backend: unsupported vertical sampling");
  for (p = 0; p < 3; p++) comp[p] = rowp[p];
  while (ci->output_scanline < ci->output_height) {
    const unsigned imcu = ci->output_scanline/(unsigned)lines;
    for (p = 0; p < img->nplanes; p++) {
      const image_plane *pl = &img->plane[p];
      const int rows = ci->comp_info[p].v_samp*8;
      int j;
      for (j = 0; j < rows; j++) {
        unsigned y = imcu*rows + j;
        if (y >= pl->height) y = pl->height - 1u;
        rowp[p][j] = pl->data + (size_t)y*pl->ystride;
      }
    }
    if ((*LJ.read_raw)(ci, comp, (lj8_dim)lines) == 0) {
      return jga_fail("libjpeg backend: jpeg_read_raw_data made no progress");
    }
  }
  (*LJ.finish)(ci);
  return EXIT_SUCCESS;
}

static int lj_pixels(lj_ctx *ctx, image *img) {
  lj8_decompress *ci = &ctx->cinfo;
  const size_t pitch = (size_t)img->width*img->nplanes;      /* src/jpeg_wrap.c:215-219 */
  lj8_row row = img->pixels;
  ci->do_fancy_upsampling = 0;
  ci->dct_method = LJ8_DCT_ISLOW;
  (*LJ.start)(ci);
  if ((int)ci->output_components != img->nplanes || ci->output_width != img->width) {
    return jga_fail("libjpeg backend: output is %ux%ux%i, image is %ix%ix%i", ci->output_width,
     ci->output_height, ci->output_components, img->width, img->height, img->nplanes);
  }
  while (ci->output_scanline < ci->output_height) {
    if ((*LJ.read_scanlines)(ci, &row, 1) != 1) {
      return jga_fail("libjpeg backend: jpeg_read_scanlines made no progress");
    }
    row += pitch;
  }
  (*LJ.finish)(ci);
  return EXIT_SUCCESS;
}

static int lj_image(jpeg_decode_ctx *dec, image *img, jpeg_decode_out out) {
  static const char *const NAMES[JPEG_DECODE_OUT_MAX] = { "pack", "quant", "dct", "yuv", "rgb" };
  lj_ctx *ctx = (lj_ctx *)dec;
  if (!ctx->live) return jga_fail("libjpeg backend: decoder is closed");
  if (setjmp(ctx->trap)) return jga_fail("%s", ctx->message);
  switch (out) {
    case JPEG_DECODE_QUANT : return lj_quant(ctx, img);
    case JPEG_DECODE_YUV : return lj_planes(ctx, img);
    case JPEG_DECODE_RGB : return lj_pixels(ctx, img);
    default : break;
  }
  return jga_fail("Unsupported output '%s' for libjpeg wrapper.",
   (unsigned)out < JPEG_DECODE_OUT_MAX ? NAMES[out] : "?");
}

static void lj_reset(jpeg_decode_ctx *dec, jpeg_info *info) {
  lj_ctx *ctx = (lj_ctx *)dec;
  lj_close(ctx);
  (void)lj_open(ctx, info);
}

static void lj_free(jpeg_decode_ctx *dec) {
  lj_ctx *ctx = (lj_ctx *)dec;
  if (!ctx) return;
  lj_close(ctx);
  free(ctx);
}

JGA_EXPORT const jpeg_decode_ctx_vtbl JGA_LIBJPEG_DECODE_CTX_VTBL = {
  lj_alloc, lj_header, lj_image, lj_reset, lj_free
};

JGA_EXPORT int jga_libjpeg_available(void) {
  pthread_once(&LJ_ONCE, lj_bind);
  return LJ.ok;
}
