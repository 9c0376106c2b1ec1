/* harness.c — headless counterpart of the reference's `jpeg_gpu` program
 * (src/jpeg_gpu.c:473-506 options, 508-606 parsing, 610-704 header / dump /
 * first decode, 1215-1465 steady-state loop) with the MI355X plugin sitting where
 * XJPEG_DECODE_CTX_VTBL sits.  Same options, same `--header` and `--dump` text,
 * same per-frame call order (decode_reset -> decode_header -> decode_image, then
 * the device finishes the frame from the `-o` stage); the GLFW window, its
 * title-bar statistics and the shader passes are replaced by the HIP kernels and
 * one statistics line per second on stdout.  Plain C over the C-ABI only.
 *
 * Per `-o` stage, what the "CPU" part leaves and what the "GPU" part then does
 * (reference: upload + draw passes, src/jpeg_gpu.c:1312-1423):
 *   pack   words + block index   H2D, jga_unpack_batch, fused IDCT+RGB kernel
 *   quant  level planes          H2D, fused IDCT+RGB kernel (dequantising)
 *   dct    dequantised planes    H2D, fused IDCT+RGB kernel
 *   yuv    Y/Cb/Cr planes        H2D, jga_yuv_rgb_batch (pass 3 alone)
 *   rgb    pixels                nothing left to do (the reference only blits)
 * Additions (not in the reference): --frames N / --seconds S bound the loop that
 * the reference ends by closing its window; --check prints a checksum of the RGB
 * left in HBM. */
#include <getopt.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../../include/jpeg_gpu_amd.h"

#define NAME "jpeg_gpu_hip"

static const char *SUBSAMP_NAMES[JPEG_SUBSAMP_MAX] = {   /* src/jpeg_info.c:20-28 */
  "Unknown", "4:4:4", "4:2:2", "4:2:0", "4:4:0", "4:1:1", "Mono"
};
static const char *OUT_NAMES[JPEG_DECODE_OUT_MAX] = {    /* src/jpeg_wrap.c:24-30 */
  "pack", "quant", "dct", "yuv", "rgb"
};

/* ---- command line ---------------------------------------------------------------
 * One table drives getopt_long, the help text and the dispatch.  The option NAMES and
 * letters are the reference program's (src/jpeg_gpu.c:473-483), since scripts written for
 * it must keep working; --frames / --seconds / --check are this program's own. */
enum opt_id { OPT_HELP, OPT_NO_CPU, OPT_NO_GPU, OPT_IMPL, OPT_OUT, OPT_DUMP, OPT_HEADER,
 OPT_FRAMES, OPT_SECONDS, OPT_CHECK, OPT_COUNT };

typedef struct opt_spec {
  const char *name;            /* long name */
  char letter;                 /* short form, 0 if none */
  const char *arg;             /* placeholder of its argument, NULL if it takes none */
  const char *help;            /* '\n' separates lines; continuation lines are indented */
} opt_spec;

static const opt_spec SPECS[OPT_COUNT] = {
  [OPT_HELP]    = { "help", 'h', NULL, "Show this text." },
  [OPT_NO_CPU]  = { "no-cpu", 0, NULL, "Main loop: skip the host decode (the first frame is kept)." },
  [OPT_NO_GPU]  = { "no-gpu", 0, NULL, "Main loop: skip the device stage." },
  [OPT_IMPL]    = { "impl", 'i', "<decoder>", "Decoder plugin:\n"
                    "hipjpeg (default) => MI355X plugin (HIP kernels)\n"
                    "libjpeg => the platform's libjpeg, CPU only (quant, yuv, rgb)" },
  [OPT_OUT]     = { "out", 'o', "<format>", "Stage at which the plugin stops; the device\n"
                    " finishes the frame from there:\n"
                    "pack => run/level words + per-block index\n"
                    "quant => quantised levels, natural order\n"
                    "dct => dequantised coefficients\n"
                    "yuv (default) => component planes\n"
                    "rgb => interleaved pixels" },
  [OPT_DUMP]    = { "dump", 'd', NULL, "Print the first frame at the chosen stage and exit." },
  [OPT_HEADER]  = { "header", 'H', NULL, "Print what the frame header says and exit." },
  [OPT_FRAMES]  = { "frames", 0, "<n>", "Leave the main loop after n frames." },
  [OPT_SECONDS] = { "seconds", 0, "<s>", "Leave the main loop after s seconds (default 3)." },
  [OPT_CHECK]   = { "check", 0, NULL, "Print the Adler-32 of the last RGB frame in HBM." },
};

static void usage(void) {
  int i;
  fprintf(stderr, "Usage: %s [options] jpeg_file\n\nOptions:\n\n", NAME);
  for (i = 0; i < OPT_COUNT; i++) {
    const opt_spec *o = &SPECS[i];
    const char *line = o->help;
    char left[40];
    if (o->letter) snprintf(left, sizeof(left), "-%c --%s %s", o->letter, o->name, o->arg ? o->arg : "");
    else snprintf(left, sizeof(left), "   --%s %s", o->name, o->arg ? o->arg : "");
    while (line) {
      const char *nl = strchr(line, '\n');
      const int n = nl ? (int)(nl - line) : (int)strlen(line);
      fprintf(stderr, "  %-30s %.*s\n", line == o->help ? left : "", n, line);
      line = nl ? nl + 1 : NULL;
    }
  }
  fprintf(stderr, "\n %s reads 8-bit baseline (SOF0) JPEG files.\n\n", NAME);
}

/* getopt_long's view of SPECS: long options report their table index through `flag`-less
 * val = 256 + index, short ones their letter. */
static int next_option(int argc, char *argv[]) {
  static struct option longs[OPT_COUNT + 1];
  static char shorts[2*OPT_COUNT + 1];
  int c, i;
  if (!longs[0].name) {
    char *s = shorts;
    for (i = 0; i < OPT_COUNT; i++) {
      longs[i].name = SPECS[i].name;
      longs[i].has_arg = SPECS[i].arg ? required_argument : no_argument;
      longs[i].val = 256 + i;
      if (SPECS[i].letter) {
        *s++ = SPECS[i].letter;
        if (SPECS[i].arg) *s++ = ':';
      }
    }
  }
  c = getopt_long(argc, argv, shorts, longs, NULL);
  if (c == -1) return -1;
  if (c >= 256) return c - 256;
  for (i = 0; i < OPT_COUNT; i++) if (SPECS[i].letter == c) return i;
  return OPT_HELP;                                  /* '?': unknown option */
}

static int read_file(jpeg_info *info, const char *name) {   /* the job of src/jpeg_info.c:30-62 */
  FILE *fp = fopen(name, "rb");
  long size;
  unsigned char *buf = NULL;
  if (fp == NULL) {
    fprintf(stderr, "Error, could not open jpeg file %s\n", name);
    return EXIT_FAILURE;
  }
  if (fseek(fp, 0, SEEK_END) == 0 && (size = ftell(fp)) >= 0 && size <= 0x7fffffffL
   && fseek(fp, 0, SEEK_SET) == 0 && (buf = malloc(size ? (size_t)size : 1)) != NULL
   && fread(buf, 1, (size_t)size, fp) == (size_t)size) {
    fclose(fp);
    free(info->buf);
    info->buf = buf;
    info->size = (int)size;
    return EXIT_SUCCESS;
  }
  fprintf(stderr, "Error reading jpeg file %s\n", name);
  free(buf);
  fclose(fp);
  return EXIT_FAILURE;
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9*(double)ts.tv_nsec;
}

/* ---- --header: the text of src/jpeg_gpu.c:614-637, byte for byte (tests/test_harness.py).
 * Every line is "label, padded to 19 columns, colon, value". */
static void field(const char *label, const char *fmt, ...) {
  va_list ap;
  printf("%-19s: ", label);
  va_start(ap, fmt);
  vprintf(fmt, ap);
  va_end(ap);
  putchar('\n');
}

static void print_header(const jpeg_header *h) {
  char text[64];
  int i, k, n = 0;
  field("Image Size", "%ix%i", h->width, h->height);
  field("Bits Per Pixel", "%i", h->bits);
  field("Components", "%i", h->ncomps);
  field("Chroma Subsampling", "%s", SUBSAMP_NAMES[h->subsamp]);
  for (i = 0; i < h->ncomps; i++) {
    n += snprintf(text + n, sizeof(text) - n, "%s%ix%i", i ? " " : "", h->comp[i].hsamp, h->comp[i].vsamp);
  }
  field("Minimum Coded Unit", "%s", text);
  field("Restart Interval", "%i", h->restart_interval);
  for (i = 0; i < NQUANT_MAX; i++) {
    const jpeg_quant *q = &h->quant[i];
    if (!q->valid) continue;
    snprintf(text, sizeof(text), "Quant Table %i Bits", i);
    field(text, "%i", q->bits);
    for (k = 0; k < 64; k++) printf((k & 7) == 7 ? "%4i\n" : "%4i", q->tbl[k]);
  }
}

/* ---- --dump: the text of src/jpeg_gpu.c:643-700.  Every stage but PACK is "Plane i", a
 * matrix of "%4i " cells, a blank line; what differs is where cell (row, col) comes from. */
typedef struct cell_source {
  const image *img;
  const image_plane *plane;
  int comp;
} cell_source;

/* QUANT/DCT: the plane's share of the coefficient buffer read as `height` rows of `width`
 * shorts — how the reference indexes it, not block by block. */
static int cell_coef(const cell_source *s, long row, long col) {
  return s->plane->coef[row*s->plane->width + col];
}
static int cell_sample(const cell_source *s, long row, long col) {
  return s->plane->data[row*s->plane->ystride + col];
}
/* RGB: component `comp` of every pixel, with the pitch the only writer of img->pixels uses
 * (width*nplanes, src/jpeg_wrap.c:215-219; the reference's dump multiplies by the padded
 * plane width here — SURVEY.md Appendix E). */
static int cell_pixel(const cell_source *s, long row, long col) {
  return s->img->pixels[(row*s->img->width + col)*s->img->nplanes + s->comp];
}

static int dump_image(const image *img, jpeg_decode_out out) {
  int (*cell)(const cell_source *, long, long);
  cell_source src;
  long row, col, rows, cols;
  if (out == JPEG_DECODE_PACK) {
    int total = 0;
    for (src.comp = 0; src.comp < img->nplanes; src.comp++) {
      printf("Plane %i Packed Data: %i\n", src.comp, img->plane[src.comp].packed);
      total += img->plane[src.comp].packed;
    }
    printf("Packed Data : %i\n", total);
    return EXIT_SUCCESS;
  }
  cell = out == JPEG_DECODE_YUV ? cell_sample : out == JPEG_DECODE_RGB ? cell_pixel
   : (out == JPEG_DECODE_QUANT || out == JPEG_DECODE_DCT) ? cell_coef : NULL;
  if (!cell) {
    fprintf(stderr, "Unsupported output '%s'.\n", (unsigned)out < JPEG_DECODE_OUT_MAX ? OUT_NAMES[out] : "?");
    return EXIT_FAILURE;
  }
  src.img = img;
  for (src.comp = 0; src.comp < img->nplanes; src.comp++) {
    src.plane = &img->plane[src.comp];
    rows = out == JPEG_DECODE_RGB ? img->height : src.plane->height;
    cols = out == JPEG_DECODE_RGB ? img->width : src.plane->width;
    printf("Plane %i\n", src.comp);
    for (row = 0; row < rows; row++) {
      for (col = 0; col < cols; col++) printf("%4i ", cell(&src, row, col));
      putchar('\n');
    }
    putchar('\n');
  }
  return EXIT_SUCCESS;
}

/* Device side of the main loop: buffers sized once, reused every frame. */
typedef struct gpu_side {
  jga_geom g;
  void *stream;
  short *d_coef;
  unsigned short *d_pack, *d_q;
  int *d_index;
  unsigned char *d_yuv, *d_rgb;
  long long nindex, rgb_cap, pack_cap;
} gpu_side;

static int gpu_init(gpu_side *s, const jpeg_header *h) {
  unsigned short q[3*64];
  int i;
  memset(s, 0, sizeof(*s));
  if (jga_geom_from_header(&s->g, h) != EXIT_SUCCESS) return EXIT_FAILURE;
  s->nindex = jga_index_count(&s->g);
  s->rgb_cap = (s->g.rgb_bytes + 15) & ~15ll;
  s->pack_cap = (s->g.coef_shorts + 1) & ~1ll;
  s->stream = jga_stream_create();
  s->d_coef = jga_device_malloc(s->g.coef_shorts*sizeof(short));
  s->d_pack = jga_device_malloc(s->pack_cap*sizeof(short));
  s->d_index = jga_device_malloc(s->nindex*sizeof(int));
  s->d_yuv = jga_device_malloc(s->g.yuv_bytes);
  s->d_rgb = jga_device_malloc(s->rgb_cap);
  s->d_q = jga_device_malloc(sizeof(q));
  if (!s->stream || !s->d_coef || !s->d_pack || !s->d_index || !s->d_yuv || !s->d_rgb || !s->d_q) {
    return EXIT_FAILURE;
  }
  memset(q, 0, sizeof(q));
  for (i = 0; i < h->ncomps; i++) memcpy(q + 64*i, h->comp[i].quant->tbl, 64*sizeof(unsigned short));
  if (jga_memcpy_h2d(s->d_q, q, sizeof(q), s->stream) != EXIT_SUCCESS) return EXIT_FAILURE;
  return jga_stream_sync(s->stream);
}

static void gpu_clear(gpu_side *s) {
  jga_device_free(s->d_coef); jga_device_free(s->d_pack); jga_device_free(s->d_index);
  jga_device_free(s->d_yuv); jga_device_free(s->d_rgb); jga_device_free(s->d_q);
  jga_stream_destroy(s->stream);
  memset(s, 0, sizeof(*s));
}

/* Finish one frame on the device from the stage the host left in `img`. */
static int gpu_frame(gpu_side *s, const image *img, jpeg_decode_out out) {
  const jga_geom *g = &s->g;
  int i;
  switch (out) {
    case JPEG_DECODE_PACK : {
      const long long even = ((long long)img->packed + 1) & ~1ll;
      if (even > s->pack_cap) return EXIT_FAILURE;
      if (jga_memcpy_h2d(s->d_pack, img->coef, even*sizeof(short), s->stream)
       || jga_memcpy_h2d(s->d_index, img->index, s->nindex*sizeof(int), s->stream)
       || jga_unpack_batch(g, 1, s->d_pack, s->pack_cap, img->packed, s->d_index, s->nindex,
       s->d_coef, g->coef_shorts, s->stream)
       || jga_idct_rgb_batch(g, 1, s->d_coef, g->coef_shorts, s->d_q, 1, s->d_rgb, s->rgb_cap,
       s->stream)) {
        return EXIT_FAILURE;
      }
      break;
    }
    case JPEG_DECODE_QUANT :
    case JPEG_DECODE_DCT : {
      if (jga_memcpy_h2d(s->d_coef, img->coef, g->coef_shorts*sizeof(short), s->stream)
       || jga_idct_rgb_batch(g, 1, s->d_coef, g->coef_shorts, s->d_q, out == JPEG_DECODE_QUANT,
       s->d_rgb, s->rgb_cap, s->stream)) {
        return EXIT_FAILURE;
      }
      break;
    }
    case JPEG_DECODE_YUV : {
      for (i = 0; i < img->nplanes; i++) {
        if (jga_memcpy_h2d(s->d_yuv + g->plane[i].data_off, img->plane[i].data,
         (size_t)img->plane[i].ystride*img->plane[i].height, s->stream)) {
          return EXIT_FAILURE;
        }
      }
      if (jga_yuv_rgb_batch(g, 1, s->d_yuv, g->yuv_bytes, s->d_rgb, s->rgb_cap, s->stream)) {
        return EXIT_FAILURE;
      }
      break;
    }
    case JPEG_DECODE_RGB : {
      if (jga_memcpy_h2d(s->d_rgb, img->pixels, g->rgb_bytes, s->stream)) return EXIT_FAILURE;
      break;
    }
    default : return EXIT_FAILURE;
  }
  return jga_stream_sync(s->stream);                           /* the reference's glFinish */
}

int main(int argc, char *argv[]) {
  jpeg_decode_ctx_vtbl vtbl = HIPJPEG_DECODE_CTX_VTBL;
  jpeg_decode_out out = JPEG_DECODE_YUV;
  int no_cpu = 0, no_gpu = 0, dump = 0, head = 0, check = 0;
  long max_frames = 0;
  double max_seconds = 3.0;
  jpeg_info info;
  jpeg_header header;
  image img;
  jpeg_decode_ctx *dec;
  int c;
  memset(&info, 0, sizeof(info));
  /* this program keeps one image for the life of each decoder context, so the plugin may
   * register its buffers once and copy results straight into them (JPEG_GPU_HIP_REGISTER=0: the plugin's
   * default for callers that make no such promise — copies on ordinary memory; -1: staged copies) */
  {
    jga_plugin_config pc;
    const char *e = getenv("JPEG_GPU_HIP_REGISTER"), *h = getenv("JPEG_GPU_HIP_ENTROPY");
    jga_plugin_config_init(&pc);
    pc.register_buffers = e ? atoi(e) : 1;
    pc.host_entropy = h && strcmp(h, "host") == 0;           /* Huffman decoding on the host for -o yuv / rgb too */
    if (jga_plugin_configure(&pc) != EXIT_SUCCESS) return EXIT_FAILURE;
  }
  while ((c = next_option(argc, argv)) >= 0) {
    switch ((enum opt_id)c) {
      case OPT_NO_CPU : no_cpu = 1; break;
      case OPT_NO_GPU : no_gpu = 1; break;
      case OPT_FRAMES : max_frames = atol(optarg); break;
      case OPT_SECONDS : max_seconds = atof(optarg); break;
      case OPT_CHECK : check = 1; break;
      case OPT_DUMP : dump = 1; break;
      case OPT_HEADER : head = 1; break;
      case OPT_IMPL : {
        if (strcmp(optarg, "hipjpeg") == 0) vtbl = HIPJPEG_DECODE_CTX_VTBL;
        else if (strcmp(optarg, "libjpeg") == 0) vtbl = JGA_LIBJPEG_DECODE_CTX_VTBL;
        else {
          fprintf(stderr, "Invalid decoder implementation: %s\n", optarg);
          usage();
          return EXIT_FAILURE;
        }
        break;
      }
      case OPT_OUT : {
        int k;
        for (k = 0; k < JPEG_DECODE_OUT_MAX && strcmp(OUT_NAMES[k], optarg) != 0; k++) ;
        if (k == JPEG_DECODE_OUT_MAX) {
          fprintf(stderr, "Invalid decoder output format: %s\n", optarg);
          usage();
          return EXIT_FAILURE;
        }
        out = (jpeg_decode_out)k;
        break;
      }
      default : usage(); return EXIT_FAILURE;
    }
  }
  for (; optind < argc; optind++) {                    /* the last file name wins */
    if (read_file(&info, argv[optind]) != EXIT_SUCCESS) return EXIT_FAILURE;
  }
  if (info.buf == NULL) {
    usage();
    return EXIT_FAILURE;
  }

  /* header, buffers, first decode (src/jpeg_gpu.c:610-704) */
  dec = (*vtbl.decode_alloc)(&info);
  if (dec == NULL) return EXIT_FAILURE;
  if ((*vtbl.decode_header)(dec, &header) != EXIT_SUCCESS) return EXIT_FAILURE;
  if (head) {
    print_header(&header);
    return EXIT_SUCCESS;
  }
  if (jga_image_init(&img, &header) != EXIT_SUCCESS) {
    fprintf(stderr, "Error initializing image\n");
    return EXIT_FAILURE;
  }
  if (dump) {
    if ((*vtbl.decode_image)(dec, &img, out) != EXIT_SUCCESS) return EXIT_FAILURE;
    return dump_image(&img, out);
  }
  if ((*vtbl.decode_image)(dec, &img, out) != EXIT_SUCCESS) return EXIT_FAILURE;
  (*vtbl.decode_free)(dec);

  /* steady state (src/jpeg_gpu.c:1215-1465) */
  {
    gpu_side gs;
    double time, last, start, cpu = 0.0;
    long frames = 0, total = 0;
    int rc = EXIT_SUCCESS;
    memset(&gs, 0, sizeof(gs));
    if (!no_gpu && gpu_init(&gs, &header) != EXIT_SUCCESS) {
      fprintf(stderr, "Error initializing the device: %s\n", jga_last_error());
      return EXIT_FAILURE;
    }
    dec = (*vtbl.decode_alloc)(&info);
    if (dec == NULL) return EXIT_FAILURE;
    /* the reference zeroes the image here (src/jpeg_gpu.c:1229); with --no-cpu the
     * first decode is kept instead, so that the device still has a frame to finish */
    if (!no_cpu) jga_image_zero(&img);
    start = time = last = now();
    for (;;) {
      if (!no_cpu) {
        (*vtbl.decode_reset)(dec, &info);
        (*vtbl.decode_header)(dec, &header);
        if ((*vtbl.decode_image)(dec, &img, out) != EXIT_SUCCESS) { rc = EXIT_FAILURE; break; }
      }
      cpu += now() - time;
      if (!no_gpu && gpu_frame(&gs, &img, out) != EXIT_SUCCESS) {
        fprintf(stderr, "Device stage failed: %s\n", jga_last_error());
        rc = EXIT_FAILURE;
        break;
      }
      frames++;
      total++;
      time = now();
      if (time - last >= 1.0 || (max_frames && total >= max_frames)
       || time - start >= max_seconds) {
        double diff = (time - last)*1000;
        cpu *= 1000;
        printf("%li FPS (cpu %0.3f ms, gpu %0.3f ms, total %.f)\n", frames, cpu/frames,
         (diff - cpu)/frames, diff);
        fflush(stdout);
        frames = 0;
        last = time;
        cpu = 0;
      }
      if ((max_frames && total >= max_frames) || time - start >= max_seconds) break;
    }
    if (rc == EXIT_SUCCESS && check && !no_gpu) {
      unsigned char *rgb = malloc(gs.g.rgb_bytes);
      unsigned long sum = 1, sum2 = 0;                /* Adler-32 */
      long long k;
      if (rgb && jga_memcpy_d2h(rgb, gs.d_rgb, gs.g.rgb_bytes, gs.stream) == EXIT_SUCCESS
       && jga_stream_sync(gs.stream) == EXIT_SUCCESS) {
        for (k = 0; k < gs.g.rgb_bytes; k++) {
          sum = (sum + rgb[k]) % 65521u;
          sum2 = (sum2 + sum) % 65521u;
        }
        printf("RGB %ix%ix%i adler32 %08lx\n", gs.g.width, gs.g.height, gs.g.nplanes,
         (sum2 << 16) | sum);
      }
      free(rgb);
    }
    if (!no_gpu) gpu_clear(&gs);
    (*vtbl.decode_free)(dec);
    free(info.buf);
    jga_image_clear(&img);
    return rc;
  }
}
