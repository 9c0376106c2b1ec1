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

static const char *OPTSTRING = "hi:o:dH";
static const struct option OPTIONS[] = {
  { "help", no_argument, NULL, 'h' },
  { "no-cpu", no_argument, NULL, 0 },
  { "no-gpu", no_argument, NULL, 0 },
  { "impl", required_argument, NULL, 'i' },
  { "out", required_argument, NULL, 'o' },
  { "dump", no_argument, NULL, 'd' },
  { "header", no_argument, NULL, 'H' },
  { "frames", required_argument, NULL, 0 },
  { "seconds", required_argument, NULL, 0 },
  { "check", no_argument, NULL, 0 },
  { NULL, 0, NULL, 0 }
};

static void usage(void) {
  fprintf(stderr,
   "Usage: %s [options] jpeg_file\n\n"
   "Options:\n\n"
   "  -h --help                      Display this help and exit.\n"
   "     --no-cpu                    Disable CPU decoding in main loop.\n"
   "     --no-gpu                    Disable GPU decoding in main loop.\n"
   "  -i --impl <decoder>            Software decoder to use.\n"
   "                                 hipjpeg (default) => MI355X plugin\n"
   "  -o --out <format>              Format software decoder should output\n"
   "                                  and send to the GPU.\n"
   "                                 pack => RLC zero packed and quantized.\n"
   "                                 quant => quantized but de-zigzaged.\n"
   "                                 dct => DCT (12-bit dequantized)\n"
   "                                 yuv (default) => YUV (4:4:4 or 4:2:0)\n"
   "                                 rgb => RGB (4:4:4)\n"
   "  -d --dump                      Dump jpeg data in the output format.\n"
   "  -H --header                    Print the jpeg header.\n"
   "     --frames <n>                Stop the main loop after n frames.\n"
   "     --seconds <s>               Stop the main loop after s seconds (default 3).\n"
   "     --check                     Print the Adler-32 of the final RGB image.\n\n"
   " %s accepts only 8-bit non-hierarchical JPEG files.\n\n", NAME, NAME);
}

static int read_file(jpeg_info *info, const char *name) {   /* src/jpeg_info.c:30-62 */
  FILE *fp = fopen(name, "rb");
  int size;
  if (fp == NULL) {
    fprintf(stderr, "Error, could not open jpeg file %s\n", name);
    return EXIT_FAILURE;
  }
  free(info->buf);
  memset(info, 0, sizeof(*info));
  fseek(fp, 0, SEEK_END);
  info->size = (int)ftell(fp);
  info->buf = malloc(info->size ? info->size : 1);
  if (info->buf == NULL) {
    fprintf(stderr, "Error, could not allocate %i bytes\n", info->size);
    fclose(fp);
    return EXIT_FAILURE;
  }
  fseek(fp, 0, SEEK_SET);
  size = (int)fread(info->buf, 1, info->size, fp);
  fclose(fp);
  if (size != info->size) {
    fprintf(stderr, "Error reading jpeg file, got %i of %i bytes\n", size, info->size);
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9*(double)ts.tv_nsec;
}

static void print_header(const jpeg_header *h) {            /* src/jpeg_gpu.c:614-637 */
  int i, j;
  printf("Image Size         : %ix%i\n", h->width, h->height);
  printf("Bits Per Pixel     : %i\n", h->bits);
  printf("Components         : %i\n", h->ncomps);
  printf("Chroma Subsampling : %s\n", SUBSAMP_NAMES[h->subsamp]);
  printf("Minimum Coded Unit : ");
  for (i = 0; i < h->ncomps; i++) {
    printf("%s%ix%i", i > 0 ? " " : "", h->comp[i].hsamp, h->comp[i].vsamp);
  }
  printf("\n");
  printf("Restart Interval   : %i\n", h->restart_interval);
  for (i = 0; i < NQUANT_MAX; i++) {
    if (h->quant[i].valid) {
      printf("Quant Table %i Bits : %i\n", i, h->quant[i].bits);
      for (j = 1; j <= 64; j++) {
        printf("%4i%s", h->quant[i].tbl[j - 1], j & 0x7 ? "" : "\n");
      }
    }
  }
}

/* src/jpeg_gpu.c:643-700.  QUANT/DCT print the plane's share of the coefficient
 * buffer as `height` rows of `width` shorts, exactly as the reference indexes it.
 * RGB prints component i of every pixel with the img->pixels pitch that its only
 * writer uses (width*3, src/jpeg_wrap.c:215-219); the reference's own loop
 * multiplies by the padded plane width there. */
static int dump_image(const image *img, jpeg_decode_out out) {
  int i, j, k;
  if (out == JPEG_DECODE_PACK) {
    int packed = 0;
    for (i = 0; i < img->nplanes; i++) {
      printf("Plane %i Packed Data: %i\n", i, img->plane[i].packed);
      packed += img->plane[i].packed;
    }
    printf("Packed Data : %i\n", packed);
    return EXIT_SUCCESS;
  }
  for (i = 0; i < img->nplanes; i++) {
    const image_plane *plane = &img->plane[i];
    printf("Plane %i\n", i);
    switch (out) {
      case JPEG_DECODE_QUANT :
      case JPEG_DECODE_DCT : {
        for (k = 0; k < plane->height; k++) {
          for (j = 0; j < plane->width; j++) printf("%4i ", plane->coef[k*plane->width + j]);
          printf("\n");
        }
        break;
      }
      case JPEG_DECODE_YUV : {
        for (k = 0; k < plane->height; k++) {
          for (j = 0; j < plane->width; j++) printf("%4i ", plane->data[k*plane->width + j]);
          printf("\n");
        }
        break;
      }
      case JPEG_DECODE_RGB : {
        for (k = 0; k < img->height; k++) {
          for (j = 0; j < img->width; j++) {
            printf("%4i ", img->pixels[((long)k*img->width + j)*img->nplanes + i]);
          }
          printf("\n");
        }
        break;
      }
      default : {
        fprintf(stderr, "Unsupported output '%s'.\n",
         (unsigned)out < JPEG_DECODE_OUT_MAX ? OUT_NAMES[out] : "?");
        return EXIT_FAILURE;
      }
    }
    printf("\n");
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
  int c, loi;
  memset(&info, 0, sizeof(info));
  /* this program keeps one image for the life of each decoder context, so the plugin may
   * register its buffers and copy results straight into them */
  setenv("JGA_PLUGIN_REGISTER", "1", 0);
  while ((c = getopt_long(argc, argv, OPTSTRING, OPTIONS, &loi)) != EOF) {
    switch (c) {
      case 0 : {
        const char *n = OPTIONS[loi].name;
        if (strcmp(n, "no-cpu") == 0) no_cpu = 1;
        else if (strcmp(n, "no-gpu") == 0) no_gpu = 1;
        else if (strcmp(n, "frames") == 0) max_frames = atol(optarg);
        else if (strcmp(n, "seconds") == 0) max_seconds = atof(optarg);
        else if (strcmp(n, "check") == 0) check = 1;
        break;
      }
      case 'i' : {
        if (strcmp("hipjpeg", optarg) == 0) vtbl = HIPJPEG_DECODE_CTX_VTBL;
        else {
          fprintf(stderr, "Invalid decoder implementation: %s\n", optarg);
          usage();
          return EXIT_FAILURE;
        }
        break;
      }
      case 'o' : {
        int k, found = 0;
        for (k = 0; k < JPEG_DECODE_OUT_MAX; k++) {
          if (strcmp(OUT_NAMES[k], optarg) == 0) { out = (jpeg_decode_out)k; found = 1; }
        }
        if (!found) {
          fprintf(stderr, "Invalid decoder output format: %s\n", optarg);
          usage();
          return EXIT_FAILURE;
        }
        break;
      }
      case 'd' : dump = 1; break;
      case 'H' : head = 1; break;
      case 'h' :
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
