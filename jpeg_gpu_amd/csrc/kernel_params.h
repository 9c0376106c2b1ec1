/* kernel_params.h — launch descriptor shared by the kernels and their C-ABI
 * wrappers.  One descriptor covers a batch of images with identical geometry
 * (the batch shape of BASELINE.json's configs). */
#ifndef JGA_KERNEL_PARAMS_H
#define JGA_KERNEL_PARAMS_H (1)
#include <stdint.h>

/* n / d == (n * mul) >> shift for every 0 <= n < 2^31 */
typedef struct jga_divisor {
  uint32_t mul;
  uint32_t shift;
} jga_divisor;

typedef struct jga_kparams {
  const int16_t *coef;        /* image i at coef + i*coef_stride (shorts) */
  const uint16_t *qtab;       /* image i plane p at qtab + (i*3+p)*64 */
  uint8_t *out;               /* image i at out + i*out_stride (bytes) */
  long long coef_stride;
  long long out_stride;
  int nimages;
  int nplanes;
  int dequant;                /* 1: coef holds quantised levels (QUANT stage) */
  int width, height;          /* true size */
  int w0_blocks;              /* luma blocks per row; RS = w0_blocks*64 shorts */
  int slots_per_image;        /* 128-byte slots in one coefficient buffer */
  int nhmb;                   /* MCUs per row */
  int nvmb;                   /* MCU rows */
  jga_divisor div_w0;         /* by w0_blocks */
  jga_divisor div_hb[3];      /* by w0_blocks >> xdec of each plane */
  int out_aligned;            /* output rows allow dword/qword vector stores */
  int plane_hblocks[3];
  int plane_vblocks[3];
  int plane_xdec[3];
  int plane_slot0[3];         /* first slot of each plane */
  long long plane_coef_off[3];
  long long plane_data_off[3];
  const int16_t *dc;          /* NULL, or DC values beside the planes: image i slot s at dc[i*dc_stride + s] */
  long long dc_stride;
} jga_kparams;

#ifdef __cplusplus
extern "C" {
#endif
int jga_launch_rgb(const jga_kparams *P, int xdec, int ydec, int staged,
 void *stream);
int jga_launch_yuv(const jga_kparams *P, int staged, void *stream);
/* P->coef = the YUV-stage bytes (as int16*), coef_stride in BYTES */
int jga_launch_yuv_rgb(const jga_kparams *P, int uxdec, int uydec, int vxdec, int vydec, void *stream);
#ifdef __cplusplus
}
#endif
#endif
