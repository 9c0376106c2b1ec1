/* pack_params.h — launch descriptor of the PACK expansion kernel (pack_kernels.hip). */
#ifndef JGA_PACK_PARAMS_H
#define JGA_PACK_PARAMS_H (1)
#include <stdint.h>

typedef struct jga_pack_params {
  const uint16_t *pack;       /* image i: words at pack + i*pack_stride */
  const int32_t *index;       /* image i: block start words at index + i*index_stride */
  int16_t *coef;              /* image i: planes at coef + i*coef_stride (shorts) */
  long long pack_stride;      /* words, even */
  long long index_stride;     /* ints */
  long long coef_stride;      /* shorts */
  long long pack_words;       /* readable words per image (<= pack_stride) */
  int nimages;
  int nplanes;
  int w0_blocks;              /* luma blocks per row; RS = w0_blocks*64 shorts */
  int plane_hblocks[3];
  int plane_xdec[3];
  int plane_first[4];         /* flat number of each plane's first real block; [nplanes] = total */
  int plane_index0[3];        /* first index entry of each plane (src/image.c:93-94) */
  long long plane_coef_off[3];
  /* scan order (MCU-interleaved), the order the words are stored in */
  int nhmb, nvmb;             /* MCUs per row / rows */
  int nslots;                 /* blocks per MCU */
  struct { uint32_t mul, shift; } div_nslots, div_nhmb;   /* n/d == (n*mul) >> shift, n < 2^31 */
  unsigned long long slot_desc[2]; /* 6 bits per slot: plane | sbx << 2 | sby << 4; ten slots per word
                                 * (18 = luma 4x4 + 2 is the most the reference's factors give) */
  int plane_hs[3], plane_vs[3];
} jga_pack_params;

#ifdef __cplusplus
extern "C" {
#endif
int jga_launch_unpack(const jga_pack_params *P, void *stream);
#ifdef __cplusplus
}
#endif
#endif
