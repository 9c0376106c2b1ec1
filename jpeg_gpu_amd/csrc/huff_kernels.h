/* huff_kernels.h — launch arguments of the GPU entropy stage (huff_kernels.hip). */
#ifndef JGA_HUFF_KERNELS_H
#define JGA_HUFF_KERNELS_H (1)
#include <stddef.h>
#include "huff_common.h"

#define HJ_MAX_ROUNDS 256
#define HJ_LIST_CSTRIDE 64       /* words between two list counters: a 256-byte line each */

typedef struct hj_args {
  const hj_image *images;      /* [nimages] */
  const hj_segment *segs;      /* all images' segments, image-major */
  const uint32_t *sub_seg;     /* per subsequence: segment index local to its image */
  const hj_tables *tables;     /* [nimages] */
  const hj_wide_ac *wide;      /* 12-bit AC tables, or NULL: [nimages] of them (every round takes them), or — wide_shared — ONE set */
  int wide_shared;             /* for all images, which only the list rounds take (the dense rounds then run with the 9-bit ones) */
  const uint8_t *scan;         /* all images' entropy-coded bytes */
  uint64_t *S;                 /* states: nsub + nseg entries per image */
  uint64_t *last_in;           /* start state of each lane's latest run */
  uint32_t *R;                 /* blocks completed by each lane's latest run */
  uint32_t *B;                 /* blocks before the lane, within its segment */
  int16_t *dc_diff;            /* image i at dc_diff + i*dc_stride: DC difference of every block, scan order */
  int16_t *dc_val;             /* image i at dc_val + i*dc_stride: DC value of every block, by coefficient-buffer slot */
  long long dc_stride;         /* entries per image (>= coef_shorts/64) */
  uint32_t *blk_pos;           /* small batches: image i at blk_pos + i*dc_stride, bit position + 1 of every block start, scan order */
  int dc_chunks_per_image;     /* hj_dc_chunks_per_image() */
  uint32_t *scan_part;         /* chunk totals of the prefix-sum pass (hj_scan_part_bytes) */
  uint32_t *ran;               /* [HJ_MAX_ROUNDS] non-zero if any lane ran in that round (a list round: if it left work for the next) */
  uint32_t *list;              /* work lists of the list rounds (hj_sync_list): image-local subsequence numbers; round r reads the
                                * entries list[(r & 1)*list_stride + image.sub0 ...] and appends to those of parity (r + 1) & 1 */
  uint32_t list_stride;        /* entries per parity (>= the batch's subsequences) */
  uint32_t *list_count;        /* entries of round r's list of image i at [((r & 3)*nimages + i)*HJ_LIST_CSTRIDE] */
  uint32_t *errors;            /* [nimages] bit0 inconsistent stream, bit1 bad coefficient index */
  int16_t *coef;               /* image i at coef + i*coef_stride */
  long long coef_stride;
  int nimages;
  int sub_log2;                /* subsequence length = 1 << sub_log2 bytes (5..7) */
  int flush_lanes;             /* finished blocks a wave collects before writing them out */
} hj_args;

/* regions hj_launch_init clears: up to six, each `rows` runs of `row_bytes` bytes (multiples of 16, 16-byte aligned) at a stride */
typedef struct hj_clear_region { void *base; uint64_t row_bytes, stride; uint32_t rows, pad_; } hj_clear_region;
typedef struct hj_clear_args { hj_clear_region region[6]; int nregions; } hj_clear_args;

#ifdef __cplusplus
extern "C" {
#endif
/* fills sub_seg and the start states S (guesses) on the device */
/* ... and clears ran[] and sets errors[] (to verdicts0[], device memory, or to 0) */
/* (C: regions cleared by extra workgroups of the same launch, or NULL) */
int hj_launch_init(const hj_args *A, int total_segs, int max_nsub, const uint32_t *verdicts0, const hj_clear_args *C, void *stream);
/* lean != 0: the rows read through registers (the default; 0: the stateless row reader, an A/B knob) */
int hj_launch_round(const hj_args *A, int max_nsub, int round, int max_iters, int lean, void *stream);
/* A LIST round: only the subsequences whose start state moved run, packed into dense waves from a work list per image
 * (rebuild != 0: the lists are made afresh from the states first — the first list round of a decode, or after the host
 * changed states; ordinal: how many list rounds of this decode came before).  An image whose list fits one workgroup is
 * iterated inside it, up to max_iters steps. */
int hj_launch_list_round(const hj_args *A, int max_nsub, int round, int ordinal, int max_iters, int rebuild, void *stream);
int hj_launch_scan(const hj_args *A, int total_segs, int max_nsub, void *stream);
size_t hj_scan_part_bytes(size_t total_segs, size_t total_subs);
int hj_launch_write(const hj_args *A, int max_nsub, void *stream);
/* The write pass of a small batch: block starts (A->blk_pos, zeroed beforehand), then one lane per block. */
int hj_launch_write_blocks(const hj_args *A, int max_nsub, int blocks_per_image, void *stream);
/* DC differences (A->dc_diff, left by the write pass) -> DC values by buffer slot (A->dc_val);
 * apply_slots > 0: also written into the planes' DC positions (slots per image).  `part`:
 * 12 bytes x nimages x hj_dc_chunks_per_image(). */
int hj_dc_chunks_per_image(int total_mcus, int max_segs_per_image);
int hj_launch_dc(const hj_args *A, int total_segs, int max_seg_mcus, uint32_t *part, int apply_slots, void *stream);
#ifdef __cplusplus
}
#endif
#endif
