/* unstuff_kernels.h — launch arguments of the on-device scan clean-up (unstuff_kernels.hip). */
#ifndef JGA_UNSTUFF_KERNELS_H
#define JGA_UNSTUFF_KERNELS_H (1)
#include <stddef.h>
#include "huff_common.h"

#define HJ_UNSTUFF_CHUNK 16384       /* raw bytes per workgroup (256 threads x 4 groups of 16) */

/* What the host knows about one image's entropy-coded bytes before anyone has looked at them. */
typedef struct hj_unstuff_image {
  uint32_t raw_off;            /* offset of its raw scan bytes in the batch's raw region (16-aligned) */
  uint32_t avail;              /* bytes from the start of the scan to the end of the file */
  uint32_t nseg;               /* restart intervals the frame must have: ceil(MCUs / DRI), or 1 */
  uint32_t ri;                 /* DRI in MCUs, 0 = none */
  uint32_t total_mcus;
  uint32_t chunk0;             /* its first entry of `part` */
  uint32_t nchunks;
  uint32_t pad_;
} hj_unstuff_image;

/* per image, device-only */
typedef struct hj_unstuff_info {
  uint32_t hard_end;           /* position of the first marker that is not RSTn (or `avail`) */
  uint32_t end;                /* where the scan ends: that, or the RSTn one too many */
  uint32_t found;              /* RSTn markers accepted */
  uint32_t scan_len;           /* clean bytes */
} hj_unstuff_info;

typedef struct hj_unstuff_args {
  const uint8_t *raw;          /* batch raw region */
  uint8_t *clean;              /* batch clean region: image i at images[i].scan_off */
  hj_image *images;            /* scan_len and nsub are filled in here */
  hj_segment *segs;            /* image i's at images[i].seg0 */
  const hj_unstuff_image *uimg;
  uint32_t *part;              /* [chunks][2]: kept bytes, RSTn markers per chunk -> exclusive prefixes */
  uint32_t *bnd;               /* per segment: clean position of the RSTn that closes it */
  hj_unstuff_info *info;
  uint32_t *errors;            /* [nimages] bit 0: the stream ends early / bad RSTn counter */
  int nimages;
  int sub_log2;
} hj_unstuff_args;

#ifdef __cplusplus
extern "C" {
#endif
/* raw scans (stuffed, with RSTn markers) -> clean streams + restart segments + per-image
 * lengths, all on `stream`; max_chunks = the largest nchunks of the batch */
int hj_launch_unstuff(const hj_unstuff_args *A, int max_chunks, void *stream);
#ifdef __cplusplus
}
#endif
#endif
