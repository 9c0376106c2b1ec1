/* jga_internal.h — declarations shared by the host-side C/C++ sources. */
#ifndef JGA_INTERNAL_H
#define JGA_INTERNAL_H (1)

#include "../../include/jpeg_gpu_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define JGA_EXPORT __attribute__((visibility("default")))

#include "jga_tune.h"

/* Record an error (thread-local), print it to stderr unless JGA_QUIET is set,
 * return EXIT_FAILURE. */
int jga_fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int jga_ilog(unsigned v);
int jga_cpu_budget(void);        /* affinity mask cut down to the cgroup's cpu.max grant */
int jga_subsamp_of(int xdec, int ydec, int ncomps);

/* Entropy decode with an explicit stage (entropy.c). */
enum { JGA_STAGE_PACK = 0, JGA_STAGE_QUANT = 1, JGA_STAGE_DCT = 2 };

/* What the GPU entropy stage (huff_api.cpp) needs from the marker segments. */
typedef struct jga_scan_desc {
  jpeg_header header;
  int scan_off;                 /* first byte of entropy-coded data */
  int td[3], ta[3];             /* DC / AC table selectors per component (SOS) */
  int dht_valid[8];             /* index tc*4 + th */
  unsigned char dht_bits[8][16];
  unsigned char dht_vals[8][256];
} jga_scan_desc;
int jga_scan_describe(const unsigned char *buf, int size, jga_scan_desc *d);

#ifdef __cplusplus
}
#endif
#endif
