// huff_prepare.h — host-side preparation of one JPEG for the GPU entropy stage:
// marker segments -> device-format Huffman tables, MCU slot map, restart segments.
#ifndef JGA_HUFF_PREPARE_H
#define JGA_HUFF_PREPARE_H (1)
#include <vector>
#include "huff_common.h"
#include "jga_internal.h"

struct hj_prepared {
  hj_image im;                      // scan_off / sub0 / seg0 are filled by the batch builder
  jga_geom geom;
  std::vector<hj_segment> segs;
  hj_table tabs[6];                 // [2*comp] DC, [2*comp+1] AC
  unsigned short qtab[3*64];        // per plane, natural order
  const unsigned char *scan;        // points into the caller's JPEG bytes
  uint32_t scan_len;                // entropy-coded bytes incl. RST markers, excl. EOI
};

// Returns EXIT_SUCCESS / EXIT_FAILURE (message via jga_fail).
int hj_prepare_image(const unsigned char *jpeg, int size, hj_prepared *out);

#endif
