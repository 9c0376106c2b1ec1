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
  hj_tables tabs;                   // two-level LUTs, [2*comp] DC, [2*comp+1] AC
  unsigned short qtab[3*64];        // per plane, natural order
  std::vector<unsigned char> clean; // entropy-coded bytes with FF00 -> FF and RSTn removed, + 16 pad
  uint32_t scan_len;                // clean length (without the pad)
  uint32_t raw_len;                 // entropy-coded bytes in the file incl. stuffing and RST markers
};

// Returns EXIT_SUCCESS / EXIT_FAILURE (message via jga_fail).
int hj_prepare_image(const unsigned char *jpeg, int size, hj_prepared *out);

#endif
