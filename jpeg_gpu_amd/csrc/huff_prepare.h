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
  std::vector<uint32_t> wide;       // the 12-bit AC tables (hj_wide_ac: 2 << HJ_WIDE_BITS entries), after hj_prepare_wide
  unsigned short qtab[3*64];        // per plane, natural order
  std::vector<unsigned char> clean; // (hj_prepare_image only) clean bytes + 16 pad
  uint32_t scan_len;                // clean length (without the pad)
  uint32_t raw_len;                 // entropy-coded bytes in the file incl. stuffing and RST markers
  // between the two steps:
  jga_scan_desc *desc;              // marker segments (owned; freed by hj_prepare_scan / _drop)
  uint32_t avail;                   // bytes from the start of the scan to the end of the file
  int sub_log2;                     // subsequence length hj_prepare_scan cuts the segments into (set by the caller)
  hj_prepared() : scan_len(0), raw_len(0), desc(nullptr), avail(0), sub_log2(HJ_SUB_LOG2_MAX) {}
  ~hj_prepared();
  hj_prepared(const hj_prepared &) = delete;
  hj_prepared &operator=(const hj_prepared &) = delete;
};

// Two steps, so that a batch builder can size one pinned buffer between them:
//   hj_prepare_head  marker segments -> geometry, slot map, tables, quantisers, `avail`
//   hj_prepare_scan  entropy-coded bytes -> clean stream written to `dst` (capacity >=
//                    avail + 16; 16 pad bytes of 0xFF follow the stream) + restart segments
// Both return EXIT_SUCCESS / EXIT_FAILURE (message via jga_fail); hj_prepare_head returns
// HJ_PREPARE_IRREGULAR for a valid file the device format cannot hold — Huffman tables that
// need more level-2 blocks than hj_tables has, or a frame beyond the 32-bit bit positions /
// plane offsets of the kernels (the caller may then use the host entropy stage).
#define HJ_PREPARE_IRREGULAR 2
int hj_prepare_head(const unsigned char *jpeg, int size, hj_prepared *out);
// After hj_prepare_head: the wide AC tables of out->tabs into out->wide (for batches that take them: huff_common.h).
void hj_prepare_wide(hj_prepared *out);
int hj_prepare_scan(const unsigned char *jpeg, int size, hj_prepared *out, unsigned char *dst);
// Both steps, clean stream kept in out->clean (emulation / tests).
int hj_prepare_image(const unsigned char *jpeg, int size, hj_prepared *out);

// Host walk over the stretches of ONE image that have not settled (streams that do not
// self-synchronise: flat areas, letterbox bars).  `S` / `last_in` are the image's slices of the
// state arrays read back from the device (S: nsub + nseg entries, segment s of the image at
// S + sg.sub0 + s; last_in: nsub entries), `clean` its clean scan.  In every segment, from each
// lane whose start state moved since its last run (the first such lane's state is true by
// induction from the segment start) it decodes on with hj_sync_decode, writing the state at
// every subsequence boundary, until it arrives in a state the next lane has already run from.
// Returns the number of subsequences walked.
int hj_walk_unsettled(const hj_image &im, const hj_segment *segs, const hj_tables *tabs,
 const unsigned char *clean, uint64_t *S, const uint64_t *last_in, int sub_log2);

#endif
