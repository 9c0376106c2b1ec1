// huff_prepare.cpp — see huff_prepare.h.  Replaces, for the GPU entropy stage, the
// table set-up of the reference's host decoder (DHT codeword generation
// src/xjpeg.c:293-336, MCU structure xjpeg_mcu_init 431-446, restart bookkeeping
// 593-629): tables are built once on the host, the scan itself is decoded on the GPU.
#include <stdlib.h>
#include <string.h>
#include "huff_prepare.h"

static int build_table(hj_table *t, const unsigned char bits[16], const unsigned char *vals) {
  unsigned code = 0;
  int k = 0;
  unsigned char size[256];
  unsigned short codes[256];
  memset(t, 0, sizeof(*t));
  for (int len = 1; len <= 16; len++) {
    for (int i = 0; i < bits[len - 1]; i++) {
      if (k >= 256) return 1;
      size[k] = (unsigned char)len;
      codes[k] = (unsigned short)code;
      t->sym[k] = vals[k];
      k++;
      code++;
    }
    if (code > (1u << len)) return 1;
    t->delta[len] = k - (int)code;
    t->maxcode[len] = code << (16 - len);
    code <<= 1;
  }
  t->maxcode[17] = 0xFFFFFFFFu;
  for (int i = 0; i < k; i++) {
    const int s = size[i];
    if (s <= HJ_FAST_BITS) {
      const unsigned c = (unsigned)codes[i] << (HJ_FAST_BITS - s);
      for (unsigned j = 0; j < (1u << (HJ_FAST_BITS - s)); j++) {
        t->fast[c + j] = (uint16_t)((s << 8) | t->sym[i]);
      }
    }
  }
  return 0;
}

int hj_prepare_image(const unsigned char *jpeg, int size, hj_prepared *out) {
  jga_scan_desc *d = (jga_scan_desc *)malloc(sizeof(jga_scan_desc));
  if (!d) return jga_fail("Out of memory");
  if (jga_scan_describe(jpeg, size, d) != EXIT_SUCCESS
   || jga_geom_from_header(&out->geom, &d->header) != EXIT_SUCCESS) {
    free(d);
    return EXIT_FAILURE;
  }
  const jga_geom &g = out->geom;
  hj_image &im = out->im;
  memset(&im, 0, sizeof(im));
  int slot = 0;
  for (int c = 0; c < g.nplanes; c++) {
    const jpeg_component &cp = d->header.comp[c];
    im.comp_hs[c] = (uint8_t)cp.hsamp;
    im.comp_vs[c] = (uint8_t)cp.vsamp;
    im.comp_xdec[c] = (uint8_t)g.plane[c].xdec;
    im.comp_coef_off[c] = g.plane[c].coef_off;
    for (int sy = 0; sy < cp.vsamp; sy++) {
      for (int sx = 0; sx < cp.hsamp; sx++) {
        if (slot >= HJ_MAX_SLOTS) { free(d); return jga_fail("Unsupported sampling (MCU too large)"); }
        im.slot_comp[slot] = (uint8_t)c;
        im.slot_sbx[slot] = (uint8_t)sx;
        im.slot_sby[slot] = (uint8_t)sy;
        slot++;
      }
    }
    if (build_table(&out->tabs[2*c], d->dht_bits[d->td[c]], d->dht_vals[d->td[c]])
     || build_table(&out->tabs[2*c + 1], d->dht_bits[4 + d->ta[c]], d->dht_vals[4 + d->ta[c]])) {
      free(d);
      return jga_fail("Error invalid DHT.");
    }
    memcpy(out->qtab + 64*c, cp.quant->tbl, 64*sizeof(unsigned short));
  }
  for (int c = g.nplanes; c < 3; c++) memset(out->qtab + 64*c, 0, 64*sizeof(unsigned short));
  im.nslots = slot;
  im.nhmb = g.nhmb;
  im.w0_blocks = g.w0/8;

  // entropy-coded bytes: split at RSTn markers, stop at the first other marker
  const unsigned char *scan = jpeg + d->scan_off;
  const uint32_t avail = (uint32_t)(size - d->scan_off);
  const uint32_t total_mcus = (uint32_t)g.nhmb*(uint32_t)g.nvmb;
  const uint32_t ri = (uint32_t)d->header.restart_interval;
  out->scan = scan;
  out->segs.clear();
  uint32_t seg_start = 0, pos = 0, mcu0 = 0, nsub = 0;
  int expect = 0;
  bool done = false;
  while (!done) {
    const unsigned char *ff = pos < avail ? (const unsigned char *)memchr(scan + pos, 0xFF, avail - pos) : NULL;
    uint32_t at = ff ? (uint32_t)(ff - scan) : avail;
    int marker = (ff && at + 1 < avail) ? scan[at + 1] : 0xD9;     // running off the end == EOI
    if (ff && marker == 0x00) { pos = at + 2; continue; }          // stuffed zero
    if (ff && marker == 0xFF) { pos = at + 1; continue; }          // fill byte
    // a real marker (or the end of the buffer) closes the current segment
    hj_segment s;
    s.start = seg_start;
    s.end = at;
    s.sub0 = nsub;
    s.nsub = (s.end - s.start + HJ_SUB_BYTES - 1)/HJ_SUB_BYTES;
    if (s.nsub == 0) s.nsub = 1;
    s.mcu0 = mcu0;
    s.nmcu = ri ? (total_mcus - mcu0 < ri ? total_mcus - mcu0 : ri) : total_mcus;
    out->segs.push_back(s);
    nsub += s.nsub;
    mcu0 += s.nmcu;
    if (marker >= 0xD0 && marker <= 0xD7 && ri && mcu0 < total_mcus) {
      if (marker != 0xD0 + (expect & 7)) { free(d); return jga_fail("Error invalid RST counter in marker."); }
      expect++;
      seg_start = pos = at + 2;
    }
    else done = true;
    out->scan_len = at;
  }
  free(d);
  if (mcu0 != total_mcus) return jga_fail("Error, entropy data ended early.");
  im.nsub = nsub;
  im.nseg = (uint32_t)out->segs.size();
  im.scan_len = out->scan_len;
  return EXIT_SUCCESS;
}
