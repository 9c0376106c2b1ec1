// ASAN/UBSAN driver over the HOST stages: parse, geometry, entropy decode (planes + PACK), GPU-stage prepare
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "huff_prepare.h"
extern "C" {
#include "jga_internal.h"
}
static long long index_count(const jga_geom *g) { long long n = 0; for (int p = 0; p < g->nplanes; p++) n += (long long)(g->plane[p].hblocks << g->plane[p].xdec)*g->plane[p].cstride; return n; }
int main(int argc, char **argv) {
  int ok = 0, bad = 0, bands_ok = 0, band_diff = 0;
  for (int a = 1; a < argc; a++) {
    FILE *f = fopen(argv[a], "rb"); if (!f) continue;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> buf((size_t)n);      // exact size: ASAN sees any over-read
    if (fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); continue; }
    fclose(f);
    jpeg_header h; jga_geom g;
    if (jga_parse_header(buf.data(), (int)n, &h) != 0 || jga_geom_from_header(&g, &h) != 0) { bad++; continue; }
    std::vector<short> coef((size_t)g.coef_shorts);
    int r1 = jga_entropy_decode(buf.data(), (int)n, &g, coef.data(), 0);
    std::vector<short> coef_q(coef);                 // QUANT planes (the band check below)
    int r2 = jga_entropy_decode(buf.data(), (int)n, &g, coef.data(), 1);
    long long nidx = index_count(&g);
    std::vector<int> index((size_t)nidx);
    long long cap = g.coef_shorts + 64*nidx + 1024, nwords = 0;
    std::vector<short> pack((size_t)cap);
    int r3 = jga_entropy_decode_pack(buf.data(), (int)n, &g, pack.data(), cap, index.data(), &nwords, NULL);
    hj_prepared P;
    int r4 = hj_prepare_image(buf.data(), (int)n, &P);
    // round 4: the frame in bands (csrc/band.c) — every band written into a buffer of exactly its size, then
    // through the header parse and the entropy stage like any file
    jga_band bands[5];
    const int nb = jga_band_plan(buf.data(), n, 5, bands);
    for (int b = 0; b < nb; b++) {
      const long len = jga_band_file(buf.data(), n, &bands[b], NULL, 0);
      if (len <= 0) continue;
      std::vector<unsigned char> bf((size_t)len);
      if (jga_band_file(buf.data(), n, &bands[b], bf.data(), len) != len) { printf("band length changed: %s\n", argv[a]); return 1; }
      jpeg_header hb; jga_geom gb;
      if (jga_parse_header(bf.data(), (int)len, &hb) != 0 || jga_geom_from_header(&gb, &hb) != 0) continue;
      std::vector<short> cb((size_t)gb.coef_shorts);
      const int rb = jga_entropy_decode(bf.data(), (int)len, &gb, cb.data(), 0);
      if (rb == 0 && r1 == 0) bands_ok++;
      // an undamaged frame's band holds the frame's own blocks: first block of its luma plane
      if (rb == 0 && r1 == 0 && memcmp(cb.data(), coef_q.data() + jga_block_offset(&g, 0, 0, bands[b].mcu_row0*h.comp[0].vsamp), 128) != 0) band_diff++;
    }
    (r1 == 0 && r2 == 0 && r3 == 0 && r4 == 0) ? ok++ : bad++;
  }
  printf("files accepted by every host stage %d, rejected %d; bands decoded %d, of which differ from their frame's blocks %d\n", ok, bad, bands_ok, band_diff);
  return 0;
}
