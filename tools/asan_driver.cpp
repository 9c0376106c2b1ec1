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
  int ok = 0, bad = 0;
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
    int r2 = jga_entropy_decode(buf.data(), (int)n, &g, coef.data(), 1);
    long long nidx = index_count(&g);
    std::vector<int> index((size_t)nidx);
    long long cap = g.coef_shorts + 64*nidx + 1024, nwords = 0;
    std::vector<short> pack((size_t)cap);
    int r3 = jga_entropy_decode_pack(buf.data(), (int)n, &g, pack.data(), cap, index.data(), &nwords, NULL);
    hj_prepared P;
    int r4 = hj_prepare_image(buf.data(), (int)n, &P);
    (r1 == 0 && r2 == 0 && r3 == 0 && r4 == 0) ? ok++ : bad++;
  }
  printf("files accepted by every host stage %d, rejected %d\n", ok, bad);
  return 0;
}
