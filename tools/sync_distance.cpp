// sync_distance.cpp — how far does a run that starts from a GUESS (a symbol boundary assumed at a
// subsequence's first bit, k = 0, slot 0) decode before it has fallen into step with the true
// symbol sequence?  Prints the distribution over all subsequence starts of one file, at a
// granularity of 8 bytes.  Analysis helper (host only), built like tools/huff_emul.cpp:
//   g++ -std=c++17 -O2 -Iinclude -o tools/bin/sync_distance tools/sync_distance.cpp \
//       jpeg_gpu_amd/csrc/huff_prepare.cpp tools/bin/obj/entropy.o tools/bin/obj/layout.o
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../jpeg_gpu_amd/csrc/huff_prepare.h"

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: sync_distance file.jpg [sub_bytes]\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<unsigned char> buf;
  { unsigned char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n); }
  fclose(f);
  const int sub = argc > 2 ? atoi(argv[2]) : 128;
  hj_prepared P;
  if (hj_prepare_image(buf.data(), (int)buf.size(), &P) != EXIT_SUCCESS) return 1;
  const uint32_t MARK = 8;                      // bytes between comparisons
  hj_mem_src src; src.s = P.clean.data();
  std::vector<long long> hist(4096/MARK + 2, 0);
  long long starts = 0;
  for (const hj_segment &sg : P.segs) {
    const uint32_t nmarks = (sg.end - sg.start + MARK - 1)/MARK;
    // truth: the state at the first symbol boundary at or past each mark
    std::vector<uint64_t> truth(nmarks + 1);
    uint64_t st = hj_pack((uint64_t)sg.start*8, 0, 0);
    truth[0] = st;
    for (uint32_t m = 1; m <= nmarks; m++) {
      uint64_t stop = (uint64_t)(sg.start + m*MARK)*8;
      if (stop > (uint64_t)sg.end*8) stop = (uint64_t)sg.end*8;
      st = hj_sync_decode(src, P.im, &P.tabs, st, stop, false).end_state;
      truth[m] = st;
    }
    for (uint32_t first = sg.start + sub; first + 16 < sg.end; first += sub) {
      const uint32_t m0 = (first - sg.start)/MARK;
      uint64_t g = hj_pack((uint64_t)first*8, 0, 0);
      uint32_t d = 0;
      bool synced = false;
      for (uint32_t m = m0 + 1; m <= nmarks && d < 4096/MARK; m++) {
        uint64_t stop = (uint64_t)(sg.start + m*MARK)*8;
        if (stop > (uint64_t)sg.end*8) stop = (uint64_t)sg.end*8;
        g = hj_sync_decode(src, P.im, &P.tabs, g, stop, false).end_state;
        d++;
        if (g == truth[m]) { synced = true; break; }
      }
      hist[synced ? d : 4096/MARK + 1]++;
      starts++;
    }
  }
  printf("%s: %lld guess starts every %d bytes, clean scan %u bytes\n", argv[1], starts, sub, P.scan_len);
  long long cum = 0;
  const uint32_t show[] = {8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048, 4096};
  size_t si = 0;
  for (uint32_t d = 1; d <= 4096/MARK; d++) {
    cum += hist[d];
    if (si < sizeof show/sizeof *show && d*MARK == show[si]) {
      printf("  in step within %4u bytes: %6.2f %%\n", d*MARK, 100.0*cum/starts);
      si++;
    }
  }
  printf("  not within 4096 (or segment ended): %.2f %%\n", 100.0*hist[4096/MARK + 1]/starts);
  return 0;
}
