// sync_chain_sim.cpp — how many SERIAL decode steps does the synchronisation of one file need,
// with and without a memo of runs made for every MCU-slot hypothesis?  Analysis helper (host only):
//   g++ -std=c++17 -O2 -Iinclude -o tools/bin/sync_chain_sim tools/sync_chain_sim.cpp \
//       jpeg_gpu_amd/csrc/huff_prepare.cpp tools/bin/obj/entropy.o tools/bin/obj/layout.o
// Model: every subsequence i has an incoming state S[i]; a synchronous "step" lets every subsequence
// whose incoming state changed decode once (128 bytes: the unit of latency) and hand its end state
// on.  A memo hit costs nothing and is followed within the same step.
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include "../jpeg_gpu_amd/csrc/huff_prepare.h"

struct sim {
  hj_prepared P;
  hj_mem_src src;
  int sub, lite_skip;
  uint64_t run(const hj_segment &sg, uint32_t i, uint64_t st, long long *decodes) {
    uint64_t stop = (uint64_t)(sg.start + (i + 1)*sub)*8;
    if (stop > (uint64_t)sg.end*8) stop = (uint64_t)sg.end*8;
    (*decodes)++;
    return hj_sync_decode(src, P.im, &P.tabs, st, stop, false).end_state;
  }
};

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: sync_chain_sim file.jpg [sub_bytes] [mode] [ahead]\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<unsigned char> buf;
  { unsigned char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n); }
  fclose(f);
  sim M;
  M.sub = argc > 2 ? atoi(argv[2]) : 128;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;      // 0: as today; 1: memo of all slot hypotheses at every run; 2: ... + for `ahead` successors
  const int ahead = argc > 4 ? atoi(argv[4]) : 0;
  M.lite_skip = 64;
  if (hj_prepare_image(buf.data(), (int)buf.size(), &M.P) != EXIT_SUCCESS) return 1;
  M.src.s = M.P.clean.data();
  const int nslots = M.P.im.nslots;
  long long decodes = 0, spec_decodes = 0, total_sub = 0;
  int worst_steps = 0;
  std::vector<long long> dirty_hist(64, 0);
  for (const hj_segment &sg : M.P.segs) {
    const uint32_t n = sg.nsub;
    total_sub += n;
    std::vector<uint64_t> S(n + 1), last_in(n, ~0ull);
    std::vector<std::map<uint64_t, uint64_t>> memo(n);
    for (uint32_t i = 0; i < n; i++) S[i] = hj_pack((uint64_t)(sg.start + i*M.sub)*8, 0, 0);
    // lite run: from the guess, part-way in
    std::vector<uint64_t> prop(n + 1);
    for (uint32_t i = 0; i < n; i++) {
      uint64_t stop = (uint64_t)(sg.start + (i + 1)*M.sub)*8;
      if (stop > (uint64_t)sg.end*8) stop = (uint64_t)sg.end*8;
      uint64_t from = hj_pos(S[i]), skip = (uint64_t)M.lite_skip*8;
      if (from + 2*skip > stop) skip = from < stop ? (stop - from)/2 : 0;
      prop[i + 1] = M.run(sg, i, hj_pack(from + skip, 0, 0), &decodes);
    }
    for (uint32_t i = 1; i < n; i++) S[i] = prop[i];
    // synchronous steps
    std::vector<uint8_t> dirty(n, 1);
    int steps = 0;
    for (;;) {
      std::vector<uint32_t> act;
      for (uint32_t i = 0; i < n; i++) if (dirty[i]) act.push_back(i);
      if (act.empty()) break;
      steps++;
      if (steps < 64) dirty_hist[steps] += (long long)act.size();
      struct res { uint32_t i; uint64_t in, out; };
      std::vector<res> results;
      for (uint32_t i : act) {
        dirty[i] = 0;
        const uint64_t in = S[i];
        uint64_t out;
        auto it = memo[i].find(in);
        if (it != memo[i].end()) out = it->second;       // (should not happen: hits are followed below)
        else {
          out = M.run(sg, i, in, &decodes);
          memo[i][in] = out;
          if (mode >= 1) {
            for (int c = 0; c < nslots; c++) {
              const uint64_t v = hj_pack(hj_pos(in), c, hj_k(in));
              if (!memo[i].count(v)) memo[i][v] = M.run(sg, i, v, &spec_decodes);
            }
          }
        }
        last_in[i] = in;
        results.push_back({i, in, out});
      }
      if (mode >= 2) {
        // speculate on the successors of what ran: their CURRENT incoming (p, k) with every slot
        for (uint32_t i : act) {
          for (int a = 1; a <= ahead && i + a < n; a++) {
            const uint64_t in = S[i + a];
            for (int c = 0; c < nslots; c++) {
              const uint64_t v = hj_pack(hj_pos(in), c, hj_k(in));
              if (!memo[i + a].count(v)) memo[i + a][v] = M.run(sg, i + a, v, &spec_decodes);
            }
          }
        }
      }
      // hand over; follow memo hits for free
      for (auto &r : results) {
        uint32_t i = r.i;
        uint64_t out = r.out;
        if (S[i] != r.in) continue;                      // its start state moved while it ran: stale
        while (i + 1 < n) {
          if (S[i + 1] == out) break;
          S[i + 1] = out;
          i++;
          auto it = memo[i].find(out);
          if (it == memo[i].end()) { dirty[i] = 1; break; }
          last_in[i] = out;
          dirty[i] = 0;
          out = it->second;
        }
      }
    }
    if (steps > worst_steps) worst_steps = steps;
    // check the fixed point against a straight decode
    uint64_t st = hj_pack((uint64_t)sg.start*8, 0, 0);
    long long dummy = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (S[i] != st) { printf("MISMATCH at %u\n", i); return 1; }
      st = M.run(sg, i, st, &dummy);
    }
  }
  printf("%s sub %d mode %d ahead %d: %lld subsequences, serial steps %d, decodes %.2f per subsequence (+ %.2f speculative)\n",
   argv[1], M.sub, mode, ahead, total_sub, worst_steps, (double)decodes/total_sub, (double)spec_decodes/total_sub);
  printf("  runs per step:");
  for (int s = 1; s < 64 && dirty_hist[s]; s++) printf(" %lld", dirty_hist[s]);
  printf("\n");
  return 0;
}
