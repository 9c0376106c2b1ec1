// huff_kernels.hip — GPU-parallel baseline-JPEG entropy decode (gfx950).
// Algorithm and state definitions: huff_common.h.  The kernels:
//   hj_init_states start states (guesses), segment numbers, "never ran" marks, bookkeeping words
//   hj_sync_round  one lane per subsequence; re-decodes when its start state moved.
//                  A workgroup iterates internally (states handed lane-to-lane
//                  through LDS) until none of its lanes moves, so most of the
//                  propagation needs no extra launch.
//   hj_list_build / hj_sync_list   the later rounds of batches that fill or share the device: only the
//                  subsequences that still move run, from per-image work lists (round 5; the one-wave-per-256
//                  hj_sync_sparse of rounds 2-4 was removed in round 6: slower wherever lists apply, and the
//                  batches without lists keep the dense kernel)
//   hj_scan        per restart segment: exclusive prefix sums of the block counts over the
//                  segment's lanes; the final launch also zeroes the line of the block every
//                  lane starts inside (the only lines the write pass stores piecewise)
//   hj_write       one lane per subsequence: final decode.  Blocks are assembled in LDS;
//                  a lane that finishes one waits until enough lanes of its wave have,
//                  then the wave writes them out together as full 128-byte lines.  Only
//                  the pieces of blocks that straddle lanes are scattered.  DC DIFFERENCES
//                  go to a compact array in scan order
//   hj_dc_scan     per restart segment and component: prefix sums over that array = the DC
//                  values (xjpeg.c:480), stored by coefficient-buffer slot
//   hj_dc_apply    writes them into the planes (callers that want finished QUANT planes;
//                  the block-decode kernels can take the array itself)
// Integer/byte work.  Every subsequence of a workgroup gets its own LDS copy — 35
// big-endian dwords (alignment + 128 bytes + look-ahead) at an odd stride, so lanes
// walking their own copies hit different banks — next to the image's two-level Huffman
// lookup (12 KB with the DC entries widened: hj_ltables).  LDS is what bounds occupancy: 53 440 B
// per sync workgroup (3 per CU — 54 208 B did not fit three times), 81 344 B per write workgroup
// (2 per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "huff_common.h"
#include "huff_kernels.h"
#include "jga_tune.h"

#define HJ_BLOCK 256
// staged scan bytes: 256 subsequences x 35 dwords (3 alignment + 128 + look-ahead bytes)
// A staged subsequence row: its bytes from (start & ~3) plus what the last symbol may reach
// past the end, (1 << sub_log2)/4 + 3 dwords = 11, 19 or 35 (odd: lanes walking their own rows
// hit different banks).
#define HJ_SUB_DWORDS_MAX (HJ_SUB_BYTES_MAX/4 + 3)
#define HJ_WIN_DWORDS (HJ_BLOCK*HJ_SUB_DWORDS_MAX + 8)
static __device__ __forceinline__ uint32_t hj_sub_dwords(const hj_args &A) { return (1u << (A.sub_log2 - 2)) + 3u; }
#define HJ_BLK_STRIDE 33            /* dwords per lane's block buffer (32 + 1 against conflicts) */

__device__ const uint8_t HJ_DEZZ[64] = {     // T.81 Figure A.6: zig-zag index -> natural index
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
  13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59,
  52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Bit source over one subsequence staged in LDS: its 128 bytes + look-ahead as
// big-endian dwords (byte-swapped once when staged).
struct hj_lds_src {
  const uint32_t *base;              // first dword of this subsequence's copy
  uint32_t bit0;                     // clean-scan bit position of that dword
  // Reader in the copy's own coordinates: r1 = (position - bit0) - 1.  With q = the dword
  // holding bit r1, the 32 bits that start at bit r1 + 1 are ({q[0], q[1]} >> (31 - r1 % 32)),
  // one v_alignbit_b32 whose shift operand is simply ~r1 (the instruction reads 5 bits) —
  // a shift of 0..31 in every case, including the dword-aligned one that a plain
  // "shift by 32 - offset" formulation cannot express.
  typedef void has_reader;
  struct reader {
    const uint32_t *base;
    uint32_t bit0;
    int32_t r1, stop1;
    __device__ __forceinline__ void init(const hj_lds_src &src, uint64_t pos, uint64_t stop_bit) {
      base = src.base; bit0 = src.bit0;
      r1 = (int32_t)((uint32_t)pos - bit0) - 1;
      stop1 = (int32_t)((uint32_t)stop_bit - bit0) - 1;
    }
    __device__ __forceinline__ bool before_stop() const { return r1 < stop1; }
    __device__ __forceinline__ bool room9() const { return r1 + 9 <= stop1; }
    __device__ __forceinline__ bool room(int n) const { return r1 + n <= stop1; }
    __device__ __forceinline__ uint32_t window() const {
      const uint32_t *q = base + (r1 >> 5);                  // r1 = -1: the dword before (unused bits)
      return __builtin_amdgcn_alignbit(q[0], q[1], ~(uint32_t)r1);
    }
    __device__ __forceinline__ void skip(int n) { r1 += n; }
    __device__ __forceinline__ uint64_t tell() const { return (uint64_t)((uint32_t)(r1 + 1) + bit0); }
  };
};

// The same LDS row read with the two dwords under the position in registers and the one after
// them already on its way, so a symbol's chain of dependent operations holds one LDS round trip
// (the table lookup), not two: a few more instructions per symbol, no branch.  What the dense
// kernel runs with (a lone frame's ~9 runs one after the other: 94 -> 83 us per launch; 48 x 4K:
// 2.00 -> 1.95 ms in an interleaved A/B); the stateless reader above stays for comparison
// (JGA_HUFF_LEAN=0).
struct hj_lds_reg_src {
  const uint32_t *base;
  uint32_t bit0;
  typedef void has_reader;
  struct reader {
    const uint32_t *base;
    uint32_t bit0;
    int32_t r1, stop1, d;
    uint32_t w0, w1, w2;
    __device__ __forceinline__ void init(const hj_lds_reg_src &src, uint64_t pos, uint64_t stop_bit) {
      base = src.base; bit0 = src.bit0;
      r1 = (int32_t)((uint32_t)pos - bit0) - 1;
      stop1 = (int32_t)((uint32_t)stop_bit - bit0) - 1;
      d = r1 >> 5;                                           // -1: the dword before the row (its bits are never used)
      w0 = base[d]; w1 = base[d + 1]; w2 = base[d + 2];
    }
    __device__ __forceinline__ bool before_stop() const { return r1 < stop1; }
    __device__ __forceinline__ bool room9() const { return r1 + 9 <= stop1; }
    __device__ __forceinline__ bool room(int n) const { return r1 + n <= stop1; }
    __device__ __forceinline__ uint32_t window() const { return __builtin_amdgcn_alignbit(w0, w1, ~(uint32_t)r1); }
    __device__ __forceinline__ void skip(int n) {
      r1 += n;
      const int32_t nd = r1 >> 5;                            // a symbol is at most 31 bits: one dword further at most
      const bool cross = nd != d;
      const uint32_t nx = base[nd + 2];                      // (read every time, used two crossings later; the rows
      w0 = cross ? w1 : w0;                                  //  are followed by spare dwords)
      w1 = cross ? w2 : w1;
      w2 = nx;
      d = nd;
    }
    __device__ __forceinline__ uint64_t tell() const { return (uint64_t)((uint32_t)(r1 + 1) + bit0); }
  };
};

// Bit source straight from the clean scan in global memory, for the write pass: the lane
// keeps the two dwords under its position in registers and a third one in flight, and only
// issues a (lane-divergent, L1-resident) dword load when its position crosses into the next
// dword — about every fifth symbol.  No LDS copy of the scan, so the write pass's LDS budget
// goes to the block buffers alone and more workgroups fit a CU.
struct hj_gmem_src {
  const uint32_t *scan32;            // the image's clean scan as dwords (wave-uniform address)
  uint32_t dw0;                      // dword holding the subsequence's first byte
  uint32_t ndw;                      // dwords readable from scan32 on (scan + its 16 pad bytes)
  typedef void has_reader;
  struct reader {
    const uint32_t *scan32;
    uint32_t dw0, bit0;
    int32_t r1, stop1, d;            // positions relative to bit0 = 32*dw0, minus one (as hj_lds_src)
    uint32_t w0, w1, w2raw;            // w2raw: the dword after w1 as loaded (not yet byte-swapped)
    // The dword two ahead is requested at a crossing and first TOUCHED at the next one (the
    // byte swap waits for the load): a run never stalls on the load it has just issued.
    __device__ __forceinline__ void init(const hj_gmem_src &src, uint64_t pos, uint64_t stop_bit) {
      scan32 = src.scan32; dw0 = src.dw0; bit0 = src.dw0 << 5;
      r1 = (int32_t)((uint32_t)pos - bit0) - 1;
      stop1 = (int32_t)((uint32_t)stop_bit - bit0) - 1;
      d = r1 >> 5;
      const uint32_t last = src.ndw - 1u;
      const uint32_t i0 = d < 0 ? dw0 : dw0 + (uint32_t)d;         // (the dword before bit 0 is never looked at)
      w0 = __builtin_bswap32(scan32[i0 < last ? i0 : last]);
      w1 = __builtin_bswap32(scan32[dw0 + (uint32_t)(d + 1) < last ? dw0 + (uint32_t)(d + 1) : last]);
      w2raw = scan32[dw0 + (uint32_t)(d + 2) < last ? dw0 + (uint32_t)(d + 2) : last];
    }
    __device__ __forceinline__ bool before_stop() const { return r1 < stop1; }
    __device__ __forceinline__ bool room9() const { return r1 + 9 <= stop1; }
    __device__ __forceinline__ bool room(int n) const { return r1 + n <= stop1; }
    __device__ __forceinline__ uint32_t window() const { return __builtin_amdgcn_alignbit(w0, w1, ~(uint32_t)r1); }
    __device__ __forceinline__ void skip(int n) {
      r1 += n;
      const int32_t nd = r1 >> 5;
      if (nd != d) {                 // a symbol is at most 31 bits: one dword further at most
        d = nd; w0 = w1; w1 = __builtin_bswap32(w2raw);
        // no clamp: a run stops within 31 bits of its subsequence's end, and 16 pad bytes
        // follow every image's scan in the batch buffer
        w2raw = scan32[dw0 + (uint32_t)(d + 2)];
      }
    }
    __device__ __forceinline__ uint64_t tell() const { return (uint64_t)((uint32_t)(r1 + 1) + bit0); }
  };
};

struct hj_lane_ctx {                 // what a lane knows about its subsequence
  uint32_t g, si, i;                 // batch-global subsequence, image-local segment, index in segment
  uint32_t seg_start, seg_end, seg_nsub, seg_mcu0, seg_nmcu;
  uint32_t stop_byte;                // end of this subsequence (raw byte, exclusive)
};

// Image descriptor -> LDS, dword by dword (a struct assignment through a private copy
// would be promoted to 84 B x 256 lanes of LDS by the compiler).
static __device__ __forceinline__ void hj_stage_image(hj_image *dst, const hj_image *src) {
  static_assert(sizeof(hj_image) % 4 == 0, "hj_image is copied as dwords");
  if (threadIdx.x < sizeof(hj_image)/4) {
    reinterpret_cast<uint32_t *>(dst)[threadIdx.x] = reinterpret_cast<const uint32_t *>(src)[threadIdx.x];
  }
}

// Common prologue: lane context, tables and the group's subsequences staged in LDS.
// Returns false for lanes beyond the image's last subsequence (they still took part
// in staging).  lds_start[t] receives the clean-scan byte offset of subsequence t.
// hj_tables (as uploaded: 16-bit DC entries) -> hj_ltables in LDS
template <int NB>
static __device__ __forceinline__ void hj_stage_tables(hj_ltables *dst, const hj_tables *src) {
  for (int k = threadIdx.x; k < 2 << HJ_FAST_BITS; k += NB) {
    (&dst->tab[0][0])[k] = (&src->dc[0][0])[k];
    (&dst->tab[2][0])[k] = (&src->ac[0][0])[k];
  }
  const uint4 *l2s = reinterpret_cast<const uint4 *>(src->l2);
  uint4 *l2d = reinterpret_cast<uint4 *>(dst->l2);
  for (int k = threadIdx.x; k < (int)(sizeof(src->l2)/16); k += NB) l2d[k] = l2s[k];
}

// hj_tables + the image's wide AC tables (hj_wide_ac) -> hj_ltables_wide in LDS: 40 KB, sixteen bytes per lane and trip
template <int NB>
static __device__ __forceinline__ void hj_stage_tables(hj_ltables_wide *dst, const hj_tables *src, const hj_wide_ac *wide) {
  for (int k = threadIdx.x; k < 2 << HJ_FAST_BITS; k += NB) (&dst->dc[0][0])[k] = (&src->dc[0][0])[k];
  const uint4 *ws = reinterpret_cast<const uint4 *>(wide->ac);
  uint4 *wd = reinterpret_cast<uint4 *>(dst->ac);
  for (int k = threadIdx.x; k < (int)(sizeof(wide->ac)/16); k += NB) wd[k] = ws[k];
  const uint4 *l2s = reinterpret_cast<const uint4 *>(src->l2);
  uint4 *l2d = reinterpret_cast<uint4 *>(dst->l2);
  for (int k = threadIdx.x; k < (int)(sizeof(src->l2)/16); k += NB) l2d[k] = l2s[k];
}
template <int NB>
static __device__ __forceinline__ void hj_stage_tables(hj_ltables *dst, const hj_tables *src, const hj_wide_ac *) {
  hj_stage_tables<NB>(dst, src);
}

template <bool STAGE_ROWS = true, int NB = HJ_BLOCK, class Tab = hj_ltables>
static __device__ __forceinline__ bool hj_prologue(const hj_args &A, const hj_image &im,
 Tab *lds_tabs, uint32_t *lds_win, uint16_t *lds_start, uint32_t *lds_start0, hj_lane_ctx &L) {
  const uint32_t li = blockIdx.x*NB + threadIdx.x;
  const bool in_range = li < im.nsub;
  L.g = 0; L.si = 0; L.i = 0; L.stop_byte = 0;
  L.seg_start = L.seg_end = L.seg_nsub = L.seg_mcu0 = L.seg_nmcu = 0;
  uint32_t my_start = 0;
  if (in_range) {
    L.g = im.sub0 + li;
    L.si = A.sub_seg[L.g];
    const hj_segment sg = A.segs[im.seg0 + L.si];
    L.i = li - sg.sub0;
    L.seg_start = sg.start; L.seg_end = sg.end; L.seg_nsub = sg.nsub;
    L.seg_mcu0 = sg.mcu0; L.seg_nmcu = sg.nmcu;
    my_start = sg.start + (L.i << A.sub_log2);
    L.stop_byte = my_start + (1u << A.sub_log2);
    if (L.stop_byte > sg.end) L.stop_byte = sg.end;
  }
  // (relative to the group's first subsequence: 256 x 128 bytes fit 16 bits, and the 512 bytes of
  // LDS this saves keep the round at three workgroups per CU)
  const uint32_t base0 = (uint32_t)__shfl((int)my_start, 0);   // wave 0 holds lane 0 of the group
  if (threadIdx.x == 0) *lds_start0 = base0;
  hj_stage_tables<NB>(lds_tabs, A.tables + blockIdx.y, A.wide && !A.wide_shared ? A.wide + blockIdx.y : nullptr);
  __syncthreads();
  lds_start[threadIdx.x] = in_range ? (uint16_t)(my_start - *lds_start0) : (uint16_t)0;
  __syncthreads();
  // subsequence t: hj_sub_dwords() dwords from (start & ~3)
  const uint8_t *scan = A.scan + im.scan_off;
  const uint32_t padded = (im.scan_len + 16 + 15) & ~15u;      // bytes present in the batch buffer
  const uint32_t nsubs = im.nsub - blockIdx.x*NB < NB ? im.nsub - blockIdx.x*NB : NB;
  const uint32_t sdw = hj_sub_dwords(A);
  const uint32_t magic = 0xFFFFFFFFu/sdw + 1u;                 // c/sdw == umulhi(c, magic) for c < 2^16
  for (uint32_t c = threadIdx.x; STAGE_ROWS && c < nsubs*sdw; c += NB) {
    const uint32_t sub = __umulhi(c, magic), d = c - sub*sdw;
    uint32_t a = ((*lds_start0 + lds_start[sub]) & ~3u) + 4*d;
    if (a + 4 > padded) a = padded - 4;
    lds_win[sub*sdw + d] = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(scan + a));
  }
  __syncthreads();
  return in_range;
}

// Bit source of subsequence `sub` of the group.
template <class RowSrc>
static __device__ __forceinline__ RowSrc hj_source(const uint32_t *lds_win,
 const uint16_t *lds_start, uint32_t start0, uint32_t sub, uint32_t sdw) {
  RowSrc s;
  s.base = lds_win + sub*sdw;
  s.bit0 = ((start0 + lds_start[sub]) & ~3u) << 3;
  return s;
}


template <class RowSrc, class Tab = hj_ltables>
__global__ __launch_bounds__(HJ_BLOCK) void hj_sync_round(const hj_args A, int round, int max_iters, int lite_first) {
  __shared__ __attribute__((aligned(16))) Tab lds_tabs;
  __shared__ uint32_t lds_win_mem[1 + HJ_WIN_DWORDS];      // [0]: the dword "before" row 0 (hj_lds_src::reader)
  uint32_t *lds_win = lds_win_mem + 1;
  __shared__ uint64_t lds_S[HJ_BLOCK + 1];       // start state of each subsequence of the group
  __shared__ uint16_t lds_R[HJ_BLOCK];           // blocks completed by its latest run (the end state lives in lds_S / S)
  __shared__ uint32_t lds_stop[HJ_BLOCK];        // stop byte | bit 31: has a successor in its segment
  __shared__ uint32_t lds_sidx_last;             // entry of the group's last subsequence in the global S array
  __shared__ uint8_t lds_dirty[HJ_BLOCK], lds_ran[HJ_BLOCK];
  __shared__ uint8_t lds_act[HJ_BLOCK];
  __shared__ uint32_t lds_wcnt[HJ_BLOCK/64];
  __shared__ uint16_t lds_start[HJ_BLOCK];
  __shared__ uint32_t lds_start0;
  __shared__ hj_image s_im;
  const hj_image im = A.images[blockIdx.y];      // scalar fields only; indexed ones via s_im
  const uint32_t t = threadIdx.x;
  {
    // cheap exit before any staging: did any lane's start state move since its last run?
    const uint32_t li = blockIdx.x*HJ_BLOCK + t;
    bool need = false;
    if (li < im.nsub) {
      const uint32_t g = im.sub0 + li;
      need = A.S[g + im.seg0 + A.sub_seg[g]] != A.last_in[g];
    }
    if (!__syncthreads_or(need)) return;
  }
  hj_stage_image(&s_im, A.images + blockIdx.y);
  hj_lane_ctx L;
  const bool on = hj_prologue<true, HJ_BLOCK, Tab>(A, im, &lds_tabs, lds_win, lds_start, &lds_start0, L);
  const uint32_t start0 = lds_start0;
  const uint32_t sidx = L.g + im.seg0 + L.si;          // this subsequence's entry of S
  {
    const uint64_t st = on ? A.S[sidx] : 0;
    lds_S[t] = st;
    lds_dirty[t] = on && st != A.last_in[L.g];
    lds_ran[t] = 0;
    lds_stop[t] = L.stop_byte | (on && L.i + 1 < L.seg_nsub ? 0x80000000u : 0u);
    if (t == HJ_BLOCK - 1) lds_sidx_last = sidx;
  }
  __syncthreads();
  // Iterate inside the group.  Each iteration packs the subsequences whose start state
  // moved into dense waves (any lane can decode any subsequence of the staged window),
  // so the work follows the number of runs, not iterations x 256.
  const uint32_t lane = t & 63, wave = t >> 6;
  const hj_slot_words slot_tables = hj_slot_table_words(s_im);
  // (only a few iterations: the long, thin tail of the propagation is left to later
  // launches, which cost one short run each instead of keeping this group resident)
  for (int it = 0; it < max_iters; it++) {
    const bool need = lds_dirty[t] != 0;
    const unsigned long long m = __ballot(need);
    if (lane == 0) lds_wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = 0, total = 0;
    for (uint32_t w = 0; w < HJ_BLOCK/64; w++) {
      const uint32_t c = lds_wcnt[w];
      off += w < wave ? c : 0;
      total += c;
    }
    if (total == 0) break;
    if (need) {
      lds_act[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)t;
      lds_dirty[t] = 0;
    }
    __syncthreads();
    // The first run of all starts every lane from a guess: only its end state means anything,
    // so it is a LITE run (no block count, started part-way into the subsequence)
    // and every lane is marked for a counted run from whatever state it is handed next.
    const bool lite = lite_first && round == 0 && it == 0;
    if (t < total) {
      const uint32_t sub = lds_act[t];
      const uint64_t start = lds_S[sub];
      const uint32_t sb = lds_stop[sub];
      hj_run r;
      if (lite) {
        // ... and since falling into step takes tens of bytes, not 128, it need not start at the
        // subsequence's first bit either: it starts `lite_first - 1` bytes in (never past the
        // middle of what the subsequence holds)
        const uint64_t stop_bit = (uint64_t)(sb & 0x7fffffffu)*8;
        uint64_t from = hj_pos(start);
        uint64_t skip = (uint64_t)(lite_first - 1)*8;
        if (from + 2*skip > stop_bit) skip = from < stop_bit ? (stop_bit - from)/2 : 0;
        r = hj_sync_decode<RowSrc, true, Tab>(hj_source<RowSrc>(lds_win, lds_start, start0, sub, hj_sub_dwords(A)), s_im, &lds_tabs,
         hj_pack(from + skip, hj_slot(start), hj_k(start)), stop_bit, (sb >> 31) == 0u, slot_tables);
        lds_ran[sub] = 2;                                    // ran, but nothing to publish
        lds_dirty[sub] = 1;                                  // (its own flag: no other lane writes it now)
      }
      else {
        r = hj_sync_decode<RowSrc, false, Tab>(hj_source<RowSrc>(lds_win, lds_start, start0, sub, hj_sub_dwords(A)), s_im, &lds_tabs, start,
         (uint64_t)(sb & 0x7fffffffu)*8, (sb >> 31) == 0u, slot_tables);
        lds_R[sub] = (uint16_t)r.nblocks;
        lds_ran[sub] = 1;
      }
      if (sb & 0x80000000u) {
        if (sub + 1 < HJ_BLOCK) {
          if (lds_S[sub + 1] != r.end_state) { lds_S[sub + 1] = r.end_state; lds_dirty[sub + 1] = 1; }
        }
        else A.S[lds_sidx_last + 1] = r.end_state;         // first subsequence of the next group
      }
    }
    __syncthreads();
  }
  // Publish.  Entry t > 0 of S is written only by this group (it was handed over in
  // LDS); entry 0 belongs to the previous group's last lane and is left alone.
  if (on) {
    const uint64_t st = lds_S[t];
    if (t > 0 && st != A.S[sidx]) A.S[sidx] = st;
    if (lds_ran[t] == 1) {                        // (2: only a lite run — still "never ran")
      A.R[L.g] = lds_R[t];
      // clean: its latest run started from st.  dirty: the state moved after that run
      // began, so whatever last_in held before must not make it look settled.
      A.last_in[L.g] = lds_dirty[t] ? ~0ull : st;
    }
  }
  if (__syncthreads_or(on && lds_ran[t]) && t == 0) atomicOr(&A.ran[round], 1u);
}

typedef uint32_t hj_v4u __attribute__((ext_vector_type(4)));
// ---- list rounds (round 5) ---------------------------------------------------------------------------
// After the first round's in-group iterations 6-7 % of a photograph's subsequences still move, a quarter of those
// a step later, and so on down a chain of five or six more steps (4:2:0: the MCU slot has to fall into step as
// well).  Rounds 2-4 walked that tail with one wave per 256 subsequences (hj_sync_sparse): sixteen of its lanes busy in
// the first step, four in the second, and every wave of the batch resident for three steps' time — 19 000 vector
// instructions per wave for what three or four lanes do.  A LIST round runs exactly the subsequences that moved:
//   hj_list_build   one lane per subsequence: those whose start state differs from the one their latest run
//                   began in go onto their image's work list (wave-aggregated append);
//   hj_sync_list    a workgroup takes 256 list entries of one image (tables in LDS, the entries' rows staged
//                   in LDS like the dense kernel's — any 140 bytes of the scan), runs them, and appends every
//                   successor whose start state it changed to the NEXT round's list.  One launch = one step
//                   of the chain, at the speed of a lane with a SIMD almost to itself (~30 us), on a few
//                   hundred waves instead of 4 608.
// An image whose whole list fits one workgroup ("solo") is iterated inside it: successors and their states stay in
// LDS, step after step without a launch in between, up to max_iters; what still moves then goes to the next
// round's list.  Nobody else touches a solo image in that launch, and in a shared image every subsequence is on
// the list at most once (its one predecessor is the only lane that can put it there), so no two lanes ever run
// the same subsequence in one launch.  A lane that reads a start state its predecessor is just replacing runs
// from the old or the new one — and is on the next list in both cases.
// Counters: list_count[(r & 3)][image] is round r's list length; round r zeroes the counter of round r + 2.
// (every image's counter has a 256-byte line to itself — HJ_LIST_CSTRIDE words: device-wide atomics on one line are
// served one after the other by one memory channel, ~8 ns each: 18 000 wave-level appends to 48 neighbouring counters
// took hj_list_build 157 us; one append per workgroup onto a line per image takes it ~10)
#define HJ_LIST_BLOCK 256
static __device__ __forceinline__ uint32_t *hj_list_counter(const hj_args &A, int round, uint32_t img) {
  return A.list_count + ((uint32_t)(round & 3)*(uint32_t)A.nimages + img)*HJ_LIST_CSTRIDE;
}
__global__ __launch_bounds__(256) void hj_list_build(const hj_args A, int round) {
  __shared__ uint32_t wcnt[4], wbase;
  const hj_image *im = A.images + blockIdx.y;
  const uint32_t li = blockIdx.x*256u + threadIdx.x, nsub = im->nsub, sub0 = im->sub0;
  bool dirty = false;
  if (li < nsub) {
    const uint32_t g = sub0 + li;
    dirty = A.S[g + im->seg0 + A.sub_seg[g]] != A.last_in[g];
  }
  const unsigned long long m = __ballot(dirty);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    wbase = total ? atomicAdd(hj_list_counter(A, round, blockIdx.y), total) : 0u;
  }
  __syncthreads();
  if (!dirty) return;
  uint32_t pos = wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (uint32_t w = 0; w < wave; w++) pos += wcnt[w];
  A.list[(size_t)(round & 1)*A.list_stride + sub0 + pos] = li;
}

// An entry's row in LDS: the bytes from (first byte of its subsequence & ~15) on — its lane loads them itself, sixteen
// at a time and all of them in flight at once (a loop of dependent dword loads over scattered rows cost a step 15-20 us)
// — as big-endian dwords at an odd stride: 4 x loads + 3 (43, 27, 19 for subsequences of 128, 64, 32 bytes).
#define HJ_LIST_MAX_LOADS 10
#define HJ_LIST_ROW_MAX (4*HJ_LIST_MAX_LOADS + 3)
static __device__ __forceinline__ uint32_t hj_list_loads(const hj_args &A) { return ((1u << A.sub_log2) + 15u + 12u + 15u) >> 4; }

// ROWS: the entries' rows in LDS (hj_lds_reg_src: a step of the chain at its shortest) or read from global memory as
// the run goes (hj_gmem_src: 12 KB of LDS per workgroup instead of 56, for the FIRST list round of a batch that fills
// the device — a fifth of all subsequences, bound by how many waves a CU holds, not by one lane's speed).
template <class Tab, bool ROWS>
__global__ __launch_bounds__(HJ_LIST_BLOCK) void hj_sync_list(const hj_args A, int round, int max_iters) {
  constexpr int NB = HJ_LIST_BLOCK;
  __shared__ __attribute__((aligned(16))) Tab lds_tabs;
  __shared__ uint32_t lds_win_mem[ROWS ? 1 + NB*HJ_LIST_ROW_MAX + 8 : 1];      // [0]: the dword "before" row 0
  uint32_t *lds_win = lds_win_mem + 1;
  __shared__ hj_image s_im;
  const uint32_t img = blockIdx.y, t = threadIdx.x, lane = t & 63u;
  if (blockIdx.x == 0 && t == 0) *hj_list_counter(A, round + 2, img) = 0;
  const uint32_t n_in = *hj_list_counter(A, round, img);
  if (blockIdx.x*NB >= n_in) return;
  uint32_t *cnt_out = hj_list_counter(A, round + 1, img);
  const hj_image im = A.images[img];             // scalar fields only; indexed ones via s_im
  const uint32_t *list_in = A.list + (size_t)(round & 1)*A.list_stride + im.sub0;
  uint32_t *list_out = A.list + (size_t)((round + 1) & 1)*A.list_stride + im.sub0;
  // solo: the image's whole list is this workgroup's — its lanes walk their chains on, step after step
  const int steps = n_in <= (uint32_t)NB ? max_iters : 1;
  const uint8_t *scan = A.scan + im.scan_off;
  const uint32_t padded = (im.scan_len + 16 + 15) & ~15u;
  const uint32_t sub = 1u << A.sub_log2;
  const uint32_t nload = hj_list_loads(A), sdw = 4u*nload + 3u;
  typedef hj_v4u __attribute__((may_alias)) v4u_alias;
  hj_v4u v[ROWS ? HJ_LIST_MAX_LOADS : 1];
  auto load_row = [&](uint32_t first) {          // sixteen-byte loads: every image's scan starts on a multiple of 16 and is followed by its pad
#pragma unroll
    for (uint32_t j = 0; j < HJ_LIST_MAX_LOADS; j++) {
      if (ROWS && j < nload) {
        uint32_t a = (first & ~15u) + 16u*j;
        if (a + 16u > padded) a = padded - 16u;
        v[j] = *reinterpret_cast<const v4u_alias *>(scan + a);
      }
    }
  };
  // what the first chunk's lanes run: looked up while the tables are on their way into LDS
  uint32_t chunk = blockIdx.x;
  uint32_t li = 0, g = 0, sidx = 0, first = 0, seg_end = 0, left = 0;   // left: subsequences of the segment behind this one
  uint64_t start = 0;
  auto describe = [&](uint32_t c) -> bool {
    if (c*NB + t >= n_in) return false;
    li = list_in[c*NB + t];
    g = im.sub0 + li;
    const uint32_t si = A.sub_seg[g];
    const hj_segment sg = A.segs[im.seg0 + si];
    const uint32_t i = li - sg.sub0;
    first = sg.start + (i << A.sub_log2);
    seg_end = sg.end;
    left = sg.nsub - 1u - i;
    sidx = g + im.seg0 + si;
    start = A.S[sidx];
    load_row(first);
    return true;
  };
  bool active = describe(chunk);
  hj_stage_image(&s_im, A.images + img);
  hj_stage_tables<NB>(&lds_tabs, A.tables + img, A.wide ? A.wide + (A.wide_shared ? 0u : img) : nullptr);
  __syncthreads();
  const hj_slot_words slot_tables = hj_slot_table_words(s_im);
  bool left_work = false;
  for (;;) {
    for (int it = 0; ; it++) {
      bool moved = false;
      if (active) {
        const bool last = left == 0u;
        uint32_t stop = first + sub;
        if (stop > seg_end) stop = seg_end;
        const uint64_t next_in = last ? 0ull : A.S[sidx + 1];       // (this lane is the only one that ever writes it)
        hj_run r;
        if (ROWS) {
          uint32_t *row = lds_win + t*sdw;
#pragma unroll
          for (uint32_t j = 0; j < HJ_LIST_MAX_LOADS; j++) {
            if (j < nload) {
              row[4*j] = __builtin_bswap32(v[j].x); row[4*j + 1] = __builtin_bswap32(v[j].y);
              row[4*j + 2] = __builtin_bswap32(v[j].z); row[4*j + 3] = __builtin_bswap32(v[j].w);
            }
          }
          if (!last && it + 1 < steps) load_row(first + sub);       // the successor's row, on its way while this one is decoded
          hj_lds_reg_src src;
          src.base = row;
          src.bit0 = (first & ~15u) << 3;
          r = hj_sync_decode<hj_lds_reg_src, false, Tab>(src, s_im, &lds_tabs, start, (uint64_t)stop*8, last, slot_tables);
        }
        else {
          hj_gmem_src src;
          src.scan32 = reinterpret_cast<const uint32_t *>(scan);
          src.dw0 = first >> 2;
          src.ndw = padded >> 2;
          r = hj_sync_decode<hj_gmem_src, false, Tab>(src, s_im, &lds_tabs, start, (uint64_t)stop*8, last, slot_tables);
        }
        A.R[g] = r.nblocks;
        A.last_in[g] = start;
        if (!last && next_in != r.end_state) { A.S[sidx + 1] = r.end_state; moved = true; }
        // on to the successor: the same segment, the next 2^sub_log2 bytes
        start = r.end_state;
        li++; g++; sidx++; first += sub; left--;
      }
      active = moved;
      if (it + 1 >= steps) break;
      if (!__syncthreads_or(active)) break;                  // (also orders a step's stores before the next step's loads)
    }
    // chains that still move go onto the next round's list
    const unsigned long long m = __ballot(active);
    if (m != 0ull) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(cnt_out, (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (active) list_out[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = li;
      left_work = true;
    }
    chunk += gridDim.x;
    if (chunk*NB >= n_in) break;
    active = describe(chunk);                                // (solo: one chunk; otherwise one step per chunk, rows are private)
  }
  if (left_work && lane == 0) atomicOr(&A.ran[round], 1u);
}

#define HJ_SCAN_BLOCK 1024          /* threads per chunk */
#define HJ_SCAN_ITEMS 4             /* consecutive lanes summed by one thread */
#define HJ_SCAN_CHUNK_LOG2 12       /* HJ_SCAN_BLOCK*HJ_SCAN_ITEMS lanes per workgroup */
static_assert(HJ_SCAN_BLOCK*HJ_SCAN_ITEMS == 1 << HJ_SCAN_CHUNK_LOG2, "chunk size");
// Exclusive prefix sums of the blocks completed over the lanes of every segment, in two launches
// so that a long segment is not one workgroup's serial loop (one 4K 4:4:4 image without restart
// markers: 196 us -> 2 x ~10): a workgroup takes a chunk of 4096 lanes (each thread sums four
// consecutive lanes, a shuffle scan runs over the wavefront, the sixteen wave totals go through
// LDS);
//   FINAL = false  stores the chunk's total,
//   FINAL = true   adds the totals of the chunks before it in the segment and writes B.
// Chunk c of segment gs keeps its total at scan_part[gs + (first lane of the segment >> 12)
// + c]: distinct and increasing over the batch, below total_seg + total_sub/4096 + 1.
// FINAL also prepares the planes for the write pass, which stores a block that two lanes share as
// 2-byte pieces onto a ZERO background and every other block as a whole 128-byte line: each lane
// whose start state lies inside a block zeroes that block's line here (now that B says which
// block it is) — 128 bytes per subsequence instead of a memset of every plane of the batch
// (1.2 GB per 48 x 4K, on the side stream: a quarter of the HBM traffic of a decode of lighter
// content, whose pipeline ran 285-300 Gpixel/s with it and 337-346 without).
template <bool FINAL>
__global__ __launch_bounds__(HJ_SCAN_BLOCK) void hj_scan(const hj_args A) {
  __shared__ uint32_t wtot[HJ_SCAN_BLOCK/64];
  __shared__ uint32_t zline[FINAL ? HJ_SCAN_BLOCK*HJ_SCAN_ITEMS : 1];   // 128-byte lines to zero (~0: none)
  __shared__ hj_image s_im;
  // blockIdx.x = batch-global segment; find its image (few images: linear search)
  const uint32_t gs = blockIdx.x, c = blockIdx.y;
  int img = 0;
  while (img + 1 < A.nimages && A.images[img + 1].seg0 <= gs) img++;
  const hj_image im = A.images[img];
  const hj_segment sg = A.segs[gs];
  const uint32_t c0 = c << HJ_SCAN_CHUNK_LOG2;
  if (c0 >= sg.nsub) return;
  const uint32_t si = gs - im.seg0;
  const uint32_t total = sg.nmcu*(uint32_t)im.nslots;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t first = im.sub0 + sg.sub0;
  uint32_t *part = A.scan_part + (size_t)(gs + (first >> HJ_SCAN_CHUNK_LOG2) + c);
  const uint32_t i0 = c0 + threadIdx.x*HJ_SCAN_ITEMS;
  const uint32_t g0 = first + i0;
  uint32_t v[HJ_SCAN_ITEMS], slot[HJ_SCAN_ITEMS];
  bool inside[HJ_SCAN_ITEMS];                                 // the lane starts inside a block
  uint32_t mine = 0;
  if (FINAL) hj_stage_image(&s_im, A.images + img);           // (read after the barrier below)
#pragma unroll
  for (int j = 0; j < HJ_SCAN_ITEMS; j++) {
    v[j] = 0;
    slot[j] = 0;
    inside[j] = false;
    if (i0 + j < sg.nsub) {
      v[j] = A.R[g0 + j];
      if (FINAL) {
        const uint64_t st = A.S[g0 + j + im.seg0 + si];
        slot[j] = (uint32_t)hj_slot(st);
        inside[j] = hj_k(st) != 0;
      }
    }
    mine += v[j];
  }
  uint32_t run = 0;
  if (FINAL) {                                               // chunks before this one
    for (uint32_t k = 1; k <= c; k++) run += part[-(int)k];
  }
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t x = (uint32_t)__shfl_up((int)inc, d);
    if ((int)lane >= d) inc += x;
  }
  if (lane == 63u) wtot[wave] = inc;
  __syncthreads();
  uint32_t tot = 0;
  run += inc - mine;
  for (uint32_t w = 0; w < HJ_SCAN_BLOCK/64; w++) {
    const uint32_t x = wtot[w];
    tot += x;
    if (w < wave) run += x;
  }
  if (!FINAL) {
    if (threadIdx.x == 0) part[0] = tot;
    return;
  }
  bool bad = false;
  // data ran out before the last MCU (seen by the segment's last chunk)
  if (c0 + (1u << HJ_SCAN_CHUNK_LOG2) >= sg.nsub && threadIdx.x == 0 && run + tot < total) bad = true;
#pragma unroll
  for (int j = 0; j < HJ_SCAN_ITEMS; j++) {
    uint32_t z = ~0u;
    if (i0 + j < sg.nsub) {
      A.B[g0 + j] = run;
      // the slot a lane starts in must agree with the number of blocks before it
      if (run < total && slot[j] != run % (uint32_t)im.nslots) bad = true;
      else if (run < total && inside[j]) {
        z = (uint32_t)(hj_block_offset(s_im, sg.mcu0 + run/(uint32_t)im.nslots, (int)slot[j]) >> 6);
      }
    }
    zline[threadIdx.x*HJ_SCAN_ITEMS + j] = z;
    run += v[j];
  }
  if (bad) atomicOr(&A.errors[img], 1u);
  __syncthreads();
  // eight threads per line, 16 bytes each: whole-line stores
  typedef __attribute__((address_space(1))) hj_v4u global_v4u;
  int16_t *planes = A.coef + (long long)img*A.coef_stride;
  const hj_v4u zero = {0u, 0u, 0u, 0u};
  for (uint32_t e = threadIdx.x >> 3; e < HJ_SCAN_BLOCK*HJ_SCAN_ITEMS; e += HJ_SCAN_BLOCK/8) {
    const uint32_t l = zline[e];
    if (l != ~0u) *((global_v4u *)((uintptr_t)planes + (size_t)l*128u) + (threadIdx.x & 7u)) = zero;
  }
}

// Output side of the write pass.  Every coefficient lands in the lane's LDS block buffer first.
// At a write-out point the WAVE writes the blocks of its waiting lanes together: the k-th
// waiting lane's block is handled by lanes 8k..8k+7 of a pass, 16 bytes each, so every store
// instruction writes whole 128-byte lines (16-byte pieces at a 128-byte stride per lane cost
// 2-4x more in the memory pipeline).
typedef int16_t __attribute__((may_alias)) hj_i16_alias;    // 16-bit view of the dword buffer
struct hj_block_out {
  const hj_image *im;
  int16_t *coef;                     // this image's planes
  uint32_t *blk;                     // this lane's LDS buffer: 32 dwords + 1 (its block's byte offset)
  uint32_t *wave_blk;                // buffer of lane 0 of this wave
  uint8_t *rank_lane;                // per wave: lane number of the k-th waiting lane
  uint32_t mbx, mby;                 // MCU of the block being decoded
  int flush_lanes;
  __device__ __forceinline__ void init(uint32_t mcu) {
    mby = mcu/(uint32_t)im->nhmb;
    mbx = mcu - mby*(uint32_t)im->nhmb;
  }
  __device__ __forceinline__ bool any(bool x) const { return __ballot(x) != 0ull; }
  // Write out when `flush_lanes` lanes hold a finished block, or when nobody can decode on.
  __device__ __forceinline__ bool flush_due(bool waiting, bool running) const {
    const unsigned long long w = __ballot(waiting);
    return w != 0ull && ((int)__popcll(w) >= flush_lanes || __ballot(running) == 0ull);
  }
  // offset (shorts) of the current MCU's block `slot` in the image's planes
  // (inverse of the MCU loop nest + block placement of src/xjpeg.c:461-472, 556-561), from the
  // slot's descriptor: one 8-byte LDS read instead of seven byte reads per block
  const uint2 *slotd;                // [nslots] {plane base + sbx*64, hs | vs << 8 | sby << 16 | xdec << 24}
  uint32_t rs;                       // shorts per row of luma blocks
  static __device__ __forceinline__ uint2 describe(const hj_image &m, int slot) {
    const int comp = m.slot_comp[slot];
    return make_uint2((uint32_t)m.comp_coef_off[comp] + ((uint32_t)m.slot_sbx[slot] << 6),
     (uint32_t)m.comp_hs[comp] | (uint32_t)m.comp_vs[comp] << 8 | (uint32_t)m.slot_sby[slot] << 16
     | (uint32_t)m.comp_xdec[comp] << 24);
  }
  __device__ __forceinline__ uint32_t offset(int slot) const {
    const uint2 d = slotd[slot];
    const uint32_t hs = d.y & 255u, vs = (d.y >> 8) & 255u, sby = (d.y >> 16) & 255u, xd = d.y >> 24;
    const uint32_t by = mby*vs + sby;
    return d.x + ((mbx*hs) << 6) + rs*(by >> xd) + (rs >> xd)*(by & ((1u << xd) - 1u));
  }
  __device__ __forceinline__ void next_block(int slot) {
    if (slot + 1 == im->nslots) {
      mbx++;
      if (mbx == (uint32_t)im->nhmb) { mbx = 0; mby++; }
    }
  }
  // Every lane of the wave calls this together: the blocks of the lanes with `have` leave their
  // buffers.  A block that is `partial` (shared with a neighbouring lane: begun before this run,
  // or unfinished at its end) leaves as 2-byte stores of its non-zeros onto its line, which hj_scan<true> zeroed
  // (disjoint positions: the lanes need no ordering between them), a whole one as a 128-byte
  // line; either way 8 lanes handle one block, 16 bytes each, and the buffers are zero afterwards.
  __device__ __forceinline__ void flush_blocks(bool have, bool partial, bool complete, int slot) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = __ballot(have);
    if (mask == 0ull) return;
    // whole blocks first, shared ones after them: each kind gets its own passes, so the 2-byte
    // stores of the shared ones are only issued by passes that hold nothing else (one or two per
    // write-out instead of most of them)
    const unsigned long long pmask = __ballot(have && partial), fmask = mask & ~pmask;
    const uint32_t nfull = (uint32_t)__popcll(fmask), cnt = (uint32_t)__popcll(mask);
    if (have) {
      const unsigned long long m = partial ? pmask : fmask;
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      rank_lane[(partial ? nfull : 0u) + r] = (uint8_t)lane;
      blk[32] = offset(slot)*2u;                             // byte offset of the block (a multiple of 128)
    }
    const uint32_t part = lane & 7u;
    typedef __attribute__((address_space(1))) hj_v4u global_v4u;
    for (uint32_t k = lane >> 3; k < nfull; k += 8) {
      uint32_t *src = wave_blk + (uint32_t)rank_lane[k]*HJ_BLK_STRIDE;
      const uint32_t off = src[32];
      hj_v4u v;
      v.x = src[4*part]; v.y = src[4*part + 1]; v.z = src[4*part + 2]; v.w = src[4*part + 3];
      src[4*part] = 0; src[4*part + 1] = 0; src[4*part + 2] = 0; src[4*part + 3] = 0;
      __builtin_nontemporal_store(v, (global_v4u *)((uintptr_t)coef + off) + part);
    }
    for (uint32_t k = nfull + (lane >> 3); k < cnt; k += 8) {
      uint32_t *src = wave_blk + (uint32_t)rank_lane[k]*HJ_BLK_STRIDE;
      const uint32_t off = src[32];
      const uint32_t d4[4] = {src[4*part], src[4*part + 1], src[4*part + 2], src[4*part + 3]};
      src[4*part] = 0; src[4*part + 1] = 0; src[4*part + 2] = 0; src[4*part + 3] = 0;
      typedef __attribute__((address_space(1))) int16_t global_i16;
      global_i16 *dst = (global_i16 *)((uintptr_t)coef + off) + 8*part;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (d4[q] & 0xffffu) dst[2*q] = (int16_t)(d4[q] & 0xffffu);
        if (d4[q] >> 16) dst[2*q + 1] = (int16_t)(d4[q] >> 16);
      }
    }
    if (have && complete) next_block(slot);                 // on to the next block's place
  }
};

// ---- the write pass -------------------------------------------------------------------------------
// One lane per subsequence, final decode from its true start state.  The scan is read from global
// memory (hj_gmem_src), a workgroup is 512 lanes: 81 KB of LDS (block buffers + one copy of the
// tables), 2 x 8 waves per CU.  The per-symbol path is built around what the counters said about
// the first edition (54 scalar + 16 branch instructions per 100 vector ones: exec-mask bookkeeping
// of a loop in which every "rare" case is taken by some lane of the wave on nearly every trip):
//   * the tables are re-encoded while they are staged (hj_wtables): DC and AC entries are both
//     32 bits wide, so a symbol's lookup is ONE ds_read_b32 from a table base selected by k == 0;
//     an EOB advances 127, so that "k + adv has bit 7 set" means EOB and "64 < (k + adv) & 127"
//     means an AC run past coefficient 63; bit 16 marks bit patterns that are no code.  The
//     error tests become an OR and a MAX per symbol, looked at once after the run;
//   * the magnitude is cut out with v_bfe_u32 / v_bfe_i32 (width 0 gives 0: no "s == 0" case);
//   * no DC prediction: a block's DC coefficient is stored as the DIFFERENCE it was coded as, and
//     the lane that decoded it also leaves it in A.dc_diff[block in scan order] when the block
//     leaves its buffer; hj_dc_scan turns differences into values afterwards;
//   * every symbol stores its value: the de-zigzag table is extended so that the positions an
//     EOB / a bad run computes (>= 64) land in the spare half-dword behind the block buffer, and a
//     ZRL stores a zero where a zero is — no "is there a value" branch; the store happens one
//     symbol LATE, so that a symbol costs the wave one LDS round trip, not two in a row;
//   * pieces of blocks shared with a neighbouring lane ride the same collective write-out as
//     whole blocks (8 lanes per block, 16 bytes each) with 2-byte stores of their non-zeros.
// What is left under a branch: the level-2 lookup of codes longer than 9 bits and the scan
// dword refill.  815 -> 670 us per 48 x 4K (round 3).
struct hj_wtables {
  uint32_t dc[2][1 << HJ_FAST_BITS];
  uint32_t ac[2][1 << HJ_FAST_BITS];
  uint16_t l2[HJ_L2_BLOCKS*128];
};
#define HJ_W_NOCODE 0x10000u
static __device__ __forceinline__ uint32_t hj_wentry(uint32_t e) {      // e: a 16-bit hj_tables entry
  if (HJ_IS_ESCAPE(e)) return e;
  if (HJ_E_ADV(e) == 64) e |= 127u << 5;                                 // EOB: 64 -> 127
  if (HJ_E_LEN(e) > 16) e |= HJ_W_NOCODE;
  return e;
}
#ifndef HJ_WRITE_UNROLL
#define HJ_WRITE_UNROLL 4
#endif
// (Round 6 A/B, commit f9cf3e7: a second AC symbol per look-up where nine bits hold both — entry pairs made while the
// tables are staged, two owed stores per trip — 661-666 us against 667-673 on the bench's files, 408-413 against
// 404-406 on photograph-like content: every trip pays for the second symbol whether it exists or not (1.42 symbols per
// look-up).  Not kept: profiles/r6_entropy_ab.md.)
#define HJ_DEZZ_EXT 192              /* k + adv - 1 <= 63 + 127 */
#define HJ_WRITE_BLOCK 512

__global__ __launch_bounds__(HJ_WRITE_BLOCK) void hj_write(const hj_args A) {
  constexpr int NB = HJ_WRITE_BLOCK;
  __shared__ __attribute__((aligned(16))) hj_wtables lds_tabs;
  __shared__ uint32_t lds_blk[NB*HJ_BLK_STRIDE];
  __shared__ hj_image s_im;
  __shared__ uint32_t s_dezz[HJ_DEZZ_EXT];           // BYTE offset of zig-zag position i in a block buffer (dwords: the
                                                     // loaded value is used as it comes, a symbol later)
  __shared__ uint8_t s_rank[NB];
  __shared__ uint2 s_slotd[HJ_MAX_SLOTS];
  const hj_image im0 = A.images[blockIdx.y];
  if (blockIdx.x*NB >= im0.nsub) return;              // grid.x covers the largest image
  hj_stage_image(&s_im, A.images + blockIdx.y);
  if (threadIdx.x < (uint32_t)im0.nslots) s_slotd[threadIdx.x] = hj_block_out::describe(A.images[blockIdx.y], (int)threadIdx.x);
  if (threadIdx.x < HJ_DEZZ_EXT) s_dezz[threadIdx.x] = threadIdx.x < 64 ? 2u*HJ_DEZZ[threadIdx.x] : 128u;
  {
    const hj_tables *T = A.tables + blockIdx.y;
    for (int i = threadIdx.x; i < 2 << HJ_FAST_BITS; i += NB) {
      (&lds_tabs.dc[0][0])[i] = hj_wentry((&T->dc[0][0])[i]);
      (&lds_tabs.ac[0][0])[i] = hj_wentry((&T->ac[0][0])[i] & 0xffffu);
    }
    for (int i = threadIdx.x; i < HJ_L2_BLOCKS*128; i += NB) {
      const uint32_t e = T->l2[i];
      lds_tabs.l2[i] = (uint16_t)(HJ_E_ADV(e) == 64 ? e | (127u << 5) : e);     // (no-code is tested when read)
    }
  }
  // lane context (as hj_prologue, without its staging)
  const uint32_t li = blockIdx.x*NB + threadIdx.x;
  const bool on = li < im0.nsub;
  uint32_t g = 0, si = 0, i_in_seg = 0, seg_end = 0, seg_nsub = 0, seg_mcu0 = 0, seg_nmcu = 0, my_start = 0;
  if (on) {
    g = im0.sub0 + li;
    si = A.sub_seg[g];
    const hj_segment sg = A.segs[im0.seg0 + si];
    i_in_seg = li - sg.sub0;
    seg_end = sg.end; seg_nsub = sg.nsub; seg_mcu0 = sg.mcu0; seg_nmcu = sg.nmcu;
    my_start = sg.start + (i_in_seg << A.sub_log2);
  }
  uint32_t *blk = lds_blk + threadIdx.x*HJ_BLK_STRIDE;
#pragma unroll
  for (int q = 0; q < 32; q++) blk[q] = 0;                  // own buffer only
  __syncthreads();                                           // tables, s_im, s_dezz
  const hj_image &im = s_im;
  const uint32_t total = seg_nmcu*(uint32_t)im.nslots;
  const uint32_t b0 = on ? A.B[g] : 0u;
  const bool live = on && b0 < total;
  const uint32_t sidx = g + im.seg0 + si;
  const uint64_t start = live ? A.S[sidx] : 0ull;
  const uint64_t stop = !live ? 0ull : i_in_seg + 1 < seg_nsub ? hj_pos(A.S[sidx + 1]) : (uint64_t)seg_end*8;
  hj_block_out out;
  out.im = &im;
  out.coef = A.coef + (long long)blockIdx.y*A.coef_stride;
  out.blk = blk;
  out.wave_blk = lds_blk + (threadIdx.x & ~63u)*HJ_BLK_STRIDE;
  out.rank_lane = s_rank + (threadIdx.x & ~63u);
  out.slotd = s_slotd;
  out.rs = (uint32_t)im0.w0_blocks*64u;
  out.init(seg_mcu0 + b0/(uint32_t)im.nslots);
  out.flush_lanes = A.flush_lanes;
  const uint32_t max_blocks = live ? total - b0 : 0u;
  // this lane's blocks in scan order: DC differences go to dcd[n]
  int16_t *dcd = A.dc_diff + (long long)blockIdx.y*A.dc_stride + (size_t)seg_mcu0*(uint32_t)im.nslots + b0;

  const uint32_t slot_tbl_bits = hj_slot_tables(im);
  const int nslots = im.nslots;
  hj_gmem_src gsrc;
  gsrc.scan32 = reinterpret_cast<const uint32_t *>(A.scan + im0.scan_off);
  gsrc.dw0 = my_start >> 2;
  gsrc.ndw = ((im0.scan_len + 16 + 15) & ~15u) >> 2;
  hj_gmem_src::reader br;
  br.init(gsrc, hj_pos(start), stop);
  int k = hj_k(start), c = hj_slot(start);
  bool head = k == 0, waiting = false;
  uint32_t n = 0, errbits = 0;
  int errm = 0, dcv = 0;
  int tbl = (int)((slot_tbl_bits >> (2*c)) & 3u);
  const uint32_t *tb_dc = lds_tabs.dc[tbl & 1], *tb_ac = lds_tabs.ac[tbl >> 1];
  // The DC differences this lane decodes belong to consecutive blocks (all its blocks but, when
  // it starts inside one, the first): they leave four at a time, as one 8-byte store — a 2-byte
  // store per block into lines that eight lanes share cost the pass 75 us per 48 x 4K.
  struct {
    int16_t *at;                     // where the next store goes
    uint64_t acc;
    uint32_t cnt;
    __device__ __forceinline__ void push(int v) {
      acc |= (uint64_t)(uint16_t)v << (16u*(cnt & 3u));
      cnt++;
      if ((cnt & 3u) == 0u) {
        typedef uint64_t __attribute__((aligned(2), may_alias)) u64_a2;
        *reinterpret_cast<u64_a2 *>(at) = acc;
        at += 4;
        acc = 0;
      }
    }
    __device__ __forceinline__ void finish() {
      typedef uint32_t __attribute__((aligned(2), may_alias)) u32_a2;
      if (cnt & 2u) { *reinterpret_cast<u32_a2 *>(at) = (uint32_t)acc; at += 2; acc >>= 32; }
      if (cnt & 1u) *at = (int16_t)acc;
    }
  } dcq;
  dcq.at = dcd + (head ? 0 : 1);
  dcq.acc = 0;
  dcq.cnt = 0;
  uint8_t *blk8 = reinterpret_cast<uint8_t *>(blk);
  uint32_t pz = 128;
  int pv = 0;
  for (;;) {
    bool running = !waiting && br.before_stop() && n < max_blocks;
    if (!out.any(running || waiting)) break;
#pragma unroll
    for (int u = 0; u < HJ_WRITE_UNROLL; u++) {
      if (u) running = !waiting && br.before_stop() && n < max_blocks;
      if (!running) continue;
      const uint32_t w = br.window();
      const bool isdc = k == 0;
      const uint32_t *tb = isdc ? tb_dc : tb_ac;
      uint32_t e = tb[w >> (32 - HJ_FAST_BITS)];
      if ((e & 31u) == 0u) {                                 // a code longer than 9 bits
        e = lds_tabs.l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
        if (HJ_E_LEN(e) > 16) e |= HJ_W_NOCODE;
      }
      const uint32_t tot = e & 31u, s = (e >> 12) & 15u;
      const uint32_t off = 32u - tot;
      const int vu = (int)__builtin_amdgcn_ubfe(w, off, s);
      const int vs = __builtin_amdgcn_sbfe((int)w, off, s);  // < 0 iff the top magnitude bit is set (value >= 0)
      const int v = vu - (vs < 0 ? 0 : (int)((1u << s) - 1u));   // T.81 F.2.2.1 EXTEND; s = 0 gives 0
      dcv = isdc ? v : dcv;                                  // the block's DC difference, kept for dcd[]
      const int kn = k + (int)((e >> 5) & 127u);             // one past this coefficient's zig-zag index
      *reinterpret_cast<hj_i16_alias *>(blk8 + pz) = (int16_t)pv;   // (the previous symbol's value)
      pz = s_dezz[kn - 1];                                   // (>= 64: the spare half-dword)
      pv = v;
      errbits |= e;
      errm = max(errm, kn & 127);
      br.skip((int)tot);
      waiting = kn >= 64;
      k = waiting ? 0 : kn;
    }
    *reinterpret_cast<hj_i16_alias *>(blk8 + pz) = (int16_t)pv;     // the store still owed
    pz = 128;
    if (out.flush_due(waiting, running && !waiting)) {
      if (waiting && head) dcq.push(dcv);                    // (the lane that decoded the DC symbol reports it)
      out.flush_blocks(waiting, !head, true, c);
      if (waiting) {
        n++;
        c = c + 1 == nslots ? 0 : c + 1;
        tbl = (int)((slot_tbl_bits >> (2*c)) & 3u);
        tb_dc = lds_tabs.dc[tbl & 1]; tb_ac = lds_tabs.ac[tbl >> 1];
        head = true;
        waiting = false;
      }
    }
  }
  // blocks left unfinished (a later lane holds the rest): their coefficients so far
  const bool rest = k != 0 && n < max_blocks;
  if (rest && head) dcq.push(dcv);
  dcq.finish();
  out.flush_blocks(rest, true, false, c);
  if ((errbits & HJ_W_NOCODE) || errm > 64) atomicOr(&A.errors[blockIdx.y], 2u);
}

// ---- the write pass of a SMALL batch: one lane per BLOCK (round 5) ---------------------------------------
// A lone frame's write pass is what ONE lane takes for the ~206 symbols of its subsequence and its ~8 write-outs:
// 90 us of a 1080p frame's 410 with most of the device idle.  Once the states are settled the places where blocks
// start are as good as known — a synchronisation run passes every one of them — so a small batch takes two launches
// instead of hj_write:
//   hj_block_starts  one lane per subsequence: the counted run once more (12-bit packs when the batch brought them,
//                    its row in LDS), storing the bit position of every block start it passes at
//                    blk_pos[block number in scan order] (+ 1: 0 = "no such block", what hj_init_states leaves there);
//   hj_write_blocks  one lane per block: ~25 symbols from blk_pos, every block whole in its lane's LDS buffer, the
//                    wave's 64 blocks written out together as 128-byte lines, DC differences as consecutive 2-byte
//                    stores.  No block is shared between lanes, so nothing is stored piecewise.
// Same symbols decoded, same error tests (a bit pattern that is no code, a run past coefficient 63) as hj_write.
template <class Tab>
__global__ __launch_bounds__(HJ_LIST_BLOCK) void hj_block_starts(const hj_args A) {
  constexpr int NB = HJ_LIST_BLOCK;
  __shared__ __attribute__((aligned(16))) Tab lds_tabs;
  __shared__ uint32_t lds_win_mem[1 + NB*HJ_LIST_ROW_MAX + 8];
  uint32_t *lds_win = lds_win_mem + 1;
  __shared__ hj_image s_im;
  const uint32_t img = blockIdx.y, t = threadIdx.x;
  const hj_image im = A.images[img];
  if (blockIdx.x*NB >= im.nsub) return;
  const uint32_t li = blockIdx.x*NB + t;
  const bool on = li < im.nsub;
  const uint8_t *scan = A.scan + im.scan_off;
  const uint32_t padded = (im.scan_len + 16 + 15) & ~15u;
  const uint32_t nload = hj_list_loads(A), sdw = 4u*nload + 3u;
  typedef hj_v4u __attribute__((may_alias)) v4u_alias;
  uint32_t g = 0, si = 0, first = 0, seg_block0 = 0, total = 0, b0 = 0;
  uint64_t start = 0, stop = 0;
  bool live = false;
  hj_v4u v[HJ_LIST_MAX_LOADS];                               // the lane's row: all of its loads in flight before the first is used,
  if (on) {                                                  // and the tables staged while they are on their way
    g = im.sub0 + li;
    si = A.sub_seg[g];
    const hj_segment sg = A.segs[im.seg0 + si];
    const uint32_t i = li - sg.sub0;
    first = sg.start + (i << A.sub_log2);
    seg_block0 = sg.mcu0*(uint32_t)im.nslots;
    total = sg.nmcu*(uint32_t)im.nslots;
    b0 = A.B[g];
    live = b0 < total;
    const uint32_t sidx = g + im.seg0 + si;
    start = A.S[sidx];
    stop = i + 1 < sg.nsub ? hj_pos(A.S[sidx + 1]) : (uint64_t)sg.end*8;
#pragma unroll
    for (uint32_t j = 0; j < HJ_LIST_MAX_LOADS; j++) {
      if (j < nload) {
        uint32_t a = (first & ~15u) + 16u*j;
        if (a + 16u > padded) a = padded - 16u;
        v[j] = *reinterpret_cast<const v4u_alias *>(scan + a);
      }
    }
  }
  hj_stage_image(&s_im, A.images + img);
  hj_stage_tables<NB>(&lds_tabs, A.tables + img, A.wide && !A.wide_shared ? A.wide + img : nullptr);
  if (on) {
    uint32_t *row = lds_win + t*sdw;
#pragma unroll
    for (uint32_t j = 0; j < HJ_LIST_MAX_LOADS; j++) {
      if (j < nload) {
        row[4*j] = __builtin_bswap32(v[j].x); row[4*j + 1] = __builtin_bswap32(v[j].y);
        row[4*j + 2] = __builtin_bswap32(v[j].z); row[4*j + 3] = __builtin_bswap32(v[j].w);
      }
    }
  }
  __syncthreads();                                           // tables, s_im
  if (!live) return;
  uint32_t *pos = A.blk_pos + (size_t)img*(size_t)A.dc_stride + seg_block0;
  const hj_slot_words W = hj_slot_table_words(s_im);
  const uint32_t c2end = 2u*(uint32_t)im.nslots;
  hj_lds_reg_src src;
  src.base = lds_win + t*sdw;
  src.bit0 = (first & ~15u) << 3;
  hj_lds_reg_src::reader br;
  br.init(src, hj_pos(start), stop);
  int k = hj_k(start);
  uint32_t c2 = 2u*(uint32_t)hj_slot(start), n = b0;
  if (k == 0) pos[n] = (uint32_t)hj_pos(start) + 1u;          // the lane starts where a block starts
  while (br.before_stop()) {                                 // hj_sync_decode's loop, with a store where a block ends
    const uint32_t w = br.window();
    const uint32_t e = hj_lookup_t(&lds_tabs, hj_field2(k == 0 ? W.dc : W.ac, c2), w);
    const bool packed = HJ_P_BITS(e) != 0 && k + HJ_P_PREFIX(e) < 64 && br.room(hj_pack_bits<Tab>::value);
    br.skip(packed ? HJ_P_BITS(e) : HJ_E_TOT(e));
    const int kn = k + (packed ? HJ_P_ADV(e) : HJ_E_ADV(e));
    const bool done = kn >= 64;
    if (done) {
      n++;
      if (n < total) pos[n] = (uint32_t)br.tell() + 1u;      // (the block behind the segment's last one is nobody's)
    }
    c2 = done ? (c2 + 2u == c2end ? 0u : c2 + 2u) : c2;
    k = done ? 0 : kn;
  }
}

#define HJ_WB_BLOCK 256
__global__ __launch_bounds__(HJ_WB_BLOCK) void hj_write_blocks(const hj_args A, uint32_t blocks_per_image) {
  constexpr int NB = HJ_WB_BLOCK;
  __shared__ __attribute__((aligned(16))) hj_wtables lds_tabs;
  __shared__ uint32_t lds_blk[NB*HJ_BLK_STRIDE];
  __shared__ hj_image s_im;
  __shared__ uint32_t s_dezz[HJ_DEZZ_EXT];
  __shared__ uint2 s_slotd[HJ_MAX_SLOTS];
  const uint32_t img = blockIdx.y, t = threadIdx.x, lane = t & 63u;
  const hj_image im0 = A.images[img];
  hj_stage_image(&s_im, A.images + img);
  if (t < (uint32_t)im0.nslots) s_slotd[t] = hj_block_out::describe(A.images[img], (int)t);
  if (t < HJ_DEZZ_EXT) s_dezz[t] = t < 64 ? 2u*HJ_DEZZ[t] : 128u;
  {
    const hj_tables *T = A.tables + img;
    for (int i = t; i < 2 << HJ_FAST_BITS; i += NB) {
      (&lds_tabs.dc[0][0])[i] = hj_wentry((&T->dc[0][0])[i]);
      (&lds_tabs.ac[0][0])[i] = hj_wentry((&T->ac[0][0])[i] & 0xffffu);
    }
    for (int i = t; i < HJ_L2_BLOCKS*128; i += NB) {
      const uint32_t e = T->l2[i];
      lds_tabs.l2[i] = (uint16_t)(HJ_E_ADV(e) == 64 ? e | (127u << 5) : e);
    }
  }
  uint32_t *blk = lds_blk + t*HJ_BLK_STRIDE;
#pragma unroll
  for (int q = 0; q < 32; q++) blk[q] = 0;
  const uint32_t b = blockIdx.x*NB + t;                      // block number in scan order
  const uint32_t nslots = (uint32_t)im0.nslots;
  uint32_t pos1 = 0, mcu = 0, c = 0;
  uint64_t stop = 0;
  if (b < blocks_per_image) {
    pos1 = A.blk_pos[(size_t)img*(size_t)A.dc_stride + b];
    mcu = b/nslots;
    c = b - mcu*nslots;
    // its restart interval: every segment but an image's last holds as many MCUs as the first
    const uint32_t per = A.segs[im0.seg0].nmcu;
    const uint32_t si = im0.nseg > 1u ? mcu/per : 0u;
    stop = (uint64_t)A.segs[im0.seg0 + (si < im0.nseg ? si : im0.nseg - 1u)].end*8;
  }
  __syncthreads();                                           // tables, s_im, s_dezz, s_slotd
  const bool live = pos1 != 0u;
  const hj_image &im = s_im;
  const uint32_t slot_tbl_bits = hj_slot_tables(im);
  const int tbl = (int)((slot_tbl_bits >> (2u*c)) & 3u);
  const uint32_t *tb_dc = lds_tabs.dc[tbl & 1], *tb_ac = lds_tabs.ac[tbl >> 1];
  hj_gmem_src gsrc;
  gsrc.scan32 = reinterpret_cast<const uint32_t *>(A.scan + im0.scan_off);
  gsrc.ndw = ((im0.scan_len + 16 + 15) & ~15u) >> 2;
  gsrc.dw0 = live ? (pos1 - 1u) >> 5 : 0u;
  hj_gmem_src::reader br;
  br.init(gsrc, live ? (uint64_t)(pos1 - 1u) : 0ull, live ? stop : 0ull);
  uint8_t *blk8 = reinterpret_cast<uint8_t *>(blk);
  int k = 0, dcv = 0, pv = 0, errm = 0;
  uint32_t pz = 128, errbits = 0;
  bool done = !live;
  for (;;) {
    bool running = !done && br.before_stop();
    if (__ballot(running) == 0ull) break;
#pragma unroll
    for (int u = 0; u < HJ_WRITE_UNROLL; u++) {
      if (u) running = !done && br.before_stop();
      if (!running) continue;
      const uint32_t w = br.window();
      const bool isdc = k == 0;
      const uint32_t *tb = isdc ? tb_dc : tb_ac;
      uint32_t e = tb[w >> (32 - HJ_FAST_BITS)];
      if ((e & 31u) == 0u) {                                 // a code longer than 9 bits
        e = lds_tabs.l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
        if (HJ_E_LEN(e) > 16) e |= HJ_W_NOCODE;
      }
      const uint32_t tot = e & 31u, s = (e >> 12) & 15u;
      const uint32_t off = 32u - tot;
      const int vu = (int)__builtin_amdgcn_ubfe(w, off, s);
      const int vs = __builtin_amdgcn_sbfe((int)w, off, s);
      const int v = vu - (vs < 0 ? 0 : (int)((1u << s) - 1u));   // T.81 F.2.2.1 EXTEND; s = 0 gives 0
      dcv = isdc ? v : dcv;
      const int kn = k + (int)((e >> 5) & 127u);
      *reinterpret_cast<hj_i16_alias *>(blk8 + pz) = (int16_t)pv;   // (the previous symbol's value)
      pz = s_dezz[kn - 1];
      pv = v;
      errbits |= e;
      errm = max(errm, kn & 127);
      br.skip((int)tot);
      done = kn >= 64;
      k = done ? 0 : kn;
    }
  }
  *reinterpret_cast<hj_i16_alias *>(blk8 + pz) = (int16_t)pv;       // the store still owed
  const bool whole = live && done;
  blk[32] = ~0u;                                             // "nothing to write out" (the half-dword behind the buffer took the stores that go nowhere)
  if (whole) {
    hj_block_out out;
    out.im = &im;
    out.slotd = s_slotd;
    out.rs = (uint32_t)im0.w0_blocks*64u;
    out.init(mcu);
    blk[32] = out.offset((int)c)*2u;                         // byte offset of the block (a multiple of 128)
    A.dc_diff[(long long)img*A.dc_stride + b] = (int16_t)dcv;
  }
  // a block that ran out of data before its last coefficient: hj_scan has said so already (blocks it did not count
  // have no entry in blk_pos); kept as a second line of defence
  if (live && !done) atomicOr(&A.errors[img], 1u);
  if ((errbits & HJ_W_NOCODE) || errm > 64) atomicOr(&A.errors[img], 2u);
  // the wave's blocks leave together: lanes 8k..8k+7 of a pass write block k's 128 bytes, 16 each
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  typedef __attribute__((address_space(1))) hj_v4u global_v4u;
  int16_t *coef = A.coef + (long long)img*A.coef_stride;
  const uint32_t *wave_blk = lds_blk + (t & ~63u)*HJ_BLK_STRIDE;
  const uint32_t part = lane & 7u;
  for (uint32_t q = lane >> 3; q < 64u; q += 8u) {
    const uint32_t *srcb = wave_blk + q*HJ_BLK_STRIDE;
    const uint32_t off = srcb[32];
    if (off == ~0u) continue;
    hj_v4u v;
    v.x = srcb[4*part]; v.y = srcb[4*part + 1]; v.z = srcb[4*part + 2]; v.w = srcb[4*part + 3];
    __builtin_nontemporal_store(v, (global_v4u *)((uintptr_t)coef + off) + part);
  }
}

// ---- DC prediction (xjpeg.c:480), after the fact ------------------------------------------------------
// dc_diff[b] = the DC difference of block b of the image in scan order (MCU by MCU, slot by slot),
// left there by the write pass.  DC value of a block = sum of the differences of its component's
// blocks from the start of its restart interval up to and including itself (mod 2^16).  One thread
// takes one MCU (its per-component sums), a workgroup a chunk of 1024 MCUs of one segment; as in
// hj_scan, FINAL = false stores chunk totals, FINAL = true adds the chunks before and writes the
// values — to dc_val[slot of the block in the coefficient buffer] (offset/64: the order the
// block-decode kernels and hj_dc_apply find them in).
#define HJ_DC_BLOCK 256
#define HJ_DC_ITEMS 4
#define HJ_DC_CHUNK (HJ_DC_BLOCK*HJ_DC_ITEMS)
template <bool FINAL>
__global__ __launch_bounds__(HJ_DC_BLOCK) void hj_dc_scan(const hj_args A, uint32_t *part_base) {
  __shared__ uint32_t wtot[HJ_DC_BLOCK/64][3];
  __shared__ hj_image s_im;
  const uint32_t gs = blockIdx.x, c = blockIdx.y;
  int img = 0;
  {
    int lo = 0, hi = A.nimages - 1;                         // image of this (batch-global) segment
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (A.images[mid].seg0 <= gs) lo = mid; else hi = mid - 1;
    }
    img = lo;
  }
  const hj_segment sg = A.segs[gs];
  const uint32_t m0 = c*HJ_DC_CHUNK;
  if (m0 >= sg.nmcu) return;
  hj_stage_image(&s_im, A.images + img);
  __syncthreads();
  const hj_image &im = s_im;
  const uint32_t nslots = (uint32_t)im.nslots;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // chunk totals live at part[3*(first MCU chunk slot of the segment + c)]: segments are laid
  // out one after the other by their first MCU, so (image's chunk base + mcu0/CHUNK + si + c) is distinct
  const uint32_t si = gs - im.seg0;
  uint32_t *part = part_base + 3*((size_t)img*A.dc_chunks_per_image + (sg.mcu0/HJ_DC_CHUNK) + si + c);
  const int16_t *dcd = A.dc_diff + (long long)img*A.dc_stride;
  uint32_t slot_comp_bits = 0;
  for (uint32_t q = 0; q < nslots; q++) slot_comp_bits |= (uint32_t)im.slot_comp[q] << (2*q);
  const uint32_t i0 = m0 + threadIdx.x*HJ_DC_ITEMS;
  uint32_t sum[HJ_DC_ITEMS][3];
  uint32_t mine[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < HJ_DC_ITEMS; j++) {
    sum[j][0] = sum[j][1] = sum[j][2] = 0;
    if (i0 + j < sg.nmcu) {
      const int16_t *d = dcd + (size_t)(sg.mcu0 + i0 + j)*nslots;
      for (uint32_t q = 0; q < nslots; q++) {
        const uint32_t comp = (slot_comp_bits >> (2*q)) & 3u, v = (uint32_t)(int)d[q];
        sum[j][0] += comp == 0 ? v : 0; sum[j][1] += comp == 1 ? v : 0; sum[j][2] += comp == 2 ? v : 0;
      }
    }
#pragma unroll
    for (int f = 0; f < 3; f++) mine[f] += sum[j][f];
  }
  uint32_t run[3] = {0, 0, 0};
  if (FINAL) {
    for (uint32_t k = 1; k <= c; k++) {
#pragma unroll
      for (int f = 0; f < 3; f++) run[f] += part[f - 3*(int)k];
    }
  }
  uint32_t inc[3] = {mine[0], mine[1], mine[2]};
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int f = 0; f < 3; f++) {
      const uint32_t x = (uint32_t)__shfl_up((int)inc[f], d);
      if ((int)lane >= d) inc[f] += x;
    }
  }
  if (lane == 63u) {
#pragma unroll
    for (int f = 0; f < 3; f++) wtot[wave][f] = inc[f];
  }
  __syncthreads();
  uint32_t tot[3] = {0, 0, 0};
#pragma unroll
  for (int f = 0; f < 3; f++) run[f] += inc[f] - mine[f];
  for (uint32_t w = 0; w < HJ_DC_BLOCK/64; w++) {
#pragma unroll
    for (int f = 0; f < 3; f++) {
      const uint32_t x = wtot[w][f];
      tot[f] += x;
      if (w < wave) run[f] += x;
    }
  }
  if (!FINAL) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int f = 0; f < 3; f++) part[f] = tot[f];
    }
    return;
  }
  int16_t *dcv = A.dc_val + (long long)img*A.dc_stride;
#pragma unroll
  for (int j = 0; j < HJ_DC_ITEMS; j++) {
    if (i0 + j >= sg.nmcu) break;
    const uint32_t mcu = sg.mcu0 + i0 + j;
    const int16_t *d = dcd + (size_t)mcu*nslots;
    for (uint32_t q = 0; q < nslots; q++) {
      const uint32_t comp = (slot_comp_bits >> (2*q)) & 3u;
      const uint32_t v = (comp == 0 ? run[0] : comp == 1 ? run[1] : run[2]) + (uint32_t)(int)d[q];
      run[0] = comp == 0 ? v : run[0]; run[1] = comp == 1 ? v : run[1]; run[2] = comp == 2 ? v : run[2];
      dcv[hj_block_offset(im, mcu, (int)q) >> 6] = (int16_t)v;
    }
  }
}

// dc_val -> the DC position of every block of the planes (callers that want finished QUANT planes).
// One thread per block slot of the coefficient buffer; slots that hold no block (the layout's holes)
// carry a zero, which is what the cleared planes hold there anyway.
__global__ __launch_bounds__(256) void hj_dc_apply(const hj_args A, int slots_per_image) {
  const int s = blockIdx.x*256 + threadIdx.x;
  if (s >= slots_per_image) return;
  A.coef[(long long)blockIdx.y*A.coef_stride + (long long)s*64] = A.dc_val[(long long)blockIdx.y*A.dc_stride + s];
}

// Start states, segment numbers and "never ran" marks of every subsequence, written on the device instead of
// uploaded (12 bytes per 128 bytes of scan): lane 0 of a segment starts in its known state
// (segment start, k = 0, slot 0, xjpeg.c:612-618), the others at a GUESS — a symbol starts on
// their first byte.  A workgroup takes 4096 subsequences of one segment.
// What the write pass and the DC pass need cleared before they run — the two DC arrays (blocks a damaged stream never
// reaches, slots that hold no block) and the slots at the end of a decimated plane that hold no block.  (Rounds 2-4
// queued these as three or four memsets on a side stream, round 5 first as one launch of their own on the decode's
// stream, then as workgroups of hj_init_states.)
static __device__ __forceinline__ void hj_clear_regions(const hj_clear_args &C, uint64_t me, uint64_t nblocks) {
  typedef __attribute__((address_space(1))) hj_v4u global_v4u;
  const hj_v4u zero = {0u, 0u, 0u, 0u};
  for (int q = 0; q < C.nregions; q++) {
    const hj_clear_region r = C.region[q];
    const uint64_t per_row = r.row_bytes >> 4, total = per_row*r.rows;
    for (uint64_t u = me*256u + threadIdx.x; u < total; u += nblocks*256u) {
      const uint64_t row = per_row == total ? 0 : u/per_row, col = u - row*per_row;
      *(global_v4u *)((uintptr_t)r.base + row*r.stride + (col << 4)) = zero;
    }
  }
}
// (round 5: the clears of a decode — hj_clear_regions — ride this launch as extra workgroups, blockIdx.x >= nsegs:
// one launch less in every decode's chain, ~5 us of a lone frame's)
__global__ __launch_bounds__(256) void hj_init_states(const hj_args A, uint32_t *sub_seg, const uint32_t *verdicts0,
 const hj_clear_args C, uint32_t nsegs) {
  if (blockIdx.x >= nsegs) {
    hj_clear_regions(C, (uint64_t)(blockIdx.x - nsegs)*gridDim.y + blockIdx.y, (uint64_t)(gridDim.x - nsegs)*gridDim.y);
    return;
  }
  const uint32_t gs = blockIdx.x;
  // (also the decode's bookkeeping words, instead of two memsets in front of it: "did anything
  // run in round r", and every image's verdict — what the on-device scan clean-up already
  // found, or nothing)
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    for (int i = threadIdx.x; i < HJ_MAX_ROUNDS; i += 256) A.ran[i] = 0;
    if (A.list_count) for (int i = threadIdx.x; i < 4*A.nimages; i += 256) A.list_count[i*HJ_LIST_CSTRIDE] = 0;
    for (int i = threadIdx.x; i < A.nimages; i += 256) A.errors[i] = verdicts0 ? verdicts0[i] : 0u;
  }
  int lo = 0, hi = A.nimages - 1;                           // image of this (batch-global) segment
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (A.images[mid].seg0 <= gs) lo = mid; else hi = mid - 1;
  }
  const hj_image *im = A.images + lo;
  const uint32_t sub0 = im->sub0, seg0 = im->seg0, si = gs - seg0;
  const hj_segment sg = A.segs[gs];
  uint64_t *S = A.S + sub0 + seg0 + sg.sub0 + si;           // nsub + 1 entries
  uint32_t *ss = sub_seg + sub0 + sg.sub0;
  uint64_t *last_in = A.last_in + sub0 + sg.sub0;
  const uint32_t k0 = blockIdx.y << 12;
  if (k0 >= sg.nsub) return;
  const uint32_t k1 = k0 + 4096u < sg.nsub ? k0 + 4096u : sg.nsub;
  for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
    ss[k] = si;
    S[k] = hj_pack((uint64_t)(sg.start + (k << A.sub_log2))*8, 0, 0);
    last_in[k] = ~0ull;                                      // "never ran"
  }
  if (k1 == sg.nsub && threadIdx.x == 0) S[sg.nsub] = 0;
}
extern "C" int hj_launch_init(const hj_args *A, int total_segs, int max_nsub, const uint32_t *verdicts0,
 const hj_clear_args *C, void *stream) {
  const unsigned gy = (unsigned)((max_nsub + 4095) >> 12);
  // workgroups for the clears: four 16-byte units per lane, at most 4096 of them
  uint64_t units = 0;
  for (int k = 0; C && k < C->nregions; k++) units += (C->region[k].row_bytes >> 4)*C->region[k].rows;
  uint64_t want = (units + 1023)/1024;
  if (want > 4096) want = 4096;
  const unsigned cx = (unsigned)((want + gy - 1)/gy);
  hj_clear_args none;
  none.nregions = 0;
  hipLaunchKernelGGL(hj_init_states, dim3(total_segs + cx, gy), dim3(256), 0, (hipStream_t)stream, *A,
   const_cast<uint32_t *>(A->sub_seg), verdicts0, C ? *C : none, (uint32_t)total_segs);
  return (int)hipGetLastError();
}
extern "C" int hj_launch_round(const hj_args *A, int max_nsub, int round, int max_iters, int lean,
 void *stream) {
  // JGA_HUFF_LITE: 0 = the first run of all counts like any other; n > 0 = it is a lite run that
  // starts n - 1 bytes into its subsequence (A/B knob)
  // (default 49: measured 2.45 ms / 5 rounds from the first bit, 2.42-2.47 ms / 4 rounds from byte 48,
  // a lone 1080p frame 0.77 -> 0.70 ms: profiles/r2_lite_first_run_ab.txt)
  static const int lite_first = jga_tune("JGA_HUFF_LITE") ? atoi(jga_tune("JGA_HUFF_LITE")) : 49;
  dim3 grid((max_nsub + HJ_BLOCK - 1)/HJ_BLOCK, A->nimages);
  if (lean && A->wide && !A->wide_shared) {   // ... and with the 12-bit AC tables (small batches: hj_wide_ac)
    hipLaunchKernelGGL((hj_sync_round<hj_lds_reg_src, hj_ltables_wide>), grid, dim3(HJ_BLOCK), 0, (hipStream_t)stream, *A, round, max_iters, lite_first);
  }
  else if (lean) {                               // rows read through registers
    hipLaunchKernelGGL(hj_sync_round<hj_lds_reg_src>, grid, dim3(HJ_BLOCK), 0, (hipStream_t)stream, *A, round, max_iters, lite_first);
  }
  else hipLaunchKernelGGL(hj_sync_round<hj_lds_src>, grid, dim3(HJ_BLOCK), 0, (hipStream_t)stream, *A, round, max_iters, lite_first);
  return (int)hipGetLastError();
}
extern "C" int hj_launch_list_round(const hj_args *A, int max_nsub, int round, int ordinal, int max_iters, int rebuild, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const unsigned groups = (unsigned)((max_nsub + HJ_BLOCK - 1)/HJ_BLOCK);
  if (rebuild) {
    // (the lists are made from the states; rebuild == 2: whatever an earlier round left on them is dropped first —
    // hj_init_states has zeroed the counters of a decode's first list round)
    if (rebuild > 1 && hipMemsetAsync(A->list_count, 0, 4*sizeof(uint32_t)*HJ_LIST_CSTRIDE*(size_t)A->nimages, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(hj_list_build, dim3(groups, A->nimages), dim3(256), 0, st, *A, round);
  }
  // A fifth of the lanes still moves after the first round's three steps, a third of those a step later: the first
  // list round of a batch that fills the device gets a workgroup per 4 groups and reads its rows from global memory,
  // the later ones one per 8 groups (grid-stride beyond that) with the rows in LDS.
  // The 12-bit tables (85 KB of LDS per workgroup: one per CU) where the expected workgroups — a fifth of the groups,
  // a third of that per list round since — are at most one per CU; a small batch that brought a set per image takes
  // them in every round.
  const size_t all_groups = (size_t)groups*(size_t)A->nimages;
  const bool crowded = rebuild == 1 && (!A->wide || A->wide_shared) && all_groups >= 1024;
  bool wide = A->wide != nullptr;
  if (wide && A->wide_shared) {
    double expect = 0.205*(double)all_groups;
    for (int k = 0; k < ordinal && k < 8; k++) expect /= 3.1;
    wide = !crowded && expect <= 256.0;
  }
  unsigned gx = crowded ? (groups + 3)/4 : (groups + 7)/8;
  if (gx > 64 && !crowded) gx = 64;
  const dim3 grid(gx, A->nimages);
  if (wide) hipLaunchKernelGGL((hj_sync_list<hj_ltables_wide, true>), grid, dim3(HJ_LIST_BLOCK), 0, st, *A, round, max_iters);
  else if (crowded) hipLaunchKernelGGL((hj_sync_list<hj_ltables, false>), grid, dim3(HJ_LIST_BLOCK), 0, st, *A, round, max_iters);
  else hipLaunchKernelGGL((hj_sync_list<hj_ltables, true>), grid, dim3(HJ_LIST_BLOCK), 0, st, *A, round, max_iters);
  return (int)hipGetLastError();
}
extern "C" int hj_launch_scan(const hj_args *A, int total_segs, int max_nsub, void *stream) {
  const dim3 grid(total_segs, (max_nsub + (1 << HJ_SCAN_CHUNK_LOG2) - 1) >> HJ_SCAN_CHUNK_LOG2);
  if (grid.y > 1) hipLaunchKernelGGL(hj_scan<false>, grid, dim3(HJ_SCAN_BLOCK), 0, (hipStream_t)stream, *A);
  hipLaunchKernelGGL(hj_scan<true>, grid, dim3(HJ_SCAN_BLOCK), 0, (hipStream_t)stream, *A);
  return (int)hipGetLastError();
}
extern "C" size_t hj_scan_part_bytes(size_t total_segs, size_t total_subs) {
  return 4*(total_segs + (total_subs >> HJ_SCAN_CHUNK_LOG2) + 2);
}
extern "C" int hj_launch_write_blocks(const hj_args *A, int max_nsub, int blocks_per_image, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((max_nsub + HJ_LIST_BLOCK - 1)/HJ_LIST_BLOCK, A->nimages);
  if (A->wide && !A->wide_shared) hipLaunchKernelGGL(hj_block_starts<hj_ltables_wide>, grid, dim3(HJ_LIST_BLOCK), 0, st, *A);
  else hipLaunchKernelGGL(hj_block_starts<hj_ltables>, grid, dim3(HJ_LIST_BLOCK), 0, st, *A);
  hipLaunchKernelGGL(hj_write_blocks, dim3((blocks_per_image + HJ_WB_BLOCK - 1)/HJ_WB_BLOCK, A->nimages), dim3(HJ_WB_BLOCK), 0, st, *A,
   (uint32_t)blocks_per_image);
  return (int)hipGetLastError();
}
extern "C" int hj_launch_write(const hj_args *A, int max_nsub, void *stream) {
  dim3 grid((max_nsub + HJ_WRITE_BLOCK - 1)/HJ_WRITE_BLOCK, A->nimages);
  hipLaunchKernelGGL(hj_write, grid, dim3(HJ_WRITE_BLOCK), 0, (hipStream_t)stream, *A);
  return (int)hipGetLastError();
}
// chunks of HJ_DC_CHUNK MCUs an image's segments can take: every segment at most one partial chunk more
extern "C" int hj_dc_chunks_per_image(int total_mcus, int max_segs_per_image) {
  return total_mcus/HJ_DC_CHUNK + max_segs_per_image + 1;
}
extern "C" int hj_launch_dc(const hj_args *A, int total_segs, int max_seg_mcus, uint32_t *part, int apply_slots,
 void *stream) {
  const dim3 grid(total_segs, (max_seg_mcus + HJ_DC_CHUNK - 1)/HJ_DC_CHUNK);
  if (grid.y > 1) hipLaunchKernelGGL(hj_dc_scan<false>, grid, dim3(HJ_DC_BLOCK), 0, (hipStream_t)stream, *A, part);
  hipLaunchKernelGGL(hj_dc_scan<true>, grid, dim3(HJ_DC_BLOCK), 0, (hipStream_t)stream, *A, part);
  if (apply_slots > 0) {
    hipLaunchKernelGGL(hj_dc_apply, dim3((apply_slots + 255)/256, A->nimages), dim3(256), 0, (hipStream_t)stream, *A, apply_slots);
  }
  return (int)hipGetLastError();
}
