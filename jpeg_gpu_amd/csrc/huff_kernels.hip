// huff_kernels.hip — GPU-parallel baseline-JPEG entropy decode (gfx950).
// Algorithm and state definitions: huff_common.h.  Three kernels:
//   hj_sync_round  one lane per subsequence; re-decodes when its start state moved.
//                  A workgroup iterates internally (states handed lane-to-lane
//                  through LDS) until none of its lanes moves, so most of the
//                  propagation needs no extra launch.
//   hj_scan        one workgroup per restart segment: exclusive prefix sums of
//                  block counts and DC-difference sums over the segment's lanes
//   hj_write       one lane per subsequence: final decode.  Blocks a lane decodes
//                  completely are assembled in LDS and leave as one 128-byte line;
//                  only the pieces of blocks that straddle lanes are scattered.
// Integer/byte work.  The scan bytes a workgroup needs (its 256 consecutive
// subsequences, ~33 KB) are staged into LDS with coalesced 16-byte loads, padded
// by one dword per 128 bytes so that lanes reading at a 128-byte stride hit
// different banks; the two-level Huffman lookup of the image (14 KB) sits next to them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "huff_common.h"
#include "huff_kernels.h"

#define HJ_BLOCK 256
// staged window: 256 subsequences (segments lie back to back in the clean stream) + look-ahead
#define HJ_WIN_BYTES (HJ_BLOCK*HJ_SUB_BYTES + 96)
#define HJ_WIN_DWORDS ((HJ_WIN_BYTES + (HJ_WIN_BYTES >> 7)*4)/4 + 8)
#define HJ_BLK_STRIDE 36            /* dwords per lane's block buffer (144 B) */

__device__ const uint8_t HJ_DEZZ[64] = {     // T.81 Figure A.6: zig-zag index -> natural index
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
  13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59,
  52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Bit source over the LDS image of [w0, w1) of the image's clean scan: big-endian
// dwords (byte-swapped once when staged), one pad dword per 32 so that lanes reading
// at a 128-byte stride hit different banks.
struct hj_lds_src {
  const uint32_t *lds;
  uint32_t w0_bits;                  // first bit of the window (w0 is 16-byte aligned)
  __device__ __forceinline__ uint32_t window32(uint32_t p) const {
    const uint32_t r = p - w0_bits, i = r >> 5, j = i + 1;
    const uint64_t v = ((uint64_t)lds[i + (i >> 5)] << 32) | lds[j + (j >> 5)];
    return (uint32_t)(v >> (32 - (r & 31)));
  }
};

struct hj_lane_ctx {                 // what a lane knows about its subsequence
  uint32_t g, si, i;                 // batch-global subsequence, image-local segment, index in segment
  uint32_t seg_start, seg_end, seg_nsub, seg_mcu0, seg_nmcu;
  uint32_t stop_byte;                // end of this subsequence (raw byte, exclusive)
};

// Common prologue: lane context, staged tables + scan window.  Returns false for
// lanes beyond the image's last subsequence (they still took part in staging).
static __device__ __forceinline__ bool hj_prologue(const hj_args &A, const hj_image &im,
 hj_tables *lds_tabs, uint32_t *lds_win, uint32_t *lds_misc, hj_lds_src &src,
 hj_lane_ctx &L) {
  const uint32_t li = blockIdx.x*HJ_BLOCK + threadIdx.x;
  const bool in_range = li < im.nsub;
  L.g = 0; L.si = 0; L.i = 0; L.stop_byte = 0;
  L.seg_start = L.seg_end = L.seg_nsub = L.seg_mcu0 = L.seg_nmcu = 0;
  if (in_range) {
    L.g = im.sub0 + li;
    L.si = A.sub_seg[L.g];
    const hj_segment sg = A.segs[im.seg0 + L.si];
    L.i = li - sg.sub0;
    L.seg_start = sg.start; L.seg_end = sg.end; L.seg_nsub = sg.nsub;
    L.seg_mcu0 = sg.mcu0; L.seg_nmcu = sg.nmcu;
    L.stop_byte = sg.start + (L.i + 1)*HJ_SUB_BYTES;
    if (L.stop_byte > sg.end) L.stop_byte = sg.end;
    if (threadIdx.x == 0) lds_misc[0] = (sg.start + L.i*HJ_SUB_BYTES) & ~15u;   // window start
    if (threadIdx.x == HJ_BLOCK - 1 || li + 1 == im.nsub) lds_misc[1] = L.stop_byte;
  }
  // tables of this image -> LDS (14 KB, 16-byte chunks)
  {
    const uint4 *tsrc = reinterpret_cast<const uint4 *>(A.tables + blockIdx.y);
    uint4 *tdst = reinterpret_cast<uint4 *>(lds_tabs);
    for (int k = threadIdx.x; k < (int)(sizeof(hj_tables)/16); k += HJ_BLOCK) tdst[k] = tsrc[k];
  }
  __syncthreads();
  const uint32_t w0 = lds_misc[0];
  uint32_t w1 = lds_misc[1] + 48;                     // look-ahead of the last lane
  const uint32_t padded = (im.scan_len + 16 + 15) & ~15u;   // bytes present in the batch buffer
  if (w1 > padded) w1 = padded;
  if (w1 > w0 + HJ_WIN_BYTES) w1 = w0 + HJ_WIN_BYTES;
  const uint4 *gsrc = reinterpret_cast<const uint4 *>(A.scan + im.scan_off + w0);
  const uint32_t nchunks = (w1 - w0 + 15) >> 4;
  for (uint32_t c = threadIdx.x; c < nchunks; c += HJ_BLOCK) {
    const uint4 v = gsrc[c];
    const uint32_t a = c << 4;
    uint32_t *d = lds_win + ((a + ((a >> 7) << 2)) >> 2);
    d[0] = __builtin_bswap32(v.x); d[1] = __builtin_bswap32(v.y);
    d[2] = __builtin_bswap32(v.z); d[3] = __builtin_bswap32(v.w);
  }
  __syncthreads();
  src.lds = lds_win;
  src.w0_bits = w0 << 3;
  return in_range;
}

// Result of a run, packed for LDS.
struct hj_run16 {
  uint64_t end_state;
  uint16_t nblocks;
  int16_t dcsum[3];
};

__global__ __launch_bounds__(HJ_BLOCK) void hj_sync_round(const hj_args A, int round, int max_iters) {
  __shared__ __attribute__((aligned(16))) hj_tables lds_tabs;
  __shared__ uint32_t lds_win[HJ_WIN_DWORDS];
  __shared__ uint64_t lds_S[HJ_BLOCK + 1];       // start state of each subsequence of the group
  __shared__ hj_run16 lds_R[HJ_BLOCK];           // result of its latest run
  __shared__ uint32_t lds_stop[HJ_BLOCK];        // stop byte | bit 31: has a successor in its segment
  __shared__ uint32_t lds_sidx[HJ_BLOCK];        // its entry of the global S array
  __shared__ uint8_t lds_dirty[HJ_BLOCK], lds_ran[HJ_BLOCK];
  __shared__ uint16_t lds_act[HJ_BLOCK];
  __shared__ uint32_t lds_wcnt[HJ_BLOCK/64];
  __shared__ uint32_t lds_misc[4];
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
  if (t == 0) s_im = im;
  hj_lds_src src;
  hj_lane_ctx L;
  const bool on = hj_prologue(A, im, &lds_tabs, lds_win, lds_misc, src, L);
  const uint32_t sidx = L.g + im.seg0 + L.si;          // this subsequence's entry of S
  {
    const uint64_t st = on ? A.S[sidx] : 0;
    lds_S[t] = st;
    lds_dirty[t] = on && st != A.last_in[L.g];
    lds_ran[t] = 0;
    lds_stop[t] = L.stop_byte | (on && L.i + 1 < L.seg_nsub ? 0x80000000u : 0u);
    lds_sidx[t] = sidx;
  }
  __syncthreads();
  // Iterate inside the group.  Each iteration packs the subsequences whose start state
  // moved into dense waves (any lane can decode any subsequence of the staged window),
  // so the work follows the number of runs, not iterations x 256.
  const uint32_t lane = t & 63, wave = t >> 6;
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
      lds_act[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)t;
      lds_dirty[t] = 0;
    }
    __syncthreads();
    if (t < total) {
      const uint32_t sub = lds_act[t];
      const uint64_t start = lds_S[sub];
      const uint32_t sb = lds_stop[sub];
      const hj_run r = hj_sync_decode(src, s_im, &lds_tabs, start, (uint64_t)(sb & 0x7fffffffu)*8);
      hj_run16 r16;
      r16.end_state = r.end_state; r16.nblocks = (uint16_t)r.nblocks;
      r16.dcsum[0] = r.dcsum[0]; r16.dcsum[1] = r.dcsum[1]; r16.dcsum[2] = r.dcsum[2];
      lds_R[sub] = r16;
      lds_ran[sub] = 1;
      if (sb & 0x80000000u) {
        if (sub + 1 < HJ_BLOCK) {
          if (lds_S[sub + 1] != r.end_state) { lds_S[sub + 1] = r.end_state; lds_dirty[sub + 1] = 1; }
        }
        else A.S[lds_sidx[sub] + 1] = r.end_state;         // first subsequence of the next group
      }
    }
    __syncthreads();
  }
  // Publish.  Entry t > 0 of S is written only by this group (it was handed over in
  // LDS); entry 0 belongs to the previous group's last lane and is left alone.
  if (on) {
    const uint64_t st = lds_S[t];
    if (t > 0 && st != A.S[sidx]) A.S[sidx] = st;
    if (lds_ran[t]) {
      const hj_run16 r16 = lds_R[t];
      hj_run r;
      r.end_state = r16.end_state; r.nblocks = r16.nblocks; r.error = 0;
      r.dcsum[0] = r16.dcsum[0]; r.dcsum[1] = r16.dcsum[1]; r.dcsum[2] = r16.dcsum[2];
      A.R[L.g] = r;
      // clean: its latest run started from st.  dirty: the state moved after that run
      // began, so whatever last_in held before must not make it look settled.
      A.last_in[L.g] = lds_dirty[t] ? ~0ull : st;
    }
  }
  if (__syncthreads_or(on && lds_ran[t]) && t == 0) atomicOr(&A.ran[round], 1u);
}

// Exclusive prefix sums over the lanes of one segment.  Sequential over chunks of
// 256 lanes, Hillis-Steele inside a chunk.
__global__ __launch_bounds__(HJ_BLOCK) void hj_scan(const hj_args A) {
  __shared__ uint32_t sb[HJ_BLOCK];
  __shared__ int sd[3][HJ_BLOCK];
  // blockIdx.x = batch-global segment; find its image (few images: linear search)
  const uint32_t gs = blockIdx.x;
  int img = 0;
  while (img + 1 < A.nimages && A.images[img + 1].seg0 <= gs) img++;
  const hj_image im = A.images[img];
  const hj_segment sg = A.segs[gs];
  const uint32_t si = gs - im.seg0;
  const uint32_t total = sg.nmcu*(uint32_t)im.nslots;
  uint32_t base_b = 0;
  int base_d[3] = {0, 0, 0};
  bool bad = false;
  for (uint32_t c0 = 0; c0 < sg.nsub; c0 += HJ_BLOCK) {
    const uint32_t i = c0 + threadIdx.x;
    const bool on = i < sg.nsub;
    const uint32_t g = im.sub0 + sg.sub0 + i;
    hj_run r;
    r.nblocks = 0; r.dcsum[0] = r.dcsum[1] = r.dcsum[2] = 0;
    if (on) r = A.R[g];
    sb[threadIdx.x] = r.nblocks;
    sd[0][threadIdx.x] = r.dcsum[0]; sd[1][threadIdx.x] = r.dcsum[1]; sd[2][threadIdx.x] = r.dcsum[2];
    __syncthreads();
    for (int d = 1; d < HJ_BLOCK; d <<= 1) {
      uint32_t vb = 0;
      int v0 = 0, v1 = 0, v2 = 0;
      if ((int)threadIdx.x >= d) {
        vb = sb[threadIdx.x - d];
        v0 = sd[0][threadIdx.x - d]; v1 = sd[1][threadIdx.x - d]; v2 = sd[2][threadIdx.x - d];
      }
      __syncthreads();
      sb[threadIdx.x] += vb;
      sd[0][threadIdx.x] += v0; sd[1][threadIdx.x] += v1; sd[2][threadIdx.x] += v2;
      __syncthreads();
    }
    if (on) {
      const uint32_t excl = base_b + sb[threadIdx.x] - r.nblocks;
      A.B[g] = excl;
      A.D[3*g + 0] = (int16_t)(base_d[0] + sd[0][threadIdx.x] - r.dcsum[0]);
      A.D[3*g + 1] = (int16_t)(base_d[1] + sd[1][threadIdx.x] - r.dcsum[1]);
      A.D[3*g + 2] = (int16_t)(base_d[2] + sd[2][threadIdx.x] - r.dcsum[2]);
      // the slot a lane starts in must agree with the number of blocks before it
      const uint64_t st = A.S[g + im.seg0 + si];
      if (excl < total && hj_slot(st) != (int)(excl % (uint32_t)im.nslots)) bad = true;
    }
    base_b += sb[HJ_BLOCK - 1];
    base_d[0] += sd[0][HJ_BLOCK - 1]; base_d[1] += sd[1][HJ_BLOCK - 1]; base_d[2] += sd[2][HJ_BLOCK - 1];
    __syncthreads();
  }
  if (base_b < total) bad = true;                            // data ran out before the last MCU
  if (bad) atomicOr(&A.errors[img], 1u);
}

// Write sink.  A block this lane decodes from its first coefficient ("owned") is
// built in the lane's LDS buffer and stored as one 128-byte line when complete;
// everything else (the tail of a block begun by an earlier lane, the head of a
// block this lane cannot finish) goes straight to the pre-zeroed planes as
// 2-byte stores — disjoint positions, so no ordering is needed between lanes.
struct hj_write_sink {
  const hj_image *im;
  int16_t *coef;
  uint32_t *blk;                     // this lane's 36-dword LDS buffer (zero between blocks)
  uint32_t mcu0, b0, total;
  int pred0, pred1, pred2;           // DC predictors (scalars: no indexed private array)
  int64_t off;
  bool ok, owned;

  __device__ __forceinline__ void flush_owned() {
    uint4 *dst = reinterpret_cast<uint4 *>(coef + off);
    const uint4 *s = reinterpret_cast<const uint4 *>(blk);
    const uint4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < 8; n++) {
      dst[n] = s[n];
      reinterpret_cast<uint4 *>(blk)[n] = z;
    }
  }
  __device__ __forceinline__ void block_begin(uint32_t n, int c, int k) {
    if (n && owned && ok) flush_owned();               // the previous block is complete
    const uint32_t b = b0 + n;
    ok = b < total;
    owned = k == 0;
    if (ok) off = hj_block_offset(*im, mcu0 + b/(uint32_t)im->nslots, c);
  }
  __device__ __forceinline__ void put(int idx, int v) {
    if (!ok) return;
    if (owned) reinterpret_cast<int16_t *>(blk)[idx] = (int16_t)v;
    else coef[off + idx] = (int16_t)v;
  }
  __device__ __forceinline__ void dc(int comp, int v) {
    pred0 += comp == 0 ? v : 0;
    pred1 += comp == 1 ? v : 0;
    pred2 += comp == 2 ? v : 0;
    put(0, (int16_t)(comp == 0 ? pred0 : comp == 1 ? pred1 : pred2));   // wraps like xjpeg.c:480
  }
  __device__ __forceinline__ void ac(int k, int v) { put(HJ_DEZZ[k], v); }
  // end of the run with the current block incomplete (k != 0): a later lane adds the
  // rest, so this lane's part is scattered; a complete block was flushed by block_begin
  __device__ __forceinline__ void finish(int k) {
    if (!(ok && owned && k != 0)) return;
    const int16_t *s = reinterpret_cast<const int16_t *>(blk);
    for (int n = 0; n < 64; n++) {
      const int16_t v = s[n];
      if (v) coef[off + n] = v;
    }
  }
};

__global__ __launch_bounds__(HJ_BLOCK) void hj_write(const hj_args A) {
  __shared__ __attribute__((aligned(16))) hj_tables lds_tabs;
  __shared__ uint32_t lds_win[HJ_WIN_DWORDS];
  __shared__ __attribute__((aligned(16))) uint32_t lds_blk[HJ_BLOCK*HJ_BLK_STRIDE];
  __shared__ uint32_t lds_misc[4];
  __shared__ hj_image s_im;
  const hj_image im0 = A.images[blockIdx.y];
  if (blockIdx.x*HJ_BLOCK >= im0.nsub) return;              // grid.x covers the largest image
  if (threadIdx.x == 0) s_im = im0;
  for (int k = threadIdx.x; k < HJ_BLOCK*HJ_BLK_STRIDE; k += HJ_BLOCK) lds_blk[k] = 0;
  hj_lds_src src;
  hj_lane_ctx L;
  const bool on = hj_prologue(A, im0, &lds_tabs, lds_win, lds_misc, src, L);   // syncs: s_im, lds_blk ready
  if (!on) return;
  const hj_image &im = s_im;
  const uint32_t total = L.seg_nmcu*(uint32_t)im.nslots;
  const uint32_t b0 = A.B[L.g];
  if (b0 >= total) return;
  const uint32_t sidx = L.g + im.seg0 + L.si;
  const uint64_t start = A.S[sidx];
  const uint64_t stop = L.i + 1 < L.seg_nsub ? hj_pos(A.S[sidx + 1]) : (uint64_t)L.seg_end*8;
  hj_write_sink ws;
  ws.im = &im;
  ws.coef = A.coef + (long long)blockIdx.y*A.coef_stride;
  ws.blk = lds_blk + threadIdx.x*HJ_BLK_STRIDE;
  ws.mcu0 = L.seg_mcu0; ws.b0 = b0; ws.total = total;
  ws.pred0 = A.D[3*L.g + 0]; ws.pred1 = A.D[3*L.g + 1]; ws.pred2 = A.D[3*L.g + 2];
  ws.off = 0; ws.ok = false; ws.owned = false;
  const hj_run r = hj_decode(src, im, &lds_tabs, start, stop, total - b0, ws);
  if (r.error) atomicOr(&A.errors[blockIdx.y], 2u);
}

extern "C" int hj_launch_round(const hj_args *A, int max_nsub, int round, int max_iters, void *stream) {
  dim3 grid((max_nsub + HJ_BLOCK - 1)/HJ_BLOCK, A->nimages), block(HJ_BLOCK);
  hipLaunchKernelGGL(hj_sync_round, grid, block, 0, (hipStream_t)stream, *A, round, max_iters);
  return (int)hipGetLastError();
}
extern "C" int hj_launch_scan(const hj_args *A, int total_segs, void *stream) {
  hipLaunchKernelGGL(hj_scan, dim3(total_segs), dim3(HJ_BLOCK), 0, (hipStream_t)stream, *A);
  return (int)hipGetLastError();
}
extern "C" int hj_launch_write(const hj_args *A, int max_nsub, void *stream) {
  dim3 grid((max_nsub + HJ_BLOCK - 1)/HJ_BLOCK, A->nimages), block(HJ_BLOCK);
  hipLaunchKernelGGL(hj_write, grid, block, 0, (hipStream_t)stream, *A);
  return (int)hipGetLastError();
}
