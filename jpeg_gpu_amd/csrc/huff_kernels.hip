// huff_kernels.hip — GPU-parallel baseline-JPEG entropy decode (gfx950).
// Algorithm and state definitions: huff_common.h.  Three kernels:
//   hj_sync_round  one lane per subsequence; re-decodes when its start state moved
//   hj_scan        one workgroup per restart segment: exclusive prefix sums of
//                  block counts and DC-difference sums over the segment's lanes
//   hj_write       one lane per subsequence: final decode, coefficients scattered
//                  into the packed planes (pre-zeroed), DC integrated
// Integer/byte work, HBM/latency bound; the Huffman lookup tables of the image a
// workgroup works on are staged in LDS (6 x 1 KB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "huff_common.h"
#include "huff_kernels.h"

#define HJ_BLOCK 256

__device__ const uint8_t HJ_DEZZ[64] = {     // T.81 Figure A.6: zig-zag index -> natural index
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
  13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59,
  52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Stage the six fast tables of image `img` in LDS; fast[] points at the copies.
static __device__ __forceinline__ void stage_tables(const hj_table *tabs, uint16_t *lds,
 const uint16_t **fast) {
  constexpr int WORDS = (1 << HJ_FAST_BITS)/2;           // dwords per table
  uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
  for (int t = 0; t < 6; t++) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(tabs[t].fast);
    for (int i = threadIdx.x; i < WORDS; i += HJ_BLOCK) dst[t*WORDS + i] = src[i];
    fast[t] = lds + t*(1 << HJ_FAST_BITS);
  }
  __syncthreads();
}

__global__ __launch_bounds__(HJ_BLOCK) void hj_sync_round(const hj_args A, int round) {
  __shared__ uint16_t lds_fast[6*(1 << HJ_FAST_BITS)];
  const hj_image im = A.images[blockIdx.y];
  const uint32_t li = blockIdx.x*HJ_BLOCK + threadIdx.x;      // image-local subsequence
  const bool in_range = li < im.nsub;
  uint64_t start = 0;
  uint32_t g = 0, si = 0;
  bool need = false;
  if (in_range) {
    g = im.sub0 + li;
    si = A.sub_seg[g];
    start = A.S[g + im.seg0 + si];
    need = start != A.last_in[g];
  }
  if (!__syncthreads_or(need)) return;                        // nothing moved for this workgroup
  const hj_table *tabs = A.tables + 6*blockIdx.y;
  const uint16_t *fast[6];
  stage_tables(tabs, lds_fast, fast);
  if (!need) return;
  const hj_segment sg = A.segs[im.seg0 + si];
  const uint32_t i = li - sg.sub0;
  uint32_t stop_byte = sg.start + (i + 1)*HJ_SUB_BYTES;
  if (stop_byte > sg.end) stop_byte = sg.end;
  hj_null_sink ns;
  const hj_run r = hj_decode(A.scan + im.scan_off, sg.end, im, tabs, fast, start,
   (uint64_t)stop_byte*8, 0xFFFFFFFFu, ns);
  A.last_in[g] = start;
  A.R[g] = r;
  if (i + 1 < sg.nsub) A.S[g + im.seg0 + si + 1] = r.end_state;
  if (__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicOr(&A.ran[round], 1u);
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

struct hj_write_sink {
  const hj_image *im;
  int16_t *coef;
  uint32_t mcu0, b0, total;
  int16_t pred[3];
  int64_t off;
  bool ok;
  __device__ __forceinline__ void block_begin(uint32_t n, int c) {
    const uint32_t b = b0 + n;
    ok = b < total;
    if (ok) off = hj_block_offset(*im, mcu0 + b/(uint32_t)im->nslots, c);
  }
  __device__ __forceinline__ void dc(int comp, int v) {
    pred[comp] = (int16_t)(pred[comp] + v);
    if (ok) coef[off] = pred[comp];
  }
  __device__ __forceinline__ void ac(int k, int v) {
    if (ok) coef[off + HJ_DEZZ[k]] = (int16_t)v;
  }
};

__global__ __launch_bounds__(HJ_BLOCK) void hj_write(const hj_args A) {
  __shared__ uint16_t lds_fast[6*(1 << HJ_FAST_BITS)];
  __shared__ hj_image s_im;
  if (threadIdx.x == 0) s_im = A.images[blockIdx.y];
  const hj_table *tabs = A.tables + 6*blockIdx.y;
  const uint16_t *fast[6];
  stage_tables(tabs, lds_fast, fast);                        // also publishes s_im
  const hj_image &im = s_im;
  const uint32_t li = blockIdx.x*HJ_BLOCK + threadIdx.x;
  if (li >= im.nsub) return;
  const uint32_t g = im.sub0 + li;
  const uint32_t si = A.sub_seg[g];
  const hj_segment sg = A.segs[im.seg0 + si];
  const uint32_t i = li - sg.sub0;
  const uint32_t total = sg.nmcu*(uint32_t)im.nslots;
  const uint32_t b0 = A.B[g];
  if (b0 >= total) return;
  const uint64_t start = A.S[g + im.seg0 + si];
  const uint64_t stop = i + 1 < sg.nsub ? hj_pos(A.S[g + im.seg0 + si + 1]) : (uint64_t)sg.end*8;
  hj_write_sink ws;
  ws.im = &im;
  ws.coef = A.coef + (long long)blockIdx.y*A.coef_stride;
  ws.mcu0 = sg.mcu0; ws.b0 = b0; ws.total = total;
  ws.pred[0] = A.D[3*g + 0]; ws.pred[1] = A.D[3*g + 1]; ws.pred[2] = A.D[3*g + 2];
  ws.off = 0; ws.ok = false;
  const hj_run r = hj_decode(A.scan + im.scan_off, sg.end, im, tabs, fast, start, stop,
   total - b0, ws);
  if (r.error) atomicOr(&A.errors[blockIdx.y], 2u);
}

extern "C" int hj_launch_round(const hj_args *A, int max_nsub, int round, void *stream) {
  dim3 grid((max_nsub + HJ_BLOCK - 1)/HJ_BLOCK, A->nimages), block(HJ_BLOCK);
  hipLaunchKernelGGL(hj_sync_round, grid, block, 0, (hipStream_t)stream, *A, round);
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
