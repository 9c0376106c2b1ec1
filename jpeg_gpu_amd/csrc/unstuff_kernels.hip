// unstuff_kernels.hip — the scan's byte-level clean-up on the GPU (gfx950).
//
// Before a Huffman decoder can treat a position in the entropy-coded segment as a plain bit
// count, the marker escapes have to go: stuffed zeros (FF 00 -> FF), fill bytes (FF FF ..),
// and the RSTn markers, which also cut the scan into restart intervals (T.81 B.1.1.5, F.1.2.3;
// the reference handles them symbol by symbol inside its bit reader, src/xjpeg.c:113-127, and
// per interval at 593-629).  huff_prepare.cpp's hj_prepare_scan does this on the host — one
// core unstuffs ~12 GB/s (as fast as it could memcpy), and a GPU that decodes 16 000 4K frames a
// second wants 44 GB/s of it: fine for one GPU and 16 cores, not for eight GPUs sharing them.
// Here the raw bytes go up as they are and the device produces exactly what hj_prepare_scan
// would have: the clean stream, the segment table, the image's clean length and subsequence
// count.  The host's share of a frame shrinks to the marker parse and one memcpy.
//
// Everything is decided locally, because the second byte of an escape is never FF:
//   lead(p)   = raw[p] == FF                      next(p) = raw[p+1], or EOI past the end
//   drop(p)   = lead(p) && next(p) != 00          a marker's FF / a fill byte
//             || raw[p-1] == FF && raw[p] != FF   a stuffed zero / a marker's code byte
//   rst(p)    = lead(p) && next(p) in D0..D7
//   term(p)   = lead(p) && next(p) not in {00, FF, D0..D7}
// The scan ends at the first term, or at the RSTn that would open one interval more than the
// frame has (the host loop's `mcu0 < total_mcus` test); RSTn number k must carry counter k & 7.
// Four launches: per-chunk counts -> per-image prefix over the chunks + the end -> scatter
// (bytes through an LDS line, so that global writes are whole dwords) -> segment table.
// Integer/byte work, HBM-bound: reads the raw bytes twice (count, scatter), writes them once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "unstuff_kernels.h"

#define UB 256                       /* threads per workgroup */
#define UG 4                         /* 16-byte groups per thread: 64 consecutive raw bytes */
static_assert(UB*16*UG == HJ_UNSTUFF_CHUNK, "chunk size");

namespace {

// Masks carry ONE BIT PER BYTE, at bit 7 of the byte's place in the thread's four dwords (so
// SWAR byte tests produce them directly and popcount counts bytes).
struct byte_class {
  uint32_t valid[4], drop[4], rst[4], term[4];
  uint32_t w[4];                     // the 16 bytes
  uint32_t next;                     // the byte after them (EOI past the end)
  __device__ __forceinline__ uint32_t kept() const {
    return (uint32_t)(__popc(valid[0] & ~drop[0]) + __popc(valid[1] & ~drop[1])
     + __popc(valid[2] & ~drop[2]) + __popc(valid[3] & ~drop[3]));
  }
  __device__ __forceinline__ uint32_t nrst() const {
    return (uint32_t)(__popc(rst[0]) + __popc(rst[1]) + __popc(rst[2]) + __popc(rst[3]));
  }
};
#define UB_BIT(m, i) (((m)[(i) >> 2] >> (8*((i) & 3) + 7)) & 1u)     /* i: compile-time */

#define UB_K7 0x7F7F7F7Fu
// bit 7 of every byte of x that is zero (exact, no carries between bytes)
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) { return ~(((x & UB_K7) + UB_K7) | x | UB_K7); }

// Classify the 16 bytes at p0 (a multiple of 16).  Bytes at or beyond `limit` are not valid
// (not counted, not copied); what follows a byte is judged against `avail`, the real end of the
// file, whatever the limit.
// `v` = the 16 bytes, `prev` / `next` = the bytes either side of them as they lie in the file
// (the caller has them in registers or fetches them).
__device__ __forceinline__ void classify(const uint4 v, uint32_t prev, uint32_t next, uint32_t avail,
 uint32_t limit, uint32_t p0, byte_class &c) {
#pragma unroll
  for (int k = 0; k < 4; k++) c.valid[k] = c.drop[k] = c.rst[k] = c.term[k] = c.w[k] = 0;
  c.next = 0xD9u;
  if (p0 >= limit || p0 >= avail) return;
  c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
  if (p0 + 16 < avail) c.next = next;
  if (p0 + 16 <= limit && p0 + 16 <= avail) {
    // all sixteen bytes count: four dwords at a time
    uint32_t F[5], Z[5], R[5];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      F[k] = zero_bytes(~c.w[k]);                              // == FF
      Z[k] = zero_bytes(c.w[k]);                               // == 00
      R[k] = zero_bytes((c.w[k] ^ 0xD0D0D0D0u) & 0xF8F8F8F8u);  // D0..D7
    }
    F[4] = c.next == 0xFFu ? 0x80u : 0u;
    Z[4] = c.next == 0u ? 0x80u : 0u;
    R[4] = (c.next & 0xF8u) == 0xD0u ? 0x80u : 0u;
    uint32_t carry = prev == 0xFFu ? 0x80u : 0u;               // "the byte before is FF", shifted in
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t Fn = __builtin_amdgcn_alignbit(F[k + 1], F[k], 8);
      const uint32_t Zn = __builtin_amdgcn_alignbit(Z[k + 1], Z[k], 8);
      const uint32_t Rn = __builtin_amdgcn_alignbit(R[k + 1], R[k], 8);
      const uint32_t Fp = (F[k] << 8) | carry;
      carry = F[k] >> 24;
      c.valid[k] = 0x80808080u;
      c.drop[k] = (F[k] & ~Zn) | (Fp & ~F[k]);
      c.rst[k] = F[k] & Rn;
      c.term[k] = F[k] & ~(Zn | Fn | Rn);
    }
    return;
  }
  // the ragged end of the file (or of the scan): byte by byte
  uint32_t pb = prev;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const uint32_t b = (c.w[i >> 2] >> (8*(i & 3))) & 255u;
    uint32_t nb = i < 15 ? (c.w[(i + 1) >> 2] >> (8*((i + 1) & 3))) & 255u : c.next;
    if (p0 + (uint32_t)i + 1u >= avail) nb = 0xD9u;
    const bool in = p0 + (uint32_t)i < limit && p0 + (uint32_t)i < avail;
    const bool lead = b == 0xFFu;
    const bool isrst = lead && (nb & 0xF8u) == 0xD0u;
    const bool drop = (lead && nb != 0u) || (pb == 0xFFu && !lead);
    const bool term = lead && nb != 0u && nb != 0xFFu && !isrst;
    const uint32_t bit = 0x80u << (8*(i & 3));
    if (in) {
      c.valid[i >> 2] |= bit;
      if (drop) c.drop[i >> 2] |= bit;
      if (isrst) c.rst[i >> 2] |= bit;
      if (term) c.term[i >> 2] |= bit;
    }
    pb = b;
  }
}

// First byte (0..15) flagged in a mask, or 16.
__device__ __forceinline__ uint32_t first_of(const uint32_t (&m)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) if (m[k]) return 4u*k + ((uint32_t)__builtin_ctz(m[k]) >> 3);
  return 16u;
}

// Exclusive prefix and total of two counters over the workgroup.
__device__ __forceinline__ void block_scan2(uint32_t a, uint32_t b, uint32_t &ea, uint32_t &eb,
 uint32_t &ta, uint32_t &tb, uint32_t (*wt)[2]) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t x = (uint32_t)__shfl_up((int)ia, d), y = (uint32_t)__shfl_up((int)ib, d);
    if ((int)lane >= d) { ia += x; ib += y; }
  }
  if (lane == 63u) { wt[wave][0] = ia; wt[wave][1] = ib; }
  __syncthreads();
  ea = ia - a; eb = ib - b; ta = 0; tb = 0;
#pragma unroll
  for (uint32_t w = 0; w < UB/64; w++) {
    const uint32_t x = wt[w][0], y = wt[w][1];
    if (w < wave) { ea += x; eb += y; }
    ta += x; tb += y;
  }
  __syncthreads();
}

}  // namespace

// A thread's 64 bytes: four groups, with the bytes either side of each.
struct thread_bytes {
  uint4 v[UG];
  uint32_t prev[UG], next[UG];
  __device__ __forceinline__ void load(const uint8_t *raw, uint32_t avail, uint32_t pt) {
    const uint4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < UG; j++) {
      v[j] = pt + 16u*j < avail ? *reinterpret_cast<const uint4 *>(raw + pt + 16u*j) : zero;   // (padded to 16)
    }
    prev[0] = pt && pt <= avail ? raw[pt - 1] : 0u;
    next[UG - 1] = pt + 16u*UG < avail ? raw[pt + 16u*UG] : 0xD9u;
#pragma unroll
    for (int j = 1; j < UG; j++) {
      prev[j] = v[j - 1].w >> 24;
      next[j - 1] = v[j].x & 255u;
    }
  }
};

// 1: kept bytes and RSTn markers of every chunk; position of the first non-RST marker.
__global__ __launch_bounds__(UB) void hj_unstuff_count(const hj_unstuff_args A) {
  __shared__ uint32_t wt[UB/64][2];
  const hj_unstuff_image u = A.uimg[blockIdx.y];
  if (blockIdx.x >= u.nchunks) return;
  const uint32_t pt = blockIdx.x*HJ_UNSTUFF_CHUNK + threadIdx.x*(16u*UG);
  thread_bytes tb;
  tb.load(A.raw + u.raw_off, u.avail, pt);
  uint32_t kept = 0, nrst = 0, ft = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < UG; j++) {
    byte_class c;
    classify(tb.v[j], tb.prev[j], tb.next[j], u.avail, u.avail, pt + 16u*j, c);
    kept += c.kept();
    nrst += c.nrst();
    const uint32_t f = first_of(c.term);
    if (f < 16u && ft == 0xFFFFFFFFu) ft = pt + 16u*j + f;
  }
  if (ft != 0xFFFFFFFFu) atomicMin(&A.info[blockIdx.y].hard_end, ft);
  uint32_t ek, er, tk, tr;
  block_scan2(kept, nrst, ek, er, tk, tr, wt);
  if (threadIdx.x == 0) {
    A.part[2*(size_t)(u.chunk0 + blockIdx.x)] = tk;
    A.part[2*(size_t)(u.chunk0 + blockIdx.x) + 1] = tr;
  }
}

// 2: one workgroup per image: exclusive prefixes over its chunks, and where the scan ends.
__global__ __launch_bounds__(UB) void hj_unstuff_resolve(const hj_unstuff_args A) {
  __shared__ uint32_t wt[UB/64][2];
  __shared__ uint32_t s_cstar, s_rbase, s_pos;
  const hj_unstuff_image u = A.uimg[blockIdx.x];
  const uint32_t limit = u.nseg - 1u;                        // rank of the RSTn that would be one too many
  uint32_t ck = 0, cr = 0;
  if (threadIdx.x == 0) { s_cstar = 0xFFFFFFFFu; s_pos = 0xFFFFFFFFu; s_rbase = 0; }
  __syncthreads();
  for (uint32_t base = 0; base < u.nchunks; base += UB) {
    const uint32_t c = base + threadIdx.x;
    uint32_t k = 0, r = 0;
    if (c < u.nchunks) { k = A.part[2*(size_t)(u.chunk0 + c)]; r = A.part[2*(size_t)(u.chunk0 + c) + 1]; }
    uint32_t ek, er, tk, tr;
    block_scan2(k, r, ek, er, tk, tr, wt);
    if (c < u.nchunks) {
      A.part[2*(size_t)(u.chunk0 + c)] = ck + ek;
      A.part[2*(size_t)(u.chunk0 + c) + 1] = cr + er;
      if (r && cr + er <= limit && limit < cr + er + r) { s_cstar = c; s_rbase = cr + er; }
    }
    ck += tk; cr += tr;
  }
  __syncthreads();
  const uint32_t cstar = s_cstar;
  if (cstar != 0xFFFFFFFFu) {                                // look inside that chunk for RSTn number `limit`
    const uint32_t pt = cstar*HJ_UNSTUFF_CHUNK + threadIdx.x*(16u*UG);
    thread_bytes tb;
    tb.load(A.raw + u.raw_off, u.avail, pt);
    byte_class c[UG];
    uint32_t nr = 0;
#pragma unroll
    for (int j = 0; j < UG; j++) {
      classify(tb.v[j], tb.prev[j], tb.next[j], u.avail, u.avail, pt + 16u*j, c[j]);
      nr += c[j].nrst();
    }
    uint32_t e0, er, t0, tr;
    block_scan2(0u, nr, e0, er, t0, tr, wt);
    uint32_t rank = s_rbase + er;
#pragma unroll
    for (int j = 0; j < UG; j++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (UB_BIT(c[j].rst, i) && rank++ == limit) s_pos = pt + 16u*j + (uint32_t)i;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hj_unstuff_info *I = A.info + blockIdx.x;
    uint32_t end = I->hard_end < u.avail ? I->hard_end : u.avail;
    if (s_pos < end) end = s_pos;
    I->end = end;
    I->found = 0;
    I->scan_len = 0;
  }
}

// 3: copy the kept bytes before the end to their clean positions; note where each RSTn falls.
__global__ __launch_bounds__(UB) void hj_unstuff_scatter(const hj_unstuff_args A) {
  __shared__ uint32_t wt[UB/64][2];
  __shared__ __attribute__((aligned(16))) uint32_t stage32[HJ_UNSTUFF_CHUNK/4 + 4];
  const hj_unstuff_image u = A.uimg[blockIdx.y];
  if (blockIdx.x >= u.nchunks) return;
  hj_unstuff_info *I = A.info + blockIdx.y;
  const uint32_t end = I->end;
  const uint32_t c0 = blockIdx.x*HJ_UNSTUFF_CHUNK;
  if (c0 >= end) return;
  const uint32_t pt = c0 + threadIdx.x*(16u*UG);
  thread_bytes tb;
  tb.load(A.raw + u.raw_off, u.avail, pt);
  byte_class c[UG];
  uint32_t cnt = 0, nr = 0;
#pragma unroll
  for (int j = 0; j < UG; j++) {
    classify(tb.v[j], tb.prev[j], tb.next[j], u.avail, end, pt + 16u*j, c[j]);
#pragma unroll
    for (int k = 0; k < 4; k++) c[j].valid[k] &= ~c[j].drop[k];     // valid := kept
    cnt += (uint32_t)(__popc(c[j].valid[0]) + __popc(c[j].valid[1]) + __popc(c[j].valid[2]) + __popc(c[j].valid[3]));
    nr += c[j].nrst();
  }
  uint32_t ek, er, tk, tr;
  block_scan2(cnt, nr, ek, er, tk, tr, wt);
  const uint32_t kbase = A.part[2*(size_t)(u.chunk0 + blockIdx.x)];      // clean bytes before this chunk
  const uint32_t rbase = A.part[2*(size_t)(u.chunk0 + blockIdx.x) + 1];  // RSTn markers before it
  const hj_image *im = A.images + blockIdx.y;
  uint8_t *dst = A.clean + im->scan_off + kbase;
  const uint32_t s0 = (uint32_t)((uintptr_t)dst & 3u);
  uint8_t *stage = reinterpret_cast<uint8_t *>(stage32);
  {
    uint32_t n = s0 + ek;
#pragma unroll
    for (int j = 0; j < UG; j++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (UB_BIT(c[j].valid, i)) stage[n++] = (uint8_t)(c[j].w[i >> 2] >> (8*(i & 3)));
      }
    }
  }
  if (nr) {
    const uint32_t limit = u.nseg - 1u;
    uint32_t rank = rbase + er, before = 0;                  // kept bytes of this thread before the byte
#pragma unroll
    for (int j = 0; j < UG; j++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (UB_BIT(c[j].rst, i)) {
          if (rank < limit) {
            A.bnd[im->seg0 + rank] = kbase + ek + before;
            const uint32_t code = i < 15 ? (c[j].w[(i + 1) >> 2] >> (8*((i + 1) & 3))) & 255u : c[j].next;
            if (code != 0xD0u + (rank & 7u)) atomicOr(&A.errors[blockIdx.y], 1u);    // "invalid RST counter"
            atomicMax(&I->found, rank + 1u);
          }
          rank++;
        }
        before += UB_BIT(c[j].valid, i);
      }
    }
  }
  if (threadIdx.x == 0 && tk) atomicMax(&I->scan_len, kbase + tk);   // (one atomic per chunk)
  __syncthreads();
  // the chunk's tk clean bytes sit at stage[s0 ..): ragged ends as bytes, the middle as dwords
  const uint32_t head = tk < ((4u - s0) & 3u) ? tk : ((4u - s0) & 3u);
  if (threadIdx.x < head) dst[threadIdx.x] = stage[s0 + threadIdx.x];
  const uint32_t mid = (tk - head) >> 2;
  {
    uint32_t *g32 = reinterpret_cast<uint32_t *>(dst + head);
    const uint32_t *s32 = stage32 + ((s0 + head) >> 2);
    for (uint32_t d = threadIdx.x; d < mid; d += UB) g32[d] = s32[d];
  }
  const uint32_t tail = tk - head - 4u*mid;
  if (threadIdx.x < tail) dst[head + 4u*mid + threadIdx.x] = stage[s0 + head + 4u*mid + threadIdx.x];
}

// 4: one workgroup per image: the segment table hj_prepare_scan would have written, the
// image's clean length and subsequence count, the 16 pad bytes behind the stream.
__global__ __launch_bounds__(UB) void hj_unstuff_segments(const hj_unstuff_args A) {
  __shared__ uint32_t wt[UB/64][2];
  const hj_unstuff_image u = A.uimg[blockIdx.x];
  hj_image *im = A.images + blockIdx.x;
  const hj_unstuff_info I = A.info[blockIdx.x];
  const uint32_t found = I.found < u.nseg - 1u ? I.found : u.nseg - 1u;
  const uint32_t scan_len = I.scan_len;
  const uint32_t seg0 = im->seg0;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < u.nseg; base += UB) {
    const uint32_t k = base + threadIdx.x;
    uint32_t start = 0, endp = 0, nsub = 0;
    if (k < u.nseg) {
      start = k == 0 ? 0u : (k - 1 < found ? A.bnd[seg0 + k - 1] : scan_len);
      endp = k < found ? A.bnd[seg0 + k] : scan_len;
      if (endp < start) endp = start;
      nsub = (endp - start + (1u << A.sub_log2) - 1u) >> A.sub_log2;
      if (nsub == 0) nsub = 1;
    }
    uint32_t es, e1, ts, t1;
    block_scan2(nsub, 0u, es, e1, ts, t1, wt);
    if (k < u.nseg) {
      hj_segment s;
      s.start = start; s.end = endp; s.sub0 = carry + es; s.nsub = nsub;
      s.mcu0 = u.ri ? k*u.ri : 0u;
      s.nmcu = u.ri ? (u.total_mcus - s.mcu0 < u.ri ? u.total_mcus - s.mcu0 : u.ri) : u.total_mcus;
      A.segs[seg0 + k] = s;
    }
    carry += ts;
  }
  if (threadIdx.x == 0) {
    im->nsub = carry;
    im->scan_len = scan_len;
    if (I.found != u.nseg - 1u) atomicOr(&A.errors[blockIdx.x], 1u);        // "entropy data ended early"
  }
  if (threadIdx.x < 16) A.clean[im->scan_off + scan_len + threadIdx.x] = 0xFF;
}

extern "C" int hj_launch_unstuff(const hj_unstuff_args *A, int max_chunks, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 chunks(max_chunks, A->nimages);
  hipLaunchKernelGGL(hj_unstuff_count, chunks, dim3(UB), 0, st, *A);
  hipLaunchKernelGGL(hj_unstuff_resolve, dim3(A->nimages), dim3(UB), 0, st, *A);
  hipLaunchKernelGGL(hj_unstuff_scatter, chunks, dim3(UB), 0, st, *A);
  hipLaunchKernelGGL(hj_unstuff_segments, dim3(A->nimages), dim3(UB), 0, st, *A);
  return (int)hipGetLastError();
}
