// pack_kernels.hip — PACK wire format -> coefficient planes, on the GPU (gfx950).
//
// The reference's PACK stage ships each block as a run of 16-bit words instead of 64
// shorts (producer src/xjpeg.c:484-496, 513-519, 531-535):
//   word 0        DC level & 0xfff
//   then per AC   (run << 12) | (level & 0xfff)      ZRL = 0xF000
//   0x0000        end of block (omitted when the block ran to coefficient 63)
// plus one int per block, the index of its word 0 (`plane->index[by*hblocks + bx]`,
// raster per plane, planes back to back as src/image.c:86-95 lays them out).  The
// reference consumes it in its first GLSL pass (res/horz_pack_yuv.fs.glsl:94-127):
// zero the block, sign-extend 12-bit levels, `j += run + 1`, de-zigzag.
//
// jga_unpack_kernel does that expansion for a batch of same-geometry images and
// writes the QUANT-stage planes (Appendix-B layout) the IDCT kernels read, so only
// the compact form crosses PCIe.
//
// One WAVEFRONT expands one block at a time, one word per lane — a block is at most
// 1 + 63 words, exactly a wave64:
//   * where the wave's 64 blocks start and where they go is worked out up front, lane l
//     for block l (one gather of the 64 start indices), and handed to the per-block loop
//     as wave-uniform scalars with v_readlane;
//   * lane l loads word start+l — a coalesced 128-byte read;
//   * the first zero word (ballot + count-trailing-zeros) ends the block;
//   * zig-zag position of word l = inclusive prefix sum of (run+1) over the lanes: six
//     DPP adds (row_shr 1/2/4/8, row_bcast 15/31), no LDS;
//   * each valid lane drops its sign-extended level at its natural position in a
//     128-byte LDS line of the wave; the lines of the blocks in flight then leave 16 bytes
//     per lane, eight whole blocks per store instruction (and are zeroed for the next ones).
// Sixteen blocks are in flight per wave so that the vector load -> LDS -> store
// chain of one overlaps the others (0.68 ms with 4 in flight, 0.54 ms with 16).  Blocks
// are taken in scan order (the order the producer emits the words), so consecutive reads
// walk the word stream sequentially.
// Integer/byte work, HBM-bound: per block it reads 2 B x words + 4 B and writes 128 B.
//
// Out-of-range input is made safe, not meaningful: word reads stop at the end of the
// image's PACK buffer, a run past coefficient 63 ends the block (the reference indexes
// out of bounds there).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pack_params.h"

#define PK_BLOCK 256                 /* 4 waves */
#define PK_INFLIGHT 16                /* blocks a wave works on at once */
#define PK_PER_WAVE 64               /* consecutive blocks (scan order) per wave: one per lane */

__device__ const uint8_t PK_DEZZ[64] = {     // T.81 Figure A.6: zig-zag index -> natural index
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
  13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59,
  52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef int16_t __attribute__((may_alias)) pk_i16_alias;
typedef uint32_t __attribute__((may_alias)) pk_v4u __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ int pk_sext12(uint32_t w) {
  return (int)(w << 20) >> 20;       // horz_pack_yuv.fs.glsl:112, 123
}

// x + (x of the lane `ctrl` positions away), DPP; lanes without a source add 0
#define PK_DPP_ADD(x, ctrl, rows) \
  ((x) + __builtin_amdgcn_update_dpp(0, (x), (ctrl), (rows), 0xf, false))

// Inclusive prefix sum over the 64 lanes of a wave.
static __device__ __forceinline__ int pk_wave_scan(int x) {
  x = PK_DPP_ADD(x, 0x111, 0xf);     // row_shr:1
  x = PK_DPP_ADD(x, 0x112, 0xf);     // row_shr:2
  x = PK_DPP_ADD(x, 0x114, 0xf);     // row_shr:4
  x = PK_DPP_ADD(x, 0x118, 0xf);     // row_shr:8   -> scan inside each row of 16
  x = PK_DPP_ADD(x, 0x142, 0xa);     // row_bcast:15 into rows 1 and 3
  x = PK_DPP_ADD(x, 0x143, 0xc);     // row_bcast:31 into rows 2 and 3
  return x;
}

__global__ __launch_bounds__(PK_BLOCK) void jga_unpack_kernel(const jga_pack_params P) {
  __shared__ __attribute__((aligned(16))) uint16_t lds_line[PK_BLOCK/64][PK_INFLIGHT][64];   // one 128-byte line per block in flight
  __shared__ uint8_t s_dezz[64];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int img = blockIdx.y;
  if (threadIdx.x < 64) s_dezz[threadIdx.x] = PK_DEZZ[threadIdx.x];
#pragma unroll
  for (int u = 0; u < PK_INFLIGHT; u++) lds_line[wv][u][lane] = 0;
  __syncthreads();

  const uint32_t nslots = (uint32_t)P.nslots, nhmb = (uint32_t)P.nhmb;
  const uint32_t total = nhmb*(uint32_t)P.nvmb*nslots;            // blocks of the scan
  const uint32_t first = (blockIdx.x*(PK_BLOCK/64) + wv)*PK_PER_WAVE;
  const int32_t *index = P.index + (long long)img*P.index_stride;
  const uint16_t *pack = P.pack + (long long)img*P.pack_stride;
  int16_t *coef = P.coef + (long long)img*P.coef_stride;
  const uint32_t limit = (uint32_t)P.pack_words;
  const long long rs = (long long)P.w0_blocks*64;

  // Lane l works out, once, where block first+l starts in the stream and where it goes in
  // the planes (all 64 descriptions in parallel, the 64 start indices in one gather); the
  // loop below picks them up with v_readlane.
  uint32_t my_start = limit;
  unsigned long long my_dst = 0;
  {
    const uint32_t b = first + lane;
    if (b < total) {
      const uint32_t mcu = (uint32_t)(((uint64_t)b*P.div_nslots.mul) >> P.div_nslots.shift);
      const uint32_t slot = b - mcu*nslots;
      const uint32_t mby = (uint32_t)(((uint64_t)mcu*P.div_nhmb.mul) >> P.div_nhmb.shift);
      const uint32_t mbx = mcu - mby*nhmb;
      // slot -> plane / position inside the MCU: 6 bits per slot, ten slots per 64-bit word, and
      // three-way selects instead of indexed kernel arguments (those would be memory loads)
      const uint32_t sd = (uint32_t)(slot < 10u ? P.slot_desc[0] >> (6u*slot) : P.slot_desc[1] >> (6u*(slot - 10u))) & 63u;
      const uint32_t pl = sd & 3u, sbx = (sd >> 2) & 3u, sby = sd >> 4;
#define PK_SEL(a) (pl == 0u ? (a)[0] : pl == 1u ? (a)[1] : (a)[2])
      const uint32_t bx = mbx*(uint32_t)PK_SEL(P.plane_hs) + sbx;
      const uint32_t by = mby*(uint32_t)PK_SEL(P.plane_vs) + sby;
      const int xdec = PK_SEL(P.plane_xdec);
      const uint32_t hblocks = (uint32_t)PK_SEL(P.plane_hblocks);
      const uint32_t index0 = (uint32_t)PK_SEL(P.plane_index0);
      my_dst = (unsigned long long)(uintptr_t)(coef + PK_SEL(P.plane_coef_off) + rs*(by >> xdec)
       + (rs >> xdec)*(by & ((1u << xdec) - 1u)) + (long long)bx*64);
#undef PK_SEL
      my_start = (uint32_t)index[index0 + by*hblocks + bx];
    }
  }
  const uint32_t dst_lo = (uint32_t)my_dst, dst_hi = (uint32_t)(my_dst >> 32);

#pragma unroll 1
  for (uint32_t b0 = 0; b0 < PK_PER_WAVE && first + b0 < total; b0 += PK_INFLIGHT) {
    uint32_t start[PK_INFLIGHT];
    int16_t *dst[PK_INFLIGHT];
    uint32_t w[PK_INFLIGHT];
#pragma unroll
    for (int u = 0; u < PK_INFLIGHT; u++) {
      const int src = (int)b0 + u;                       // wave-uniform lane number
      start[u] = (uint32_t)__builtin_amdgcn_readlane((int)my_start, src);
      const unsigned long long a = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dst_hi, src) << 32)
       | (uint32_t)__builtin_amdgcn_readlane((int)dst_lo, src);
      dst[u] = reinterpret_cast<int16_t *>((uintptr_t)a);
    }
    // one word per lane
#pragma unroll
    for (int u = 0; u < PK_INFLIGHT; u++) {
      const uint32_t k = start[u] + lane;
      w[u] = (start[u] < limit && k < limit) ? pack[k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < PK_INFLIGHT; u++) {
      // words up to the first zero after the DC word belong to the block.  The terminator needs
      // no vote: a zero word adds 64 to the prefix sum, so its own position and every later one
      // fail the "position < 64" test that ends a block anyway
      const int pos = pk_wave_scan(lane == 0u ? 0 : w[u] == 0u ? 64 : (int)(w[u] >> 12) + 1);
      if (start[u] < limit && pos < 64) {
        lds_line[wv][u][s_dezz[pos]] = (uint16_t)pk_sext12(w[u]);
      }
    }
    // (a wave's LDS accesses execute in order: the writes above are visible to the reads below)
    // Write-out: the PK_INFLIGHT lines leave 16 bytes per lane — lanes 8j..8j+7 carry block j of
    // a pass — so one store instruction moves eight whole blocks instead of one.
    static_assert(PK_INFLIGHT % 8 == 0, "eight blocks per store pass");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int pass = 0; pass < PK_INFLIGHT/8; pass++) {
      const int u = pass*8 + (int)(lane >> 3), part = (int)(lane & 7u);
      pk_v4u *line = reinterpret_cast<pk_v4u *>(&lds_line[wv][u][0]) + part;
      const pk_v4u v = *line;
      const pk_v4u zero = {0u, 0u, 0u, 0u};
      *line = zero;
      const int owner = (int)b0 + u;                        // lane that described this block
      const uint32_t lo = (uint32_t)__shfl((int)dst_lo, owner), hi = (uint32_t)__shfl((int)dst_hi, owner);
      const unsigned long long a = ((unsigned long long)hi << 32) | lo;
      if (a && owner < PK_PER_WAVE) {
        typedef __attribute__((address_space(1))) pk_v4u global_v4u;     // global_store, not flat
        __builtin_nontemporal_store(v, (global_v4u *)(uintptr_t)a + part);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int jga_launch_unpack(const jga_pack_params *P, void *stream) {
  const int blocks = P->nhmb*P->nvmb*P->nslots, per_group = (PK_BLOCK/64)*PK_PER_WAVE;
  dim3 grid((blocks + per_group - 1)/per_group, P->nimages), block(PK_BLOCK);
  hipLaunchKernelGGL(jga_unpack_kernel, grid, block, 0, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
