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
// One LANE expands one block (round 3; until then a wavefront expanded one block at a time, one
// word per lane, with the zig-zag positions from a DPP prefix sum — 25 of 64 lanes busy and ~23
// wave instructions per block: the kernel was bound by instruction issue at 0.46-0.49 of the HBM
// peak):
//   * lane l of a workgroup takes block (256 x group + l) of the scan (the order the producer
//     emits the words in, so neighbouring lanes read neighbouring stretches of the stream), works
//     out where it starts and where it goes, and walks its words itself: sixteen words per trip
//     (four 8-byte loads, 2-byte aligned), `position += run + 1`, level sign-extended into the lane's own
//     128-byte LDS buffer at the natural position (33-dword stride: lanes hit different banks);
//   * a wave's loop runs as long as its longest block (<= 64 words);
//   * then the wave's 64 buffers leave as whole 128-byte lines, eight lanes per block, 16 bytes
//     each: eight blocks per store instruction.
// A third fewer instructions per block than the wave-per-block form (the lanes of short blocks
// still sit out the trips of the longest one): [MI355X] 0.449 -> 0.349 ms per 48 x 4K = 0.61 of the
// HBM peak (tools/ubench.py).  Round 6: a trip runs as three sweeps — positions, look-ups, stores — instead of a
// branch + look-up + wait + store per word: 0.355 -> 0.342 ms = 0.62 (probes and what else was tried:
// profiles/r6_short_runs.md 5).
// Integer/byte work, HBM-bound: per block it reads 2 B x words + 4 B and writes 128 B.
//
// Out-of-range input is made safe, not meaningful: word reads stop at the end of the
// image's PACK buffer, a run past coefficient 63 ends the block (the reference indexes
// out of bounds there).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pack_params.h"

#define PK_BLOCK 256                 /* 4 waves, one block per lane */
#ifndef PK_HOMOGENEOUS
#define PK_HOMOGENEOUS 0             /* 1: a wave takes one MCU slot of 64 consecutive MCUs; 0: 64 blocks in scan order */
#endif
#ifndef PK_BLK_STRIDE
#define PK_BLK_STRIDE 33             /* dwords per lane's block buffer: 32 + its destination (36: 16-byte aligned buffers, an A/B) */
#endif
#ifndef PK_CHUNK
#define PK_CHUNK 16                  /* words per trip: 8 or 16 */
#endif
#ifndef PK_PROBE
#define PK_PROBE 0                   /* 1-3: timing probes with wrong output (tools/build_variant.sh), never in the product */
#endif
#ifndef PK_PREFETCH
#define PK_PREFETCH 0                /* 1: the next trip's words are loaded before this trip's are placed */
#endif
// [MI355X] 48 x 4K, 25.4 words per block: 8 / 16 / 32 / 64 words per trip = 0.365 / 0.349 / 0.400 / 0.521 ms;
// with the prefetch 0.395 / 0.414 (the registers it holds cost more than the latency it hides)

__device__ const uint8_t PK_DEZZ[64] = {     // T.81 Figure A.6: zig-zag index -> natural index
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
  13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59,
  52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef int16_t __attribute__((may_alias)) pk_i16_alias;
typedef uint32_t __attribute__((may_alias)) pk_v4u __attribute__((ext_vector_type(4)));
typedef uint64_t __attribute__((aligned(2), may_alias)) pk_u64_a2;

static __device__ __forceinline__ int pk_sext12(uint32_t w) {
  return (int)(w << 20) >> 20;       // horz_pack_yuv.fs.glsl:112, 123
}

__global__ __launch_bounds__(PK_BLOCK) void jga_unpack_kernel(const jga_pack_params P) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_blk[PK_BLOCK*PK_BLK_STRIDE];
  __shared__ uint8_t s_dezz2[68];                          // BYTE offset of zig-zag position p in a block buffer; [64]: the spare dword
  const uint32_t lane = threadIdx.x & 63;
  const int img = blockIdx.y;
  if (threadIdx.x < 64) s_dezz2[threadIdx.x] = (uint8_t)(2u*PK_DEZZ[threadIdx.x]);
  if (threadIdx.x == 64) s_dezz2[64] = 128;                 // (words of a finished block land in dword 32, which receives the block's destination afterwards)
  uint32_t *blk = lds_blk + threadIdx.x*PK_BLK_STRIDE;
#if PK_BLK_STRIDE % 4 == 0
  {
    const pk_v4u zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 8; q++) reinterpret_cast<pk_v4u *>(blk)[q] = zero4;   // (ds_write_b128)
  }
#else
#pragma unroll
  for (int q = 0; q < 32; q++) blk[q] = 0;                  // own buffer only
#endif
  __syncthreads();                                           // s_dezz2

  const uint32_t nslots = (uint32_t)P.nslots, nhmb = (uint32_t)P.nhmb;
  const uint32_t total = nhmb*(uint32_t)P.nvmb*nslots;            // blocks of the scan
  const int32_t *index = P.index + (long long)img*P.index_stride;
  const uint16_t *pack = P.pack + (long long)img*P.pack_stride;
  const uint32_t limit = (uint32_t)P.pack_words;
  const long long rs = (long long)P.w0_blocks*64;

  // where this lane's block starts in the stream and where it goes in the planes
  uint32_t k = limit, dst = ~0u;                             // dst: 128-byte slot of the image's planes; ~0: no block
  {
#if PK_HOMOGENEOUS
    // A/B (round 4, NOT the default): a wave's trips run as long as its LONGEST block, and luma blocks are
    // systematically longer than chroma ones (the bench's files: 33.8 words against 8.4) — with the blocks dealt
    // out in scan order every wave holds both kinds and takes the luma trip count.  Here `nslots` consecutive
    // waves share 64 consecutive MCUs and wave w of them takes MCU slot w of each: a wave holds ONE kind of block
    // (modelled on the bench's files: 636 -> 495 loop instructions per 64 blocks).  [MI355X] SLOWER: 0.379-0.387 ms
    // against 0.354-0.359 (profiles/r4_pack_homogeneous_ab.txt) — neighbouring lanes' words then lie 300 bytes
    // apart instead of 50, and what the trips save the scattered reads cost twice over: the kernel is not bound by
    // its instruction count alone.
    const uint32_t wave_id = (blockIdx.x*PK_BLOCK + threadIdx.x) >> 6;
    const uint32_t sg = (uint32_t)(((uint64_t)wave_id*P.div_nslots.mul) >> P.div_nslots.shift);
    const uint32_t slot_w = wave_id - sg*nslots, mcu_w = sg*64u + lane;
    const bool mine = mcu_w < nhmb*(uint32_t)P.nvmb;
    const uint32_t b = mine ? mcu_w*nslots + slot_w : total;
#else
    const uint32_t b = blockIdx.x*PK_BLOCK + threadIdx.x;
#endif
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
      dst = (uint32_t)((PK_SEL(P.plane_coef_off) + rs*(by >> xdec)
       + (rs >> xdec)*(by & ((1u << xdec) - 1u)) + (long long)bx*64) >> 6);     // (block offsets are multiples of 64 shorts)
#undef PK_SEL
      k = (uint32_t)index[index0 + by*hblocks + bx];
    }
  }
  // Walk the block's words: word 0 is the DC level (position 0), every later one moves the
  // position by run + 1 — a zero word (end of block) by 64, which ends the walk like a run past
  // coefficient 63 does.
  uint8_t *blk8 = reinterpret_cast<uint8_t *>(blk);
  bool active = k < limit;                                   // (a start outside the stream: the block stays zero)
#if PK_PROBE == 1
  uint32_t probe_acc = 0;
#endif
  bool dcword = true;
  int pos = 0;
  // PK_CHUNK words per trip (with PK_PREFETCH the next trip's are loaded before these are placed)
  struct chunk { uint32_t q[PK_CHUNK/2]; };
  auto fetch = [&](uint32_t at, bool want) {
    chunk c;
#pragma unroll
    for (int j = 0; j < PK_CHUNK/2; j++) c.q[j] = 0u;
    if (want) {
      if (at + (uint32_t)PK_CHUNK <= limit) {
#pragma unroll
        for (int j = 0; j < PK_CHUNK/4; j++) {
          const uint64_t v = *reinterpret_cast<const pk_u64_a2 *>(pack + at + 4*j);
          c.q[2*j] = (uint32_t)v; c.q[2*j + 1] = (uint32_t)(v >> 32);
        }
      }
      else if (at < limit) {                                 // the stream's last words
#pragma unroll
        for (int j = 0; j < PK_CHUNK; j++) {
          const uint32_t w = at + (uint32_t)j < limit ? pack[at + (uint32_t)j] : 0u;
          c.q[j >> 1] |= w << (16*(j & 1));
        }
      }
    }
    return c;
  };
  chunk cur = fetch(k, active);
  while (__ballot(active) != 0ull) {
#if PK_PREFETCH
    const chunk nxt = fetch(k + (uint32_t)PK_CHUNK, active);
#endif
#if PK_PROBE
#pragma unroll
    for (int j = 0; j < PK_CHUNK; j++) {
      const uint32_t w = (j & 1) ? cur.q[j >> 1] >> 16 : cur.q[j >> 1] & 0xffffu;
      const int inc = (j == 0 && dcword) ? 0 : w == 0u ? 64 : (int)(w >> 12) + 1;
      pos += inc;
      active = active && pos < 64;
#if PK_PROBE == 1            /* timing probes only (round 6, profiles/r6_short_runs.md 5): no scatter at all: the loop's loads and arithmetic alone */
      probe_acc += active ? (uint32_t)pk_sext12(w) + (uint32_t)pos : 0u;
#elif PK_PROBE == 2          /* the scatter without its bank conflicts: every lane stores to ONE fixed halfword of its own buffer */
      if (active) *reinterpret_cast<pk_i16_alias *>(blk8 + 2*(j & 1)) = (int16_t)pk_sext12(w);
#elif PK_PROBE == 3          /* ... and without the de-zigzag look-up: the position itself as the offset (conflicts stay) */
      if (active) *reinterpret_cast<pk_i16_alias *>(blk8 + 2*pos) = (int16_t)pk_sext12(w);
#else                        /* 4: round 5's loop — per word a branch, a look-up, a wait, a store */
      if (active) *reinterpret_cast<pk_i16_alias *>(blk8 + s_dezz2[pos]) = (int16_t)pk_sext12(w);
#endif
    }
#else
    // Three sweeps over the trip's words (round 6): the positions (a serial prefix in registers), ALL the de-zigzag
    // look-ups (independent LDS reads: one latency per trip, where the branchy form — per word a test, a look-up, a
    // wait, a store — paid sixteen in a row: 0.355 -> 0.32x ms per 48 x 4K, profiles/r6_short_runs.md 5), the stores.
    // No branch: the words behind a block's end go to the spare dword.
    uint32_t at[PK_CHUNK];
#pragma unroll
    for (int j = 0; j < PK_CHUNK; j++) {
      const uint32_t w = (j & 1) ? cur.q[j >> 1] >> 16 : cur.q[j >> 1] & 0xffffu;
      const int inc = (j == 0 && dcword) ? 0 : w == 0u ? 64 : (int)(w >> 12) + 1;
      pos += inc;
      active = active && pos < 64;
      at[j] = active ? (uint32_t)pos : 64u;
    }
#pragma unroll
    for (int j = 0; j < PK_CHUNK; j++) at[j] = s_dezz2[at[j]];
#pragma unroll
    for (int j = 0; j < PK_CHUNK; j++) {
      const uint32_t w = (j & 1) ? cur.q[j >> 1] >> 16 : cur.q[j >> 1] & 0xffffu;
      *reinterpret_cast<pk_i16_alias *>(blk8 + at[j]) = (int16_t)pk_sext12(w);
    }
#endif
    dcword = false;
    k += (uint32_t)PK_CHUNK;
#if PK_PREFETCH
    cur = nxt;
#else
    cur = fetch(k, active);
#endif
  }
#if PK_PROBE == 1
  blk[0] = probe_acc;
#endif
  blk[32] = dst;                                             // (the spare dword's last use: where the block goes)
  // (a wave's LDS accesses execute in order: the writes above are visible to the reads below)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // Write-out: lanes 8j..8j+7 of a pass carry block j of it, 16 bytes each — one store instruction
  // moves eight whole blocks.
  const uint32_t *wave_blk = lds_blk + (threadIdx.x & ~63u)*PK_BLK_STRIDE;
  uint8_t *planes = reinterpret_cast<uint8_t *>(P.coef + (long long)img*P.coef_stride);
  const uint32_t part = lane & 7u;
#pragma unroll
  for (int pass = 0; pass < 8; pass++) {
    const uint32_t *src = wave_blk + ((uint32_t)pass*8u + (lane >> 3))*PK_BLK_STRIDE;
    const uint32_t off = src[32];
    pk_v4u v;
#if PK_BLK_STRIDE % 4 == 0
    v = reinterpret_cast<const pk_v4u *>(src)[part];        // (ds_read_b128)
#else
    v.x = src[4*part]; v.y = src[4*part + 1]; v.z = src[4*part + 2]; v.w = src[4*part + 3];
#endif
    if (off != ~0u) {
      typedef __attribute__((address_space(1))) pk_v4u global_v4u;     // global_store, not flat
      __builtin_nontemporal_store(v, (global_v4u *)(uintptr_t)(planes + (size_t)off*128u) + part);
    }
  }
}

extern "C" int jga_launch_unpack(const jga_pack_params *P, void *stream) {
#if PK_HOMOGENEOUS
  const int waves = ((P->nhmb*P->nvmb + 63)/64)*P->nslots;      // nslots waves per 64 MCUs
  const int blocks = waves*64;
#else
  const int blocks = P->nhmb*P->nvmb*P->nslots;
#endif
  dim3 grid((blocks + PK_BLOCK - 1)/PK_BLOCK, P->nimages), block(PK_BLOCK);
  hipLaunchKernelGGL(jga_unpack_kernel, grid, block, 0, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
