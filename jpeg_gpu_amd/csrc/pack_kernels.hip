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
// the compact form crosses PCIe.  One lane owns one block: it walks its words
// (dword loads, two words each) and drops the levels into a private 32-dword LDS
// buffer (33-dword stride: conflict-free); the wave then writes its 64 blocks out
// together, 16 bytes per lane per store, so that neighbouring blocks — which are
// neighbours in the plane layout — leave as full 1 KB bursts.  Integer/byte work,
// HBM-bound: per block it reads 2 B x words + 4 B and writes 128 B.
//
// Out-of-range input is made safe, not meaningful: word reads stop at the end of the
// image's PACK buffer, a run past coefficient 63 ends the block (the reference indexes
// out of bounds there).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pack_params.h"

#define PK_BLOCK 256
#define PK_STRIDE 33

__device__ const uint8_t PK_DEZZ[64] = {     // T.81 Figure A.6: zig-zag index -> natural index
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
  13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59,
  52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef int16_t __attribute__((may_alias)) pk_i16_alias;
typedef uint32_t pk_v4u __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ int pk_sext12(uint32_t w) {
  return (int)(w << 20) >> 20;       // horz_pack_yuv.fs.glsl:112, 123
}

__global__ __launch_bounds__(PK_BLOCK) void jga_unpack_kernel(const jga_pack_params P) {
  __shared__ uint32_t lds_blk[PK_BLOCK*PK_STRIDE];
  __shared__ uint8_t s_dezz[64];
  const uint32_t t = threadIdx.x, lane = t & 63;
  const int img = blockIdx.y;
  if (t < 64) s_dezz[t] = PK_DEZZ[t];
  uint32_t *blk = lds_blk + t*PK_STRIDE;
#pragma unroll
  for (int k = 0; k < 32; k++) blk[k] = 0;
  __syncthreads();

  // flat block number -> (plane, by, bx)
  const uint32_t f = blockIdx.x*PK_BLOCK + t;
  const bool valid = f < (uint32_t)P.plane_first[P.nplanes];
  int16_t *dst = nullptr;
  if (valid) {
    int pl = 0;
    if (P.nplanes > 1 && f >= (uint32_t)P.plane_first[1]) pl = 1;
    if (P.nplanes > 2 && f >= (uint32_t)P.plane_first[2]) pl = 2;
    const uint32_t local = f - (uint32_t)P.plane_first[pl];
    const uint32_t hb = (uint32_t)P.plane_hblocks[pl];
    const uint32_t by = local/hb, bx = local - by*hb;
    const int xdec = P.plane_xdec[pl];
    const long long rs = (long long)P.w0_blocks*64;
    dst = P.coef + (long long)img*P.coef_stride + P.plane_coef_off[pl]
     + rs*(by >> xdec) + (rs >> xdec)*(by & ((1u << xdec) - 1u)) + (long long)bx*64;
    // walk the words of this block
    const uint32_t *pw = reinterpret_cast<const uint32_t *>(P.pack + (long long)img*P.pack_stride);
    const uint32_t limit = (uint32_t)P.pack_words;
    uint32_t i = (uint32_t)P.index[(long long)img*P.index_stride + P.plane_index0[pl] + local];
    pk_i16_alias *b16 = reinterpret_cast<pk_i16_alias *>(blk);
    if (i < limit) {
      uint32_t two = pw[i >> 1];
      uint32_t w = (i & 1) ? two >> 16 : two & 0xffffu;
      b16[0] = (int16_t)pk_sext12(w);
      i++;
      int j = 0;
      while (j < 63 && i < limit) {
        if (!(i & 1)) two = pw[i >> 1];
        w = (i & 1) ? two >> 16 : two & 0xffffu;
        i++;
        if (w == 0) break;
        j += (int)(w >> 12) + 1;
        if (j > 63) break;
        b16[s_dezz[j]] = (int16_t)pk_sext12(w);
      }
    }
  }
  __syncthreads();                   // every lane's block is complete in LDS
  // Wave-cooperative write-out: 16-byte piece q of the wave's 64 blocks = part (q & 7)
  // of block (q >> 3); its address comes from the owning lane.
  const uint32_t wave_base = t & ~63u;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint32_t q = lane + 64u*(uint32_t)r;
    const uint32_t owner = q >> 3, part = q & 7u;
    const unsigned long long a = (unsigned long long)(uintptr_t)dst;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)a, (int)owner);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(a >> 32), (int)owner);
    int16_t *d = reinterpret_cast<int16_t *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    const uint32_t *s = lds_blk + (wave_base + owner)*PK_STRIDE + part*4;
    pk_v4u v;
    v.x = s[0]; v.y = s[1]; v.z = s[2]; v.w = s[3];
    if (d) __builtin_nontemporal_store(v, reinterpret_cast<pk_v4u *>(d) + part);
  }
}

extern "C" int jga_launch_unpack(const jga_pack_params *P, void *stream) {
  const int nblocks = P->plane_first[P->nplanes];
  dim3 grid((nblocks + PK_BLOCK - 1)/PK_BLOCK, P->nimages), block(PK_BLOCK);
  hipLaunchKernelGGL(jga_unpack_kernel, grid, block, 0, (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}
