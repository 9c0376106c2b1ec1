// idct_kernels.hip — the JPEG block-decode hot path as CDNA4 (gfx950) kernels.
//
// Replaces the reference's three fragment-shader passes and the CPU loop that
// defines their arithmetic:
//   dequantise ........ src/xjpeg.c:501-503, 524-527   (res/horz_quant_yuv.fs.glsl:81-99)
//   row + column IDCT .. src/dct.c:21-87, 100-121       (res/horz.fs.glsl, res/vert.fs.glsl)
//   +128, clamp, store . src/xjpeg.c:565-584
//   upsample + RGB ..... res/unyuv.fs.glsl:17-50, res/ungrey.fs.glsl
// in ONE launch with no intermediate HBM traffic (the reference writes 256 B +
// 128 B per block between its passes).
//
// Bit-exactness rules: binary32, every operation
// in the reference's association, two-step scaling, floor not trunc, NO fused
// multiply-add — this file must be compiled with -ffp-contract=off (the build
// also greps the ISA for v_fma/v_mad/v_fmac, see build.py).
//
// Mapping (MI355X-first, not a shader translation): ONE LANE OWNS ONE 8x8
// BLOCK.  Its 64 coefficients live in 64 VGPRs, both 1-D passes run in
// registers with compile-time indices, so there is no transpose, no cross-lane
// traffic and no per-lane scale table.  A wave is 64 horizontally adjacent
// blocks: its coefficient reads cover 8 KB contiguous and its pixel stores are
// 512 B (planes) / 1536 B (RGB) contiguous runs per image row.
//   * RGB kernel: a workgroup is a tile of 64 MCUs of one MCU row: (1<<xdec)*
//     (1<<ydec) luma waves + one Cb wave + one Cr wave.  Chroma waves publish
//     their 8x8 (Cb-128),(Cr-128) samples through LDS (padded, conflict-free),
//     luma waves replicate them (s>>xdec, t>>ydec), convert and store
//     interleaved RGB straight from registers.
//   * YUV kernel: flat — every wave takes 64 consecutive 128-byte blocks of the
//     packed coefficient buffer and stores 8x8 u8 tiles into the padded planes.
// No MFMA: there is no dense contraction here; the kernel is HBM-bound
// (6 B/px in+out) with ~37 VALU lane-ops/px next to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "kernel_params.h"
#include "jga_tune.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

#define DEV static __device__ __forceinline__

// Output pixels are written once and never re-read by the kernel: nontemporal
// (streaming) stores keep them from thrashing L2 — on MI355X a read+write stream
// runs 5.7 TB/s with nt stores vs 3.6-4.8 TB/s with plain ones (tools/membench2,
// profiles/r1_membench2.txt).  (Loads stay plain: the 8 row loads of a lane share L1 lines.)
DEV void st_nt(uint4 *p, const uint4 v) {
  __builtin_nontemporal_store(__builtin_bit_cast(v4u, v), reinterpret_cast<v4u *>(p));
}
DEV void st_nt(uint2 *p, const uint2 v) {
  __builtin_nontemporal_store(__builtin_bit_cast(v2u, v), reinterpret_cast<v2u *>(p));
}

// n / d for n < 2^31 with a host-precomputed reciprocal (kernel_params.h):
// keeps integer division (which hipcc expands through float rcp + fma) out of
// the kernels, so the "no fma in the ISA" build check can be strict.
DEV uint32_t fastdiv(uint32_t n, jga_divisor d) {
  return (uint32_t)(((uint64_t)n*d.mul) >> d.shift);
}

// ---- constants (closed forms) ------------------
#define S0 0.35355339059327373f   // 1/(2*sqrt(2))      3eb504f3
#define S1 0.49039264020161522f   // cos(1*pi/16)/2     3efb14be
#define S2 0.46193976625564337f   // cos(2*pi/16)/2     3eec835e
#define S3 0.41573480615127262f   // cos(3*pi/16)/2     3ed4db31
#define S4 0.35355339059327373f   // cos(4*pi/16)/2     3eb504f3
#define S5 0.27778511650980114f   // cos(5*pi/16)/2     3e8e39da
#define S6 0.19134171618254492f   // cos(6*pi/16)/2     3e43ef15
#define S7 0.097545161008064166f  // cos(7*pi/16)/2     3dc7c5c2
#define C1 1.4142135623730951f    // sqrt(2)
#define C2 1.8477590650225735f    // 2cos(pi/8)
#define C3 1.0823922002923938f    // 2(cos(pi/8)-sin(pi/8))
#define C4 2.6131259297527532f    // 2(cos(pi/8)+sin(pi/8))

// 1-D 8-point scaled IDCT, operation for operation src/dct.c:39-86.
DEV void idct8(float y0, float y1, float y2, float y3, float y4, float y5,
 float y6, float y7, float &x0, float &x1, float &x2, float &x3, float &x4,
 float &x5, float &x6, float &x7) {
  // embedded 4-point DCT-II on the even inputs (dct.c:47-55)
  float e0 = y0 + y4;
  float e1 = y0 - y4;
  float e3 = y2 + y6;
  float e2 = (y2 - y6)*C1 - e3;
  float a0 = e0 + e3;
  float a3 = e0 - e3;
  float a1 = e1 + e2;
  float a2 = e1 - e2;
  // embedded 4-point DST-IV on the odd inputs (dct.c:57-69)
  float p5 = y5 + y3;
  float p6 = y5 - y3;
  float p7 = y1 + y7;
  float p4 = y1 - y7;
  float b7 = p7 + p5;
  float o5 = (p7 - p5)*C1;
  float o8 = (p4 + p6)*C2;
  float o4 = o8 - p4*C3;
  float o6 = o8 - p6*C4;
  float b6 = b7 - o6;
  float b5 = b6 + o5;
  float b4 = b5 - o4;
  // butterflies (dct.c:71-86)
  x0 = a0 + b7;
  x7 = a0 - b7;
  x6 = a1 + b6;
  x1 = a1 - b6;
  x2 = a2 + b5;
  x5 = a2 - b5;
  x4 = a3 + b4;
  x3 = a3 - b4;
}

// Dequantise + scale one coefficient row (8 int16 in 4 dwords) into floats:
//   c = (int16)(level*q)            xjpeg.c:501-503, 524-527 (wraps mod 2^16)
//   t = ((float)c * S[j]) * S[i]    dct.c:107-108 (two roundings)
template <bool DEQUANT>
DEV void load_row(const uint4 raw, const uint32_t *__restrict__ q /*4 dwords*/,
 const float sj, float *t) {
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  const float si[8] = {S0, S1, S2, S3, S4, S5, S6, S7};
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t c = w[d];
    if (DEQUANT) {
      v2us lv = __builtin_bit_cast(v2us, w[d]);
      v2us qv = __builtin_bit_cast(v2us, q[d]);
      c = __builtin_bit_cast(uint32_t, (v2us)(lv*qv));   // v_pk_mul_lo_u16
    }
    float lo = (float)(short)(c & 0xffffu);
    float hi = (float)(short)(c >> 16);
    t[2*d] = (lo*sj)*si[2*d];
    t[2*d + 1] = (hi*sj)*si[2*d + 1];
  }
}

// Scale + row pass (dct.c:105-111) of the block held by this lane.  rows[r] =
// 16 bytes of coefficient row r; q = 64 u16 natural order as 32 dwords.
// z[r*8+i] = row-pass output i of row r.
template <bool DEQUANT>
DEV void row_pass(const uint4 (&rows)[8], const uint32_t *__restrict__ q,
 float (&z)[64]) {
  const float sj[8] = {S0, S1, S2, S3, S4, S5, S6, S7};
#pragma unroll
  for (int r = 0; r < 8; r++) {
    float y[8];
    load_row<DEQUANT>(rows[r], q + 4*r, sj[r], y);
    idct8(y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7], z[r*8 + 0],
     z[r*8 + 1], z[r*8 + 2], z[r*8 + 3], z[r*8 + 4], z[r*8 + 5], z[r*8 + 6],
     z[r*8 + 7]);
  }
}

// Same, with the quantisation rows fetched one at a time from LDS (per-lane
// table pointer: lanes of one wave may belong to different planes).
template <bool DEQUANT>
DEV void row_pass_ldsq(const uint4 (&rows)[8], const uint4 *qp, float (&z)[64]) {
  const float sj[8] = {S0, S1, S2, S3, S4, S5, S6, S7};
#pragma unroll
  for (int r = 0; r < 8; r++) {
    float y[8];
    uint32_t q[4] = {0, 0, 0, 0};
    if (DEQUANT) {
      const uint4 q4 = qp[r];
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    }
    load_row<DEQUANT>(rows[r], q, sj[r], y);
    idct8(y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7], z[r*8 + 0],
     z[r*8 + 1], z[r*8 + 2], z[r*8 + 3], z[r*8 + 4], z[r*8 + 5], z[r*8 + 6],
     z[r*8 + 7]);
  }
}

// Column pass: vector i = row-pass outputs at column i over rows 0..7, +0.5 on
// its first entry (dct.c:112-115), floor (118).  On return t[k*8+i] =
// floor(idct)(row k, col i) as an integer-valued float, already passed through
// the (short) wrap of dct.c:118 when it can matter.
// Returns max |t| over the block.
DEV float col_pass(const float (&z)[64], float (&t)[64]) {
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float o[8];
    idct8(z[0*8 + i] + 0.5f, z[1*8 + i], z[2*8 + i], z[3*8 + i], z[4*8 + i],
     z[5*8 + i], z[6*8 + i], z[7*8 + i], o[0], o[1], o[2], o[3], o[4], o[5],
     o[6], o[7]);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      float f = __builtin_floorf(o[k]);          // dct.c:118 floor
      t[k*8 + i] = f;
    }
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t[k*8 + i])),
       __builtin_fabsf(t[(k + 1)*8 + i]));       // one v_max3_f32 with |abs|
    }
  }
  // (short) cast of dct.c:118: x86-64 converts through int32 and keeps the
  // low 16 bits.  Only reachable with |coefficients| far outside what a JPEG
  // encoder emits, so it is a rarely taken exact path, not the main one.
  if (__builtin_expect(!(m < 32000.0f), 0)) {
#pragma unroll
    for (int n = 0; n < 64; n++) t[n] = (float)(short)(int)t[n];
    m = 32768.0f;
  }
  return m;
}

template <bool DEQUANT>
DEV void idct_block(const uint4 (&rows)[8], const uint32_t *__restrict__ q,
 float (&t)[64]) {
  float z[64];
  row_pass<DEQUANT>(rows, q, z);
  col_pass(z, t);
}


DEV uint32_t pack_u8x4(float a, float b, float c, float d) {
  // v_cvt_pk_u8_f32 converts with saturation to [0,255]; inputs here are
  // already integer-valued so its rounding mode is irrelevant.
  uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0);
  r = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(d, 3, r);
  return r;
}

// RGB stage arithmetic: with Y the clamped luma sample and
// u = Cb-128, v = Cr-128,
//   R = Y + 1.402f*v;  G = (Y + (-0.34414f)*u) + (-0.71414f)*v;  B = Y + 1.772f*u
//   u8 = (int)(clamp(c,0,255) + 0.5f)                     (round half up)
// The kernels evaluate three cheaper forms that are PROVEN bit-identical to it
// over the whole input domain (Y in 0..255, u,v in -128..127: 2x65536 + 16.7M
// cases, exhaustive check in tests/test_rgb_rounding.py).  With yc = Y-128 (the
// clamped IDCT output, an integer-valued float):
//   G = sat(floor((yc + fl(a*u + 128.5)) + b*v))   the +0.5 and the level shift
//       ride on the per-chroma-sample product; Y+x.5 sums are exact where it matters
//   R = sat(rne(yc + fl(1.402f*v + (128 + 2^-12))))  v_cvt_pk_u8_f32 rounds to
//   B = sat(rne(yc + fl(1.772f*u + (128 + 2^-12))))  nearest-even + saturates; it
//       differs from round-half-up only on exact .5 ties, and the 2^-12 nudge
//       lifts every tie without carrying any other value across a boundary.
// Per pixel: 1 med3 + 4 add + 1 floor + 3 cvt_pk (was 15 ops), per chroma sample
// 4 mul + 3 add.
#define Y_HALF 128.5f
#define Y_TIE 128.000244140625f      /* 128 + 2^-12 */

// Chroma products of one chroma row segment (CW samples).
template <int CW>
struct chroma_row {
  float r[CW], g1[CW], g2[CW], b[CW];
  __device__ __forceinline__ void set(const float *u, const float *v) {
#pragma unroll
    for (int c = 0; c < CW; c++) {
      r[c] = 1.402f*v[c] + Y_TIE;
      g1[c] = (-0.34414f)*u[c] + Y_HALF;
      g2[c] = (-0.71414f)*v[c];
      b[c] = 1.772f*u[c] + Y_TIE;
    }
  }
};

// One output row (8 pixels) of a luma block -> 24 RGB bytes in 6 dwords.
// CLAMP = false when no sample of the wave leaves [-128,127] (the clamp is the identity).
template <int XDEC, int CW, bool CLAMP>
DEV void rgb_row(const float *t8, const chroma_row<CW> &cr, uint4 &a, uint2 &b2) {
  float rgb[24];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const float yc = CLAMP ? __builtin_amdgcn_fmed3f(t8[i], -128.0f, 127.0f) : t8[i];
    const int c = i >> XDEC;
    rgb[3*i + 0] = yc + cr.r[c];
    rgb[3*i + 1] = __builtin_floorf((yc + cr.g1[c]) + cr.g2[c]);
    rgb[3*i + 2] = yc + cr.b[c];
  }
  a.x = pack_u8x4(rgb[0], rgb[1], rgb[2], rgb[3]);
  a.y = pack_u8x4(rgb[4], rgb[5], rgb[6], rgb[7]);
  a.z = pack_u8x4(rgb[8], rgb[9], rgb[10], rgb[11]);
  a.w = pack_u8x4(rgb[12], rgb[13], rgb[14], rgb[15]);
  b2.x = pack_u8x4(rgb[16], rgb[17], rgb[18], rgb[19]);
  b2.y = pack_u8x4(rgb[20], rgb[21], rgb[22], rgb[23]);
}

// Store one RGB row: vector stores when the row is dword-aligned and fully
// inside the image, byte stores at the right edge / for unaligned pitches.
DEV void store_rgb_row(uint8_t *o, const uint4 a, const uint2 b2, bool fast,
 int x0, int width) {
  if (fast) {
    st_nt(reinterpret_cast<uint4 *>(o), a);
    st_nt(reinterpret_cast<uint2 *>(o + 16), b2);
  }
  else {
    const uint32_t w[6] = {a.x, a.y, a.z, a.w, b2.x, b2.y};
#pragma unroll
    for (int n = 0; n < 24; n++) {
      if (x0 + n/3 < width) o[n] = (uint8_t)((w[n >> 2] >> (8*(n & 3))) & 255u);
    }
  }
}

// The same row for a whole wave whose 64 lanes own 64 horizontally adjacent blocks, all
// inside the image: the wave's 1536 bytes are contiguous, so they go through a per-wave LDS
// line and leave as 16 bytes per lane, back to back (1 KB + 512 B bursts) instead of
// 16 + 8 bytes at a 24-byte stride.  `row0` = address of lane 0's first pixel (16-byte aligned);
// `nchunks` = 16-byte pieces to store: 96 for a full wave, 1.5 per block when only the first
// (even number of) lanes hold blocks inside the image — the last tile of a row.
DEV void store_rgb_row_wave(uint32_t *wstage, int lane, uint8_t *row0, const uint4 a, const uint2 b2,
 int nchunks) {
  uint2 *w2 = reinterpret_cast<uint2 *>(wstage) + lane*3;
  w2[0] = make_uint2(a.x, a.y);
  w2[1] = make_uint2(a.z, a.w);
  w2[2] = b2;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint4 *r4 = reinterpret_cast<const uint4 *>(wstage);
  if (lane < nchunks) st_nt(reinterpret_cast<uint4 *>(row0) + lane, r4[lane]);
  if (64 + lane < nchunks) st_nt(reinterpret_cast<uint4 *>(row0) + 64 + lane, r4[64 + lane]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Fetch the 8 coefficient rows of the block at `src` (128 contiguous bytes).
DEV void load_block_direct(const int16_t *__restrict__ src, uint4 (&rows)[8]) {
  const uint4 *p = reinterpret_cast<const uint4 *>(src);
#pragma unroll
  for (int r = 0; r < 8; r++) rows[r] = p[r];   // plain loads: the 8 rows share L1 lines
}

// DC values handed in beside the planes (the GPU entropy stage leaves DC prediction out of its
// write pass: jga_huff_decode_split): coefficient 0 of the block in buffer slot `slot` comes from
// P.dc instead of from the planes.  One 2-byte load per block; wave-uniform test.
DEV void take_dc(const jga_kparams &P, int img, long long slot, uint4 (&rows)[8]) {
  if (P.dc) {
    const uint32_t v = (uint16_t)P.dc[(long long)img*P.dc_stride + slot];
    rows[0].x = (rows[0].x & 0xffff0000u) | v;
  }
}

// Coalesced fetch of a wave's 64 consecutive blocks (8 KB) through LDS: each
// instruction moves 1 KB contiguous (8 blocks); the LDS image of region k is
// [row ^ ((k>>1)&1)][block&7] in 16-byte slots so that the per-lane
// ds_read_b128 of "my block, row r" is bank-conflict free (profiles/design_diary_r3_r5.md §3).
DEV void load_block_staged(const int16_t *__restrict__ wave_src, int lane,
 uint4 *__restrict__ lds /* 512 slots of this wave */, uint4 (&rows)[8]) {
  const uint4 *p = reinterpret_cast<const uint4 *>(wave_src);
  const int j = lane & 7, rr = lane >> 3;
  uint4 tmp[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int r = rr ^ ((k >> 1) & 1);
    tmp[k] = p[k*64 + j*8 + r];                 // block 8k+j, row r
  }
#pragma unroll
  for (int k = 0; k < 8; k++) lds[k*64 + lane] = tmp[k];
  // the slots are private to this wave: order its own LDS writes before its
  // own LDS reads (other lanes' data), no workgroup barrier needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int k = lane >> 3;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    rows[r] = lds[k*64 + ((r ^ ((k >> 1) & 1))*8 + j)];
  }
}

// ---------------------------------------------------------------------------
// YUV stage: coefficient planes -> padded u8 planes (JPEG_DECODE_YUV).
// ---------------------------------------------------------------------------
template <bool DEQUANT, bool STAGED>
__global__ __launch_bounds__(256) void jga_idct_yuv_kernel(const jga_kparams P) {
  __shared__ uint4 stage[STAGED ? 4*512 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int img = blockIdx.y;
  const int wslot = (blockIdx.x*4 + wave)*64;      // first slot of this wave
  if (wslot >= P.slots_per_image) return;
  const bool in_range = wslot + lane < P.slots_per_image;
  const int s = in_range ? wslot + lane : P.slots_per_image - 1;
  // slot -> (plane, bx, by): inverse of xjpeg.c:556-561
  int pl = 0;
  if (P.nplanes == 3) pl = s >= P.plane_slot0[2] ? 2 : s >= P.plane_slot0[1] ? 1 : 0;
  const int ls = s - P.plane_slot0[pl];
  const int row = (int)fastdiv(ls, P.div_w0), col = ls - row*P.w0_blocks;
  const int xdec = P.plane_xdec[pl];
  const int hb = P.w0_blocks >> xdec;
  const int sub = (int)fastdiv(col, P.div_hb[pl]);
  const int bx = col - sub*hb;
  const int by = (row << xdec) + sub;
  const bool valid = in_range && by < P.plane_vblocks[pl];

  const int16_t *cbase = P.coef + (long long)img*P.coef_stride;
  uint4 rows[8];
  if (STAGED && wslot + 64 <= P.slots_per_image) {
    load_block_staged(cbase + (long long)wslot*64, lane, stage + wave*512,
     rows);
  }
  else load_block_direct(cbase + (long long)s*64, rows);
  take_dc(P, img, s, rows);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(P.qtab)
   + ((long long)img*3 + pl)*32;
  uint32_t qv[32];
  if (DEQUANT) {
#pragma unroll
    for (int n = 0; n < 32; n++) qv[n] = q[n];
  }
  float t[64];
  idct_block<DEQUANT>(rows, qv, t);
  if (!valid) return;
  uint8_t *dst = P.out + (long long)img*P.out_stride + P.plane_data_off[pl]
   + ((long long)by*8*P.plane_hblocks[pl] + bx)*8;
  const long long pitch = (long long)P.plane_hblocks[pl]*8;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint2 v;
    v.x = pack_u8x4(t[k*8 + 0] + 128.0f, t[k*8 + 1] + 128.0f,
     t[k*8 + 2] + 128.0f, t[k*8 + 3] + 128.0f);
    v.y = pack_u8x4(t[k*8 + 4] + 128.0f, t[k*8 + 5] + 128.0f,
     t[k*8 + 6] + 128.0f, t[k*8 + 7] + 128.0f);
    st_nt(reinterpret_cast<uint2 *>(dst + k*pitch), v);
  }
}

// ---------------------------------------------------------------------------
// RGB stage: coefficient planes -> interleaved RGB8 (JPEG_DECODE_RGB).
// A workgroup is a tile of TILE MCUs of one MCU row: its LW*LH*TILE luma blocks
// fill NLW waves (row-major over the tile's luma block rows), its TILE Cb and
// TILE Cr blocks fill the chroma wave(s).  TILE = 32 gives small 3-wave groups
// (two luma waves + one wave holding Cb in lanes 0-31 and Cr in lanes 32-63),
// so many groups interleave on a CU and the one barrier couples few waves;
// 4:4:4 keeps TILE = 64 (one wave each for Y, Cb, Cr).
// ---------------------------------------------------------------------------
template <int XDEC, int YDEC>
struct rgb_cfg {
  static constexpr int LW = 1 << XDEC, LH = 1 << YDEC;
  static constexpr int TILE = (LW*LH == 1) ? 64 : 32;     // MCUs per tile
  static constexpr int ROWLEN = TILE*LW;                   // luma blocks per tile row
  static constexpr int NLW = LW*LH*TILE/64;                // luma waves
  static constexpr int NCW = 2*TILE/64;                    // chroma waves
  static constexpr int THREADS = (NLW + NCW)*64;
  static constexpr int CW = 8 >> XDEC, CH = 8 >> YDEC;     // chroma patch per luma block
  // Work balance: a luma lane would run IDCT + 8 rows of colour conversion, a chroma lane
  // IDCT + publish only, and the chroma waves would idle through the conversion.  So the luma
  // lanes convert their first OWN rows from registers and hand the last SHARE rows to the
  // chroma lanes through LDS ([row][luma block][8] floats).
  static constexpr int NLB = LW*LH*TILE;                   // luma blocks per tile
  // (measured, tools/ab_variants.sh: 4:2:0 best at 1 shared row — 0.430 ms against 0.445 with
  // none and 0.433 with two; 4:4:4 at 3, -4 %: the hardware back-fills most idle slots itself)
#ifndef JGA_SHARE_1
#define JGA_SHARE_1 3                /* 4:4:4: 1 luma block per MCU */
#define JGA_SHARE_2 4                /* 4:2:2, 4:4:0 */
#define JGA_SHARE_4 1                /* 4:2:0, 4:1:1 */
#endif
  static constexpr int SHARE = (LW*LH == 1) ? JGA_SHARE_1 : (LW*LH == 2) ? JGA_SHARE_2 : JGA_SHARE_4;
  static constexpr int OWN = 8 - SHARE;
  // LDS floats: hand-off [comp][row][chroma block][8] + the image's 3 q tables + shared rows
  static constexpr int CHROMA_FLOATS = 2*8*TILE*8;
  static constexpr int YSHARE_FLOATS = SHARE*NLB*8;
  static constexpr int LDS_FLOATS = CHROMA_FLOATS + 3*32 + YSHARE_FLOATS;
  // Wave-staged pixel stores (store_rgb_row_wave): when a wave's 64 lanes are 64 adjacent
  // blocks of one row.  -6 % time at 4:2:0; not at 4:4:4, whose larger LDS footprint already
  // limits it to few workgroups per CU (+12 % there).
  static constexpr bool STAGE_STORES = (ROWLEN % 64 == 0) && (LW*LH >= 2);
};

// Upsample + convert + store pixel rows [0, NROWS) of one luma block.
template <int XDEC, int YDEC, bool CLAMP, int NROWS>
DEV void colour_rows(const float (&t)[64], const float *ub, const float *vb, uint8_t *obase,
 long long pitch, bool fast, int x0, int y0, int width, int height, int wchunks, uint32_t *wstage,
 int lane) {
  const bool wfast = wchunks != 0;
  typedef rgb_cfg<XDEC, YDEC> cfg;
  chroma_row<cfg::CW> cr;
#pragma unroll
  for (int k = 0; k < NROWS; k++) {
    if ((k & (cfg::LH - 1)) == 0) {
      float u[cfg::CW], v[cfg::CW];
#pragma unroll
      for (int c = 0; c < cfg::CW; c++) {
        u[c] = ub[(k >> YDEC)*(cfg::TILE*8) + c];
        v[c] = vb[(k >> YDEC)*(cfg::TILE*8) + c];
      }
      cr.set(u, v);
    }
    uint4 a;
    uint2 b;
    rgb_row<XDEC, cfg::CW, CLAMP>(t + k*8, cr, a, b);
    if (wfast) {                       // wave-uniform (as is y0 when wfast)
      if (y0 + k < height) store_rgb_row_wave(wstage, lane, obase - lane*24 + k*pitch, a, b, wchunks);
    }
    else if (y0 + k < height) store_rgb_row(obase + k*pitch, a, b, fast, x0, width);
  }
}

template <int XDEC, int YDEC, bool DEQUANT>
__global__ __launch_bounds__((rgb_cfg<XDEC, YDEC>::THREADS))
void jga_idct_rgb_kernel(const jga_kparams P) {
  typedef rgb_cfg<XDEC, YDEC> cfg;
  __shared__ float lds[cfg::LDS_FLOATS];
  float *chroma = lds;
  uint4 *qlds = reinterpret_cast<uint4 *>(lds + cfg::CHROMA_FLOATS);
  float *yshare = lds + cfg::CHROMA_FLOATS + 3*32;
  __shared__ __attribute__((aligned(16))) uint32_t stage_mem[cfg::STAGE_STORES ? cfg::NLW + cfg::NCW : 1][384];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *wstage = stage_mem[cfg::STAGE_STORES ? wave : 0];
  const int tx = blockIdx.x, mrow = blockIdx.y, img = blockIdx.z;
  const int cbx0 = tx*cfg::TILE;               // first MCU / chroma block of the tile
  const long long rs = (long long)P.w0_blocks*64;

  const bool is_luma = wave < cfg::NLW;
  int pl, bx, by, cb, comp = 0, subx = 0, suby = 0;
  if (is_luma) {
    const int idx = wave*64 + lane;
    suby = idx/cfg::ROWLEN;
    const int lx = idx - suby*cfg::ROWLEN;
    pl = 0;
    bx = cbx0*cfg::LW + lx;
    by = mrow*cfg::LH + suby;
    cb = lx >> XDEC;
    subx = lx & (cfg::LW - 1);
  }
  else {
    const int cidx = (wave - cfg::NLW)*64 + lane;
    comp = cidx/cfg::TILE;
    cb = cidx - comp*cfg::TILE;
    pl = 1 + comp;
    bx = cbx0 + cb;
    by = mrow;
  }
  const int hblocks = P.plane_hblocks[pl];
  const bool valid = bx < hblocks;
  const int cbxl = valid ? bx : hblocks - 1;   // tail lanes reload a real block
  const int xdec = is_luma ? 0 : XDEC;
  const int16_t *src = P.coef + (long long)img*P.coef_stride + P.plane_coef_off[pl]
   + rs*(by >> xdec) + (rs >> xdec)*(by & ((1 << xdec) - 1)) + (long long)cbxl*64;

  uint4 rows[8];
  load_block_direct(src, rows);                // in flight across the barrier below
  take_dc(P, img, (src - (P.coef + (long long)img*P.coef_stride)) >> 6, rows);
  if (DEQUANT) {
    if (threadIdx.x < 24) {
      qlds[threadIdx.x] = reinterpret_cast<const uint4 *>(P.qtab + (long long)img*192)[threadIdx.x];
    }
    __syncthreads();
  }
  float z[64], t[64];
  row_pass_ldsq<DEQUANT>(rows, qlds + pl*8, z);
  const float tmax = col_pass(z, t);
  // Clamping to [-128,127] is the identity unless some sample overshoots; decide once per
  // wave (64 blocks) and drop the 64 v_med3 per lane when nobody does — the usual case.
  const bool clip = __builtin_amdgcn_ballot_w64(tmax > 127.0f) != 0ull;

  if (!is_luma) {
    // publish (sample-128) clamped to [-128,127] == clamp255(s+128)-128
    float *dst = chroma + ((comp*8)*cfg::TILE + cb)*8;
    if (clip) {
#pragma unroll
      for (int n = 0; n < 64; n += 4) {
        v4f v;
        v.x = __builtin_amdgcn_fmed3f(t[n + 0], -128.0f, 127.0f);
        v.y = __builtin_amdgcn_fmed3f(t[n + 1], -128.0f, 127.0f);
        v.z = __builtin_amdgcn_fmed3f(t[n + 2], -128.0f, 127.0f);
        v.w = __builtin_amdgcn_fmed3f(t[n + 3], -128.0f, 127.0f);
        *reinterpret_cast<v4f *>(dst + (n >> 3)*(cfg::TILE*8) + (n & 4)) = v;
      }
    }
    else {
#pragma unroll
      for (int n = 0; n < 64; n += 4) {
        v4f v;
        v.x = t[n + 0]; v.y = t[n + 1]; v.z = t[n + 2]; v.w = t[n + 3];
        *reinterpret_cast<v4f *>(dst + (n >> 3)*(cfg::TILE*8) + (n & 4)) = v;
      }
    }
  }
  else {
    // hand the last SHARE rows of this luma block (clamped) to the chroma lanes
    const int idx = wave*64 + lane;
#pragma unroll
    for (int r = cfg::OWN; r < 8; r++) {
      float *dst = yshare + ((r - cfg::OWN)*cfg::NLB + idx)*8;
#pragma unroll
      for (int h = 0; h < 8; h += 4) {
        v4f v;
        v.x = t[r*8 + h]; v.y = t[r*8 + h + 1]; v.z = t[r*8 + h + 2]; v.w = t[r*8 + h + 3];
        if (clip) {
          v.x = __builtin_amdgcn_fmed3f(v.x, -128.0f, 127.0f);
          v.y = __builtin_amdgcn_fmed3f(v.y, -128.0f, 127.0f);
          v.z = __builtin_amdgcn_fmed3f(v.z, -128.0f, 127.0f);
          v.w = __builtin_amdgcn_fmed3f(v.w, -128.0f, 127.0f);
        }
        *reinterpret_cast<v4f *>(dst + h) = v;
      }
    }
  }
  __syncthreads();
  const long long pitch = (long long)P.width*3;
  uint8_t *img_out = P.out + (long long)img*P.out_stride;

  if (is_luma) {
    const int x0 = bx*8, y0 = by*8;
    // the wave's first n lanes (n even) inside the image and the run 16-byte aligned: staged
    // stores, in which the other lanes only lend a hand with the copy
    const unsigned long long inside = __builtin_amdgcn_ballot_w64(valid && x0 + 8 <= P.width);
    const int nin = (int)__builtin_popcountll(inside);
    const bool wfast = cfg::STAGE_STORES && P.out_aligned && (pitch & 15) == 0
     && nin >= 2 && (nin & 1) == 0 && inside == (nin == 64 ? ~0ull : (1ull << nin) - 1ull)
     && (((uintptr_t)img_out + (unsigned long long)__builtin_amdgcn_readfirstlane(x0)*3u) & 15u) == 0;
    const int wchunks = wfast ? nin*3/2 : 0;
    if (!valid && !wfast) return;
    // chroma patch of this luma block: rows suby*CH.., cols subx*CW..
    const float *ub = chroma + ((suby*cfg::CH)*cfg::TILE + cb)*8 + subx*cfg::CW;
    const float *vb = ub + 8*cfg::TILE*8;
    uint8_t *obase = img_out + (long long)y0*pitch + (long long)x0*3;
    const bool fast = P.out_aligned && x0 + 8 <= P.width;
    if (clip) colour_rows<XDEC, YDEC, true, cfg::OWN>(t, ub, vb, obase, pitch, fast, x0, y0, P.width, P.height, wchunks, wstage, lane);
    else colour_rows<XDEC, YDEC, false, cfg::OWN>(t, ub, vb, obase, pitch, fast, x0, y0, P.width, P.height, wchunks, wstage, lane);
    return;
  }

  // chroma lanes: the shared rows, one (luma block, row) unit at a time
  const int cl = (wave - cfg::NLW)*64 + lane;                // 0 .. NCW*64-1
#pragma unroll
  for (int u0 = 0; u0 < cfg::SHARE*cfg::NLB; u0 += cfg::NCW*64) {
    const int unit = u0 + cl;
    const int rs = unit/cfg::NLB, idx = unit - rs*cfg::NLB;  // NLB is a power of two
    const int k = cfg::OWN + rs;                             // pixel row inside the block
    const int sy = idx/cfg::ROWLEN, lx = idx - sy*cfg::ROWLEN;
    const int ubx = cbx0*cfg::LW + lx, uby = mrow*cfg::LH + sy;
    const bool live = unit < cfg::SHARE*cfg::NLB && ubx < P.plane_hblocks[0];
    const int ucb = lx >> XDEC, usubx = lx & (cfg::LW - 1);
    const float *ub = chroma + (live ? ((sy*cfg::CH + (k >> YDEC))*cfg::TILE + ucb)*8 + usubx*cfg::CW : 0);
    const float *vb = ub + 8*cfg::TILE*8;
    float u[cfg::CW], v[cfg::CW];
#pragma unroll
    for (int c = 0; c < cfg::CW; c++) { u[c] = ub[c]; v[c] = vb[c]; }
    chroma_row<cfg::CW> cr;
    cr.set(u, v);
    float y8[8];
    {
      const v4f *ys = reinterpret_cast<const v4f *>(yshare + (live ? (rs*cfg::NLB + idx)*8 : 0));
      const v4f a = ys[0], b = ys[1];
      y8[0] = a.x; y8[1] = a.y; y8[2] = a.z; y8[3] = a.w;
      y8[4] = b.x; y8[5] = b.y; y8[6] = b.z; y8[7] = b.w;
    }
    uint4 a;
    uint2 b;
    rgb_row<XDEC, cfg::CW, false>(y8, cr, a, b);             // already clamped by its luma lane
    const int x0 = ubx*8, yy = uby*8 + k;
    const unsigned long long inside = __builtin_amdgcn_ballot_w64(live && x0 + 8 <= P.width);
    const int nin = (int)__builtin_popcountll(inside);
    const bool wfast = cfg::STAGE_STORES && P.out_aligned && (pitch & 15) == 0
     && nin >= 2 && (nin & 1) == 0 && inside == (nin == 64 ? ~0ull : (1ull << nin) - 1ull)
     && (((uintptr_t)img_out + (unsigned long long)__builtin_amdgcn_readfirstlane(x0)*3u) & 15u) == 0;
    uint8_t *o = img_out + (long long)yy*pitch + (long long)x0*3;
    if (wfast) {                       // (yy is wave-uniform then)
      if (yy < P.height) store_rgb_row_wave(wstage, lane, o - lane*24, a, b, nin*3/2);
    }
    else if (live && yy < P.height) {
      store_rgb_row(o, a, b, P.out_aligned && x0 + 8 <= P.width, x0, P.width);
    }
  }
}

// ---------------------------------------------------------------------------
// Row-parallel colour stage, for the samplings whose tile holds exactly as many (luma block,
// pixel row) units per PR rows as the workgroup has lanes: 4:4:4 (BASELINE config 3,
// "exercises the no-upsample path"; PR = 3), 4:2:2 and 4:4:0 (PR = 2).
//
// In the tile kernel above a luma lane converts its own block: it needs every chroma sample of
// its MCU (4:4:4: 128 floats per lane, a 32 KB hand-off per 64 MCUs that pins the kernel at 12
// waves per CU, leaves no LDS for the wave-staged stores, and gives the luma wave 40 % more work
// than the chroma waves).  Here the IDCT stays block-per-lane (same waves, same lanes, same
// loads), but the colour stage runs in phases of PR pixel rows: every lane publishes rows
// [p*PR, p*PR+PR) of its block (clamped) and the chroma rows that go with them, then lane u
// converts ONE (luma block u % NLB, row p*PR + u / NLB) unit — 8 pixels.  All lanes carry equal
// work, LDS is 8-18 KB, and where a wave's 64 units are 64 adjacent blocks of one row (4:4:4,
// 4:2:2) their 1536 output bytes leave through the wave-staged stores.  Same arithmetic per
// pixel as everywhere (rgb_row / chroma_row).
// ---------------------------------------------------------------------------
template <int XDEC, int YDEC>
struct rows_cfg {
  typedef rgb_cfg<XDEC, YDEC> tile;
  static constexpr int PR = tile::THREADS/tile::NLB;         // luma pixel rows per phase
  static constexpr int NPH = (8 + PR - 1)/PR;                // phases
  static constexpr int CSLOTS = YDEC ? tile::LH : PR;        // chroma rows published per phase and component
  static constexpr bool STAGED = tile::ROWLEN % 64 == 0;
  // float offset of half `hh` (4 samples) of chroma block `cb`'s published row `row` (= comp*CSLOTS + slot): where
  // a luma block takes a whole chroma row (no horizontal decimation) consecutive lanes read consecutive chroma
  // blocks — two planes of half rows; where two luma blocks share one, consecutive lanes read its two halves —
  // whole rows, as the tile kernel keeps them
  static __device__ __forceinline__ int chroma_at(int row, int cb, int hh) {
    return XDEC == 0 ? ((row*2 + hh)*tile::TILE + cb)*4 : (row*tile::TILE + cb)*8 + hh*4;
  }
  static_assert(PR*tile::NLB == tile::THREADS, "one unit per lane and phase");
  static_assert(tile::NLB % 64 == 0, "a wave converts one row index");
};

template <int XDEC, int YDEC, bool DEQUANT>
__global__ __launch_bounds__((rgb_cfg<XDEC, YDEC>::THREADS))
void jga_idct_rgb_rows_kernel(const jga_kparams P) {
  typedef rgb_cfg<XDEC, YDEC> cfg;
  typedef rows_cfg<XDEC, YDEC> rc;
  // published rows, as two planes of HALF rows — [row in phase][half][luma block][4], and for the chroma rows
  // rows_cfg::chroma_at — so that the 64 lanes of a ds_write_b128 / ds_read_b128 touch 64 consecutive 16-byte chunks
  // (whole rows at a 32-byte lane stride used half the banks per pass: 3.0e7 conflict cycles per 24 frames against
  // 2.1e7 cycles of LDS instructions; now 0: -1.7 % at 4:4:4, profiles/r5_kernel_budgets.md.  The same re-layout of
  // the tile kernel's hand-off halved ITS conflicts and bought nothing: not kept.)
  __shared__ __attribute__((aligned(16))) float ypub[rc::PR*cfg::NLB*8];
  __shared__ __attribute__((aligned(16))) float cpub[2*rc::CSLOTS*cfg::TILE*8];
  __shared__ uint4 qlds[24];
  __shared__ __attribute__((aligned(16))) uint32_t stage_mem[rc::STAGED ? cfg::NLW + cfg::NCW : 1][384];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *wstage = stage_mem[rc::STAGED ? wave : 0];
  const int tx = blockIdx.x, mrow = blockIdx.y, img = blockIdx.z;
  const int cbx0 = tx*cfg::TILE;
  const long long rs = (long long)P.w0_blocks*64;

  // the block this lane transforms: as in the tile kernel
  const bool is_luma = wave < cfg::NLW;
  int pl, bx, by, cb = 0, comp = 0, myidx = 0, mysuby = 0;
  if (is_luma) {
    myidx = wave*64 + lane;
    mysuby = myidx/cfg::ROWLEN;
    pl = 0;
    bx = cbx0*cfg::LW + (myidx - mysuby*cfg::ROWLEN);
    by = mrow*cfg::LH + mysuby;
  }
  else {
    const int cidx = (wave - cfg::NLW)*64 + lane;
    comp = cidx/cfg::TILE;
    cb = cidx - comp*cfg::TILE;
    pl = 1 + comp;
    bx = cbx0 + cb;
    by = mrow;
  }
  const int hblocks = P.plane_hblocks[pl];
  const int bxl = bx < hblocks ? bx : hblocks - 1;     // tail lanes reload a real block
  const int xdec = is_luma ? 0 : XDEC;
  const int16_t *src = P.coef + (long long)img*P.coef_stride + P.plane_coef_off[pl]
   + rs*(by >> xdec) + (rs >> xdec)*(by & ((1 << xdec) - 1)) + (long long)bxl*64;
  uint4 rows[8];
  load_block_direct(src, rows);                      // in flight across the barrier below
  take_dc(P, img, (src - (P.coef + (long long)img*P.coef_stride)) >> 6, rows);
  if (DEQUANT) {
    if (threadIdx.x < 24) {
      qlds[threadIdx.x] = reinterpret_cast<const uint4 *>(P.qtab + (long long)img*192)[threadIdx.x];
    }
    __syncthreads();
  }
  float z[64], t[64];
  row_pass_ldsq<DEQUANT>(rows, qlds + pl*8, z);
  const float tmax = col_pass(z, t);
  // Clamping to [-128,127] (the level shift's clamp) is the identity unless some sample overshoots: decided once per
  // wave — 64 blocks — and done IN PLACE, all 64 values of the lane under one wave-uniform branch (a test around
  // each group of four published values used to be turned into med3 + select by the compiler: two instructions
  // per value instead of none), so that the publishing below is the same code either way.  The kernel runs on
  // its VALU issue rate (1 372 instructions per wave, profiles/r5_kernel_budgets.md): these 64 are 4.7 % of it.
  if (__builtin_amdgcn_ballot_w64(tmax > 127.0f) != 0ull) {
#pragma unroll
    for (int n = 0; n < 64; n++) t[n] = __builtin_amdgcn_fmed3f(t[n], -128.0f, 127.0f);
  }

  // the unit this lane converts in every phase: luma block uidx, row urr of the phase
  const int u = threadIdx.x;
  const int urr = __builtin_amdgcn_readfirstlane(u/cfg::NLB);
  const int uidx = u - urr*cfg::NLB;
  const int usuby = uidx/cfg::ROWLEN, ulx = uidx - usuby*cfg::ROWLEN;
  const int ubx = cbx0*cfg::LW + ulx, uby = mrow*cfg::LH + usuby;
  const bool uvalid = ubx < P.plane_hblocks[0];
  const int ucb = ulx >> XDEC, usubx = ulx & (cfg::LW - 1);
  const int uslot = YDEC ? usuby : urr;
  const long long pitch = (long long)P.width*3;
  uint8_t *img_out = P.out + (long long)img*P.out_stride;
  const int x0 = ubx*8;
  const unsigned long long inside = __builtin_amdgcn_ballot_w64(uvalid && x0 + 8 <= P.width);
  const int nin = (int)__builtin_popcountll(inside);
  const bool wfast = rc::STAGED && P.out_aligned && (pitch & 15) == 0 && nin >= 2 && (nin & 1) == 0
   && inside == (nin == 64 ? ~0ull : (1ull << nin) - 1ull)
   && (((uintptr_t)img_out + (unsigned long long)__builtin_amdgcn_readfirstlane(x0)*3u) & 15u) == 0;
  const bool fast = P.out_aligned && x0 + 8 <= P.width;

#pragma unroll
  for (int p = 0; p < rc::NPH; p++) {
    // publish (clamped above like clamp255(s+128)-128)
    if (is_luma) {
#pragma unroll
      for (int rr = 0; rr < rc::PR; rr++) {
        const int r = p*rc::PR + rr;
        if (r < 8) {
#pragma unroll
          for (int h = 0; h < 8; h += 4) {
            v4f v;
            v.x = t[r*8 + h]; v.y = t[r*8 + h + 1]; v.z = t[r*8 + h + 2]; v.w = t[r*8 + h + 3];
            *reinterpret_cast<v4f *>(ypub + ((rr*2 + (h >> 2))*cfg::NLB + myidx)*4) = v;
          }
        }
      }
    }
    else {
#pragma unroll
      for (int sl = 0; sl < rc::CSLOTS; sl++) {
        // chroma row that goes with luma row p*PR + rr of block row suby: (suby*8 + p*PR + rr) >> YDEC
        const int crow = YDEC ? sl*4 + p : p*rc::PR + sl;
        if (crow < 8) {
#pragma unroll
          for (int h = 0; h < 8; h += 4) {
            v4f v;
            v.x = t[crow*8 + h]; v.y = t[crow*8 + h + 1]; v.z = t[crow*8 + h + 2]; v.w = t[crow*8 + h + 3];
            *reinterpret_cast<v4f *>(cpub + rc::chroma_at(comp*rc::CSLOTS + sl, cb, h >> 2)) = v;
          }
        }
      }
    }
    __syncthreads();
    const int k = p*rc::PR + urr;                    // pixel row inside the luma block (wave-uniform)
    if (k < 8) {
      const v4f y0 = *reinterpret_cast<const v4f *>(ypub + ((urr*2 + 0)*cfg::NLB + uidx)*4);
      const v4f y1 = *reinterpret_cast<const v4f *>(ypub + ((urr*2 + 1)*cfg::NLB + uidx)*4);
      const float y8[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
      // the CW chroma samples of this luma block's row: samples [usubx*CW, usubx*CW + CW) of chroma block ucb's row
      float u8[cfg::CW], v8[cfg::CW];
#pragma unroll
      for (int c = 0; c < cfg::CW; c += 4) {
        const int hh = (usubx*cfg::CW + c) >> 2;
        const v4f a = *reinterpret_cast<const v4f *>(cpub + rc::chroma_at(0*rc::CSLOTS + uslot, ucb, hh));
        const v4f b = *reinterpret_cast<const v4f *>(cpub + rc::chroma_at(1*rc::CSLOTS + uslot, ucb, hh));
        u8[c] = a.x; u8[c + 1] = a.y; u8[c + 2] = a.z; u8[c + 3] = a.w;
        v8[c] = b.x; v8[c + 1] = b.y; v8[c + 2] = b.z; v8[c + 3] = b.w;
      }
      chroma_row<cfg::CW> cr;
      cr.set(u8, v8);
      uint4 a;
      uint2 b;
      rgb_row<XDEC, cfg::CW, false>(y8, cr, a, b);   // published values are clamped already
      const int yy = uby*8 + k;
      uint8_t *o = img_out + (long long)yy*pitch + (long long)x0*3;
      if (yy < P.height) {
        if (wfast) store_rgb_row_wave(wstage, lane, o - lane*24, a, b, nin*3/2);
        else if (uvalid) store_rgb_row(o, a, b, fast, x0, P.width);
      }
    }
    if (p + 1 < rc::NPH) __syncthreads();            // the next phase overwrites the published rows
  }
}

// Grey: one plane, img->pixels is 1 B/px at the true size (ungrey.fs.glsl:18;
// pixel layout src/jpeg_wrap.c:215-219).  Flat over the luma raster.
template <bool DEQUANT, bool STAGED>
__global__ __launch_bounds__(256) void jga_idct_grey_kernel(const jga_kparams P) {
  __shared__ uint4 stage[STAGED ? 4*512 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int img = blockIdx.y;
  const int wslot = (blockIdx.x*4 + wave)*64;
  if (wslot >= P.slots_per_image) return;
  const bool in_range = wslot + lane < P.slots_per_image;
  const int s = in_range ? wslot + lane : P.slots_per_image - 1;
  const int by = (int)fastdiv(s, P.div_w0), bx = s - by*P.w0_blocks;
  const int16_t *cbase = P.coef + (long long)img*P.coef_stride;
  uint4 rows[8];
  if (STAGED && wslot + 64 <= P.slots_per_image) {
    load_block_staged(cbase + (long long)wslot*64, lane, stage + wave*512,
     rows);
  }
  else load_block_direct(cbase + (long long)s*64, rows);
  take_dc(P, img, s, rows);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(P.qtab) + (long long)img*3*32;
  uint32_t qv[32];
  if (DEQUANT) {
#pragma unroll
    for (int n = 0; n < 32; n++) qv[n] = q[n];
  }
  float t[64];
  idct_block<DEQUANT>(rows, qv, t);
  if (!in_range) return;
  const int x0 = bx*8, y0 = by*8;
  uint8_t *obase = P.out + (long long)img*P.out_stride + (long long)y0*P.width + x0;
  const bool fast = P.out_aligned && x0 + 8 <= P.width;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (y0 + k >= P.height) continue;
    uint8_t *o = obase + (long long)k*P.width;
    uint2 v;
    v.x = pack_u8x4(t[k*8 + 0] + 128.0f, t[k*8 + 1] + 128.0f,
     t[k*8 + 2] + 128.0f, t[k*8 + 3] + 128.0f);
    v.y = pack_u8x4(t[k*8 + 4] + 128.0f, t[k*8 + 5] + 128.0f,
     t[k*8 + 6] + 128.0f, t[k*8 + 7] + 128.0f);
    if (fast) st_nt(reinterpret_cast<uint2 *>(o), v);
    else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (x0 + i < P.width) o[i] = (uint8_t)(((i < 4 ? v.x : v.y) >> (8*(i & 3))) & 255u);
      }
    }
  }
}

// Pass 3 on its own (the reference's "simple twin", res/yuv.fs.glsl:16-24 = the
// matrix of res/unyuv.fs.glsl:12-16, 48 on u8 planes): Y/Cb/Cr planes resident in HBM
// (YUV-stage layout: planes at plane_data_off, padded, pitch = plane width) ->
// img->pixels.  Used when a caller stops the decode at the YUV stage (the harness's
// `-o yuv`).  SURVEY.md A.5 arithmetic, evaluated literally.  One thread = 8 pixels
// of one row; HBM-bound: reads 1 + 2/(LW*LH) bytes, writes 3 bytes per pixel.
__global__ __launch_bounds__(256) void jga_yuv_rgb_kernel(const jga_kparams P, int uxdec, int uydec, int vxdec, int vydec) {
  // 8 pixels per thread: Y as one 8-byte load, chroma as 8 >> xdec bytes, 24 output bytes.  A
  // wave's 64 threads cover 512 consecutive pixels of one row = 1536 contiguous output bytes,
  // which leave through the wave's LDS line as 16 bytes per lane (store_rgb_row_wave) when the
  // row is 16-byte aligned: 24-byte pieces at a 24-byte stride cost the memory pipeline far more.
  __shared__ __attribute__((aligned(16))) uint32_t stage_mem[4][384];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x0 = (blockIdx.x*256 + threadIdx.x)*8;
  const int y = blockIdx.y, img = blockIdx.z;
  const bool live = x0 < P.width;
  if (__builtin_amdgcn_ballot_w64(live) == 0ull) return;
  const int xl = live ? x0 : 0;                      // (idle lanes of a live wave read pixel 0)
  const uint8_t *base = reinterpret_cast<const uint8_t *>(P.coef) + (long long)img*P.coef_stride;
  const uint8_t *py = base + P.plane_data_off[0] + (long long)y*(P.plane_hblocks[0]*8) + xl;
  const int n = !live ? 0 : P.width - x0 < 8 ? P.width - x0 : 8;
  const uint2 yy = *reinterpret_cast<const uint2 *>(py);          // plane rows are whole blocks
  if (P.nplanes == 1) {
    uint8_t *o = P.out + (long long)img*P.out_stride + (long long)y*P.width + x0;
    if (n == 8 && ((P.width | (int)(P.out_stride & 7)) & 7) == 0 && ((uintptr_t)P.out & 7) == 0) {
      st_nt(reinterpret_cast<uint2 *>(o), yy);
    }
    else {
      for (int i = 0; i < n; i++) o[i] = (uint8_t)(((i < 4 ? yy.x : yy.y) >> (8*(i & 3))) & 255u);
    }
    return;
  }
  // (Cb and Cr carry their own decimation, as in res/unyuv.fs.glsl:6-9, 29-31, 39-41: files whose
  // chroma planes differ are rare, and this kernel is also their colour stage — jga_idct_rgb_batch)
  const uint8_t *pu = base + P.plane_data_off[1] + (long long)(y >> uydec)*(P.plane_hblocks[1]*8)
   + (xl >> uxdec);
  const uint8_t *pv = base + P.plane_data_off[2] + (long long)(y >> vydec)*(P.plane_hblocks[2]*8)
   + (xl >> vxdec);
  uint32_t ub[2] = {0, 0}, vb[2] = {0, 0};                         // 8 >> xdec chroma bytes
  if (uxdec == 0) {
    const uint2 a = *reinterpret_cast<const uint2 *>(pu);
    ub[0] = a.x; ub[1] = a.y;
  }
  else if (uxdec == 1) ub[0] = *reinterpret_cast<const uint32_t *>(pu);
  else ub[0] = *reinterpret_cast<const uint16_t *>(pu);
  if (vxdec == 0) {
    const uint2 b = *reinterpret_cast<const uint2 *>(pv);
    vb[0] = b.x; vb[1] = b.y;
  }
  else if (vxdec == 1) vb[0] = *reinterpret_cast<const uint32_t *>(pv);
  else vb[0] = *reinterpret_cast<const uint16_t *>(pv);
  uint4 a;
  uint2 b2;
  // Cb and Cr decimated alike (every file an encoder writes): the fused kernels' arithmetic —
  // 9 operations per pixel instead of 20, proven equal to the plain form below over the whole
  // input domain (tests/test_rgb_rounding.py) — on yc = Y - 128, u = Cb - 128, v = Cr - 128
  if (uxdec == vxdec && uxdec <= 2) {
    float yc[8], u[8], v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      yc[i] = (float)(((i < 4 ? yy.x : yy.y) >> (8*(i & 3))) & 255u) - 128.0f;
      u[i] = (float)((ub[i >> 2] >> (8*(i & 3))) & 255u) - 128.0f;      // (the first 8 >> xdec are samples)
      v[i] = (float)((vb[i >> 2] >> (8*(i & 3))) & 255u) - 128.0f;
    }
    if (uxdec == 0) { chroma_row<8> cr; cr.set(u, v); rgb_row<0, 8, false>(yc, cr, a, b2); }
    else if (uxdec == 1) { chroma_row<4> cr; cr.set(u, v); rgb_row<1, 4, false>(yc, cr, a, b2); }
    else { chroma_row<2> cr; cr.set(u, v); rgb_row<2, 2, false>(yc, cr, a, b2); }
  }
  else {
  float rgb[24];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int cu = i >> uxdec, cv = i >> vxdec;
    const float Y = (float)(((i < 4 ? yy.x : yy.y) >> (8*(i & 3))) & 255u);
    const float u = (float)((ub[cu >> 2] >> (8*(cu & 3))) & 255u) - 128.0f;
    const float v = (float)((vb[cv >> 2] >> (8*(cv & 3))) & 255u) - 128.0f;
    // clamp + 0.5 + truncation of SURVEY.md A.5; v_cvt_pk_u8_f32 then converts an
    // integer-valued float in [0,255]
    rgb[3*i + 0] = __builtin_floorf(__builtin_fminf(__builtin_fmaxf(Y + 1.402f*v, 0.0f), 255.0f) + 0.5f);
    rgb[3*i + 1] = __builtin_floorf(__builtin_fminf(__builtin_fmaxf((Y + (-0.34414f)*u) + (-0.71414f)*v, 0.0f), 255.0f) + 0.5f);
    rgb[3*i + 2] = __builtin_floorf(__builtin_fminf(__builtin_fmaxf(Y + 1.772f*u, 0.0f), 255.0f) + 0.5f);
  }
  a.x = pack_u8x4(rgb[0], rgb[1], rgb[2], rgb[3]);
  a.y = pack_u8x4(rgb[4], rgb[5], rgb[6], rgb[7]);
  a.z = pack_u8x4(rgb[8], rgb[9], rgb[10], rgb[11]);
  a.w = pack_u8x4(rgb[12], rgb[13], rgb[14], rgb[15]);
  b2.x = pack_u8x4(rgb[16], rgb[17], rgb[18], rgb[19]);
  b2.y = pack_u8x4(rgb[20], rgb[21], rgb[22], rgb[23]);
  }
  uint8_t *o = P.out + (long long)img*P.out_stride + ((long long)y*P.width + x0)*3;
  // staged when the wave's first `nin` lanes (an even number) hold 8 whole pixels each and the
  // others nothing at all, and lane 0's address is 16-byte aligned
  const unsigned long long whole = __builtin_amdgcn_ballot_w64(n == 8);
  const unsigned long long some = __builtin_amdgcn_ballot_w64(n > 0);
  const int nin = (int)__builtin_popcountll(whole);
  const bool wfast = P.out_aligned && whole == some && nin >= 2 && (nin & 1) == 0
   && whole == (nin == 64 ? ~0ull : (1ull << nin) - 1ull)
   && (((uintptr_t)o - (uintptr_t)lane*24u) & 15u) == 0;
  if (__builtin_amdgcn_readfirstlane((int)wfast)) store_rgb_row_wave(stage_mem[wave], lane, o - lane*24, a, b2, nin*3/2);
  else if (live) store_rgb_row(o, a, b2, P.out_aligned && n == 8, x0, P.width);
}

// ---------------------------------------------------------------------------
// Host-side launch table (C++ linkage inside the library; the C-ABI wrappers
// live in device_api.cpp).
// ---------------------------------------------------------------------------
template <int XDEC, int YDEC>
static hipError_t launch_rgb_t(const jga_kparams &P, hipStream_t st) {
  typedef rgb_cfg<XDEC, YDEC> cfg;
  dim3 grid((P.nhmb + cfg::TILE - 1)/cfg::TILE, P.nvmb, P.nimages), block(cfg::THREADS);
  if (P.dequant) hipLaunchKernelGGL((jga_idct_rgb_kernel<XDEC, YDEC, true>), grid, block, 0, st, P);
  else hipLaunchKernelGGL((jga_idct_rgb_kernel<XDEC, YDEC, false>), grid, block, 0, st, P);
  return hipGetLastError();
}

template <int XDEC, int YDEC>
static hipError_t launch_rows_t(const jga_kparams &P, hipStream_t st) {
  typedef rgb_cfg<XDEC, YDEC> cfg;
  dim3 grid((P.nhmb + cfg::TILE - 1)/cfg::TILE, P.nvmb, P.nimages), block(cfg::THREADS);
  if (P.dequant) hipLaunchKernelGGL((jga_idct_rgb_rows_kernel<XDEC, YDEC, true>), grid, block, 0, st, P);
  else hipLaunchKernelGGL((jga_idct_rgb_rows_kernel<XDEC, YDEC, false>), grid, block, 0, st, P);
  return hipGetLastError();
}

extern "C" int jga_launch_rgb(const jga_kparams *P, int xdec, int ydec,
 int staged, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipErrorInvalidValue;
  if (P->nplanes == 1) {
    dim3 grid((P->slots_per_image + 255)/256, P->nimages), block(256);
#define JGA_GREY_LAUNCH(DQ, ST) hipLaunchKernelGGL((jga_idct_grey_kernel<DQ, ST>), grid, block, 0, st, *P)
    if (P->dequant) { if (staged) JGA_GREY_LAUNCH(true, true); else JGA_GREY_LAUNCH(true, false); }
    else { if (staged) JGA_GREY_LAUNCH(false, true); else JGA_GREY_LAUNCH(false, false); }
#undef JGA_GREY_LAUNCH
    e = hipGetLastError();
  }
  else {
    // 4:4:4, 4:2:2, 4:4:0: the row-parallel kernel (JGA_RGB_ROWS=0 selects the tile kernel, for A/B)
    static const bool tile_only = jga_tune("JGA_RGB_ROWS") && atoi(jga_tune("JGA_RGB_ROWS")) == 0;
    if (!tile_only && xdec == 0 && ydec == 0) e = launch_rows_t<0, 0>(*P, st);
    else if (!tile_only && xdec == 1 && ydec == 0) e = launch_rows_t<1, 0>(*P, st);
    else if (!tile_only && xdec == 0 && ydec == 1) e = launch_rows_t<0, 1>(*P, st);
    else if (xdec == 0 && ydec == 0) e = launch_rgb_t<0, 0>(*P, st);
    else if (xdec == 1 && ydec == 0) e = launch_rgb_t<1, 0>(*P, st);
    else if (xdec == 1 && ydec == 1) e = launch_rgb_t<1, 1>(*P, st);
    else if (xdec == 0 && ydec == 1) e = launch_rgb_t<0, 1>(*P, st);
    else if (xdec == 2 && ydec == 0) e = launch_rgb_t<2, 0>(*P, st);
    // the reference takes any sampling factors in {1, 2, 4} (src/xjpeg.c:384-391; pass 3 is generic
    // in xdec / ydec, res/unyuv.fs.glsl:30-31): the rarer ones run the same tile kernel
    else if (xdec == 2 && ydec == 1) e = launch_rgb_t<2, 1>(*P, st);
    else if (xdec == 1 && ydec == 2) e = launch_rgb_t<1, 2>(*P, st);
    else if (xdec == 0 && ydec == 2) e = launch_rgb_t<0, 2>(*P, st);
    else if (xdec == 2 && ydec == 2) e = launch_rgb_t<2, 2>(*P, st);
  }
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int jga_launch_yuv(const jga_kparams *P, int staged, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((P->slots_per_image + 255)/256, P->nimages), block(256);
#define JGA_YUV_LAUNCH(DQ, ST) hipLaunchKernelGGL((jga_idct_yuv_kernel<DQ, ST>), grid, block, 0, st, *P)
  if (P->dequant) { if (staged) JGA_YUV_LAUNCH(true, true); else JGA_YUV_LAUNCH(true, false); }
  else { if (staged) JGA_YUV_LAUNCH(false, true); else JGA_YUV_LAUNCH(false, false); }
#undef JGA_YUV_LAUNCH
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int jga_launch_yuv_rgb(const jga_kparams *P, int uxdec, int uydec, int vxdec, int vydec,
 void *stream) {
  dim3 grid((P->width + 2047)/2048, P->height, P->nimages), block(256);
  hipLaunchKernelGGL(jga_yuv_rgb_kernel, grid, block, 0, (hipStream_t)stream, *P, uxdec, uydec, vxdec, vydec);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
