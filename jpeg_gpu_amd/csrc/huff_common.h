// huff_common.h — GPU-parallel baseline-JPEG entropy decode: shared core.
//
// SURVEY.md §8(f)-1 / BASELINE config 5: the reference decodes the scan serially
// on the host (src/xjpeg.c:449-632; restart handling 593-629).  Here the scan is
// cut into fixed-size SUBSEQUENCES of raw bytes; one GPU lane decodes one
// subsequence.  Huffman streams self-synchronise, so a lane that starts at an
// arbitrary bit soon falls into step with the true symbol sequence:
//
//   state  = (raw bit position p, next coefficient index k, block slot c in MCU)
//   S[0]   = known (segment start, k = 0, c = 0); restart markers make more
//            segments, each with a known S[0] (xjpeg.c:612-618)
//   round 0: lane i decodes from a GUESS at the start of subsequence i to the
//            first symbol boundary at/after its end, and proposes S[i+1]
//   round r: lane i re-decodes from S[i]; if it reproduces S[i+1] nothing moves.
//            Rounds repeat until no entry changes.  A fixed point is exact by
//            induction (S[0] true, S[i+1] = decode(S[i])) — self-synchronisation
//            only makes it arrive in 2-3 rounds instead of n.
//   then   : per-subsequence block counts and DC-difference sums are prefix-
//            summed per segment, and a last pass writes every coefficient to its
//            slot of the packed coefficient planes, DC
//            already integrated (xjpeg.c:480).
//
// Everything here is HOST+DEVICE code: the HIP kernels (huff_kernels.hip) and
// the CPU emulation used by the not-gpu tests (tools/huff_emul.cpp) run the same
// functions.
#ifndef JGA_HUFF_COMMON_H
#define JGA_HUFF_COMMON_H (1)
#include <stdint.h>

#if defined(__HIPCC__)
# define HJ_HD __host__ __device__ __forceinline__
#else
# define HJ_HD inline
#endif

#define HJ_FAST_BITS 9
#define HJ_SUB_BYTES 128          /* subsequence length in raw scan bytes */
#define HJ_MAX_SLOTS 10           /* blocks per MCU (4:1:1 / 4:2:0 = 6) */

// One Huffman table in device form.
struct hj_table {
  uint16_t fast[1 << HJ_FAST_BITS];  // (len << 8) | symbol; 0 = code longer than FAST_BITS
  uint32_t maxcode[18];              // left-aligned 16-bit exclusive bounds per length
  int32_t delta[17];                 // symbol index = code + delta[len]
  uint8_t sym[256];
};

// Everything a lane needs to know about one image.
struct hj_image {
  uint32_t scan_off;                 // offset of this image's scan bytes in the batch buffer
  uint32_t scan_len;
  uint32_t sub0;                     // first subsequence (batch-global index)
  uint32_t nsub;
  uint32_t seg0;                     // first segment (batch-global index)
  uint32_t nseg;
  int32_t nslots;                    // blocks per MCU
  int32_t nhmb;
  int32_t w0_blocks;                 // luma blocks per row (RS = w0_blocks*64 shorts)
  uint8_t slot_comp[HJ_MAX_SLOTS];
  uint8_t slot_sbx[HJ_MAX_SLOTS];
  uint8_t slot_sby[HJ_MAX_SLOTS];
  uint8_t comp_hs[3], comp_vs[3], comp_xdec[3];
  uint8_t pad_[3];
  int64_t comp_coef_off[3];          // plane base in the image's coefficient buffer (shorts)
  // tables: [2*comp] = DC, [2*comp + 1] = AC of that component
};

struct hj_segment {                  // one restart interval (or the whole scan)
  uint32_t start, end;               // raw byte range inside the image's scan
  uint32_t sub0;                     // first subsequence, image-local index
  uint32_t nsub;
  uint32_t mcu0;                     // first MCU of the interval
  uint32_t nmcu;
};

// state word: p (bit position, raw coordinates inside the image's scan) << 16 | c << 8 | k
HJ_HD uint64_t hj_pack(uint64_t p, int c, int k) { return (p << 16) | ((uint64_t)c << 8) | (uint64_t)k; }
HJ_HD uint64_t hj_pos(uint64_t s) { return s >> 16; }
HJ_HD int hj_slot(uint64_t s) { return (int)((s >> 8) & 255); }
HJ_HD int hj_k(uint64_t s) { return (int)(s & 255); }

// Per-subsequence result of a decode run.
struct hj_run {
  uint64_t end_state;                // state at the first symbol boundary >= stop bit
  uint32_t nblocks;                  // blocks completed in the run
  int16_t dcsum[3];                  // sum of DC differences per component (mod 2^16)
  uint16_t error;                    // coefficient index ran past 63 / bad code
};

// MSB-first bit reader over the raw scan bytes: unstuffs FF 00 on the fly, keeps
// enough bookkeeping to report the RAW position of the next unread bit.
struct hj_reader {
  const uint8_t *s;
  uint32_t pos, end;                 // next raw byte to load / end of the segment
  uint64_t bits;                     // MSB-aligned window
  int nbits;
  uint32_t skipped;                  // bit j set: the j-th most recent byte was followed by a skipped 00

  HJ_HD void refill() {
    while (nbits <= 56) {
      uint32_t b = 0xFF;             // beyond the segment: padding ones, position keeps counting
      uint32_t skip = 0;
      if (pos < end) {
        b = s[pos];
        if (b == 0xFF && pos + 1 < end && s[pos + 1] == 0x00) skip = 1;
      }
      pos += 1 + skip;
      skipped = (skipped << 1) | skip;
      bits |= (uint64_t)b << (56 - nbits);
      nbits += 8;
    }
  }
  HJ_HD void init(const uint8_t *scan, uint32_t seg_end, uint64_t p) {
    s = scan;
    end = seg_end;
    pos = (uint32_t)(p >> 3);
    bits = 0;
    nbits = 0;
    skipped = 0;
    refill();
    const int skip = (int)(p & 7);
    bits <<= skip;
    nbits -= skip;
  }
  // raw bit position of the next unread bit
  HJ_HD uint64_t tell() const {
    const int nb = (nbits + 7) >> 3;                       // unread (incl. partly read) bytes
    const uint32_t sk = skipped & ((1u << nb) - 1u);
    const uint32_t byte = pos - (uint32_t)nb - (uint32_t)__builtin_popcount(sk);
    return ((uint64_t)byte << 3) + (uint64_t)(8*nb - nbits);
  }
  HJ_HD uint32_t peek(int n) const { return (uint32_t)(bits >> (64 - n)); }
  HJ_HD void skip(int n) { bits <<= n; nbits -= n; }
};

HJ_HD int hj_symbol(hj_reader &br, const hj_table *t, const uint16_t *fast) {
  const uint32_t e = fast[br.peek(HJ_FAST_BITS)];
  if (e) {
    br.skip((int)(e >> 8));
    return (int)(e & 255);
  }
  const uint32_t code = br.peek(16);
  int len = HJ_FAST_BITS + 1;
  while (len <= 16 && code >= t->maxcode[len]) len++;
  if (len > 16) {                    // not a code (only reachable from a wrong guess / bad data)
    br.skip(16);
    return 0;
  }
  br.skip(len);
  return t->sym[((int)(code >> (16 - len)) + t->delta[len]) & 255];
}

HJ_HD int hj_extend(hj_reader &br, int s) {
  int v = (int)br.peek(s);
  br.skip(s);
  if (v < (1 << (s - 1))) v -= (1 << s) - 1;
  return v;
}

// A sink receives decoded values; the sync/count passes use hj_null_sink.
struct hj_null_sink {
  HJ_HD void block_begin(uint32_t, int) {}
  HJ_HD void dc(int, int) {}
  HJ_HD void ac(int, int) {}
};

// Decode from `start` until the first symbol boundary whose raw bit position is
// >= stop_bit (or until max_blocks blocks are complete).  `tabs[2*comp]`/`[2*comp+1]`
// and the matching `fast` arrays (possibly LDS copies) are the DC/AC tables.
template <class Sink>
HJ_HD hj_run hj_decode(const uint8_t *scan, uint32_t seg_end, const hj_image &im,
 const hj_table *tabs, const uint16_t *const *fast, uint64_t start,
 uint64_t stop_bit, uint32_t max_blocks, Sink &sink) {
  hj_reader br;
  hj_run r;
  int k = hj_k(start), c = hj_slot(start);
  r.nblocks = 0;
  r.dcsum[0] = r.dcsum[1] = r.dcsum[2] = 0;
  r.error = 0;
  br.init(scan, seg_end, hj_pos(start));
  const uint32_t stop_byte = (uint32_t)(stop_bit >> 3);
  sink.block_begin(0, c);            // block 0 of this run may be one a previous lane began
  for (;;) {
    // exact position check only when the window may reach the stop byte
    if (br.pos >= stop_byte && br.tell() >= stop_bit) break;
    if (r.nblocks >= max_blocks) break;
    br.refill();
    const int comp = im.slot_comp[c];
    if (k == 0) {
      const int s = hj_symbol(br, &tabs[2*comp], fast[2*comp]) & 15;
      const int v = s ? hj_extend(br, s) : 0;
      r.dcsum[comp] = (int16_t)(r.dcsum[comp] + v);
      sink.dc(comp, v);
      k = 1;
    }
    else {
      const int rs = hj_symbol(br, &tabs[2*comp + 1], fast[2*comp + 1]);
      if (rs == 0) k = 64;                                   // EOB
      else {
        const int s = rs & 15;
        k += rs >> 4;
        const int v = s ? hj_extend(br, s) : 0;
        if (k > 63) { r.error = 1; k = 63; }
        else if (s) sink.ac(k, v);
        k++;
      }
    }
    if (k >= 64) {
      r.nblocks++;
      c = c + 1 == im.nslots ? 0 : c + 1;
      k = 0;
      sink.block_begin(r.nblocks, c);
    }
  }
  r.end_state = hj_pack(br.tell(), c, k);
  return r;
}

// Offset (shorts) of block `b` of the scan (MCU order) in the image's coefficient buffer:
// inverse of the MCU loop nest + block placement of src/xjpeg.c:461-472, 556-561.
HJ_HD int64_t hj_block_offset(const hj_image &im, uint32_t mcu, int slot) {
  const int comp = im.slot_comp[slot];
  const int my = (int)(mcu / (uint32_t)im.nhmb), mx = (int)(mcu - (uint32_t)my*(uint32_t)im.nhmb);
  const int bx = mx*im.comp_hs[comp] + im.slot_sbx[slot];
  const int by = my*im.comp_vs[comp] + im.slot_sby[slot];
  const int xd = im.comp_xdec[comp];
  const int64_t rs = (int64_t)im.w0_blocks*64;
  return im.comp_coef_off[comp] + rs*(by >> xd) + (rs >> xd)*(by & ((1 << xd) - 1))
   + ((int64_t)bx << 6);
}

#endif
