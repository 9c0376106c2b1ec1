// huff_common.h — GPU-parallel baseline-JPEG entropy decode: shared core.
//
// SURVEY.md §8(f)-1 / BASELINE config 5: the reference decodes the scan serially
// on the host (src/xjpeg.c:449-632; restart handling 593-629).  Here the scan is
// unstuffed on the host and cut into fixed-size SUBSEQUENCES; one GPU lane decodes one
// subsequence.  Huffman streams self-synchronise, so a lane that starts at an
// arbitrary bit soon falls into step with the true symbol sequence:
//
//   state  = (bit position p in the clean scan, next coefficient index k, block slot c in MCU)
//   S[0]   = known (segment start, k = 0, c = 0); restart markers make more
//            segments, each with a known S[0] (xjpeg.c:612-618)
//   round 0: lane i decodes from a GUESS at the start of subsequence i to the
//            first symbol boundary at/after its end, and proposes S[i+1]
//   round r: lane i re-decodes from S[i]; if it reproduces S[i+1] nothing moves.
//            Rounds repeat until no entry changes.  A fixed point is exact by
//            induction (S[0] true, S[i+1] = decode(S[i])) — self-synchronisation
//            only makes it arrive in 2-3 rounds instead of n.
//   then   : per-subsequence block counts are prefix-summed per segment, and a last
//            pass writes every coefficient to its slot of the packed coefficient planes
//            (SURVEY.md Appendix B).  DC prediction (xjpeg.c:480) is NOT part of the
//            synchronisation: the write pass leaves every block's DC DIFFERENCE in a
//            compact array in scan order, a prefix sum over it per restart interval and
//            component gives the DC values (hj_dc_scan), and they go into the planes —
//            or straight into the block-decode kernels, which take them as an input.
//            (Round 3: carrying three DC sums through every synchronisation run cost the
//            rounds a third of their per-symbol instructions, 0.3 of 2.4 ms per 48 x 4K.)
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
#define HJ_SUB_LOG2_MAX 7          /* subsequence length in clean scan bytes: 32, 64 or 128, */
#define HJ_SUB_LOG2_MIN 5          /* a per-batch run-time value (hj_choose_sub_log2) */
#define HJ_SUB_BYTES_MAX (1 << HJ_SUB_LOG2_MAX)
#define HJ_MAX_SLOTS 10           /* blocks per MCU (4:1:1 / 4:2:0 = 6) */

// The Huffman tables of one image in device form: a two-level lookup that never
// leaves LDS.  Level 1 is indexed by the next 9 bits and holds, for codes of up to 9
// bits, a 16-bit ENTRY
//     tot | adv << 5 | s << 12         tot  = code length + magnitude bits: what the symbol takes (1..31)
//                                      adv  = how far the zig-zag index moves: 1 for a DC code,
//                                             run + 1 for an AC code, 64 for EOB (always ends the block)
//                                      s    = magnitude bits that follow the code (symbol & 15)
// so that the decode loops need no per-symbol case analysis: k' = k + adv and the block is
// complete iff k' >= 64.  For a longer code level 1 holds tot = 0 and n + 1 in the adv/s bits:
// the code's bits 9..15 select an entry of level-2 block n (128 entries).  Bit patterns that
// are no code decode as a 17-bit EOB (AC) / zero difference (DC): they occur on
// trajectories that started out of step (harmless) or in corrupt data, where the final
// pass reports them (len > 16), like the host stage's "invalid code" errors.  With SIMT divergence a rarely taken slow path is taken by
// every wave, so it has to be as cheap as the fast one.
//
// AC entries are 32 bits wide: the low half is the ENTRY above; the high half is a PACK — what
// the next 9 bits hold when they hold MORE than one whole AC symbol (code + magnitude bits each;
// at ~3.5 bits per symbol in photographic data that is the usual case):
//     bits | adv << 4 | prefix << 11            bits   = bits all its symbols take (<= 9), 0 = no pack
//                                               adv    = their zig-zag advance together; an EOB,
//                                                        allowed as the LAST symbol only, counts 64
//                                               prefix = the advance of all but the last symbol (<= 31)
// A run may take the whole pack in one step iff k + prefix < 64, i.e. iff no symbol but the
// last completes the block (the symbol after a completed block is a DC code from another
// table), and iff the pack ends at or before the run's stop bit (the loops ask for nine bits
// of room, whatever the pack takes) — then it is, bit for bit and
// for ANY input, what the symbols decoded one by one would have been.  Synchronisation runs only want the end state and the counts, so they decode
// two to three symbols per step; the write pass, which needs every value, uses the low half.
// A frame may use two DC and two AC tables (luma / chroma, what every encoder emits): the
// component -> table choice is hj_image::comp_tbl.  10 KB in all, as before the packs.
#define HJ_L2_BLOCKS 16
struct hj_tables {
  uint16_t dc[2][1 << HJ_FAST_BITS];
  uint32_t ac[2][1 << HJ_FAST_BITS];
  uint16_t l2[HJ_L2_BLOCKS*128];
};

// Everything a lane needs to know about one image.
struct hj_image {
  uint32_t scan_off;                 // offset of this image's clean scan bytes in the batch buffer
  uint32_t scan_len;                 // clean length
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
  uint8_t comp_tbl[3];               // bit 0: which hj_tables::dc the component uses, bit 1: which ::ac
  int64_t comp_coef_off[3];          // plane base in the image's coefficient buffer (shorts)
};

struct hj_segment {                  // one restart interval (or the whole scan)
  uint32_t start, end;               // byte range inside the image's CLEAN scan (unstuffed, marker-free)
  uint32_t sub0;                     // first subsequence, image-local index
  uint32_t nsub;
  uint32_t mcu0;                     // first MCU of the interval
  uint32_t nmcu;
};

// Subsequence length of a batch (JGA_HUFF_SUB overrides).  128 bytes, except for a small batch of
// frames without subsampling.  What a lone frame waits for is the serial distance its
// unluckiest run needs to fall into step with the true symbol sequence: for 4:2:0 ~1-1.5 KB,
// because the MCU slot has to line up as well as bit and block boundaries, and that distance
// costs the same time whether it is walked as 12 runs of 128 bytes or 48 of 32, while every
// hand-over adds its own overhead — 1080p 4:2:0: 0.68 ms at 128 B, 0.80 at 64 B; one 4K frame:
// 0.75 / 0.88 / 1.18 ms at 128 / 64 / 32 B.  Grey and 4:4:4 streams fall into step within a
// few dozen bytes (97-99 % of the runs inside 128 B), so there the shorter runs win as long as
// the GPU has lanes to spare: one 4K 4:4:4 frame 0.40 -> 0.33 ms, one grey 4K 0.44 -> 0.37, one
// 1080p 4:4:4 0.40 -> 0.30; four 4K 4:4:4 frames are back at 0.50 vs 0.54
// (profiles/r2_sub_size_lone_frames.txt).
// Frames WITH restart intervals: a run falls into step at the next interval at the latest, so the
// serial distance is bounded whatever the subsequence length and the shorter runs just put more
// lanes on a small batch — one 1080p frame with an interval per MCU row 0.50 -> 0.41 ms, one 4K
// 0.55 -> 0.48, the 8K frame of BASELINE config 5 0.70 -> 0.64, four 4K frames 0.60 -> 0.57
// (profiles/r3_entropy_stage_steps.md).
#define HJ_SMALL_BATCH_BYTES (8u << 20)
#define HJ_SMALL_BATCH_BYTES_RESTARTS (16u << 20)
HJ_HD int hj_choose_sub_log2(uint64_t scan_bytes, int nslots, int restart_interval = 0) {
  // (round 6, tools/policy_sweep.py: 4:2:2 and 4:4:0 frames — four blocks per MCU — fall into step as quickly as
  // frames without subsampling: a lone 1080p 4:2:2 frame 0.250 -> 0.197 ms, 4K 0.270 -> 0.235, eight 1080p 0.305 -> 0.253)
  if (nslots <= 4 && scan_bytes <= HJ_SMALL_BATCH_BYTES) return HJ_SUB_LOG2_MAX - 1;
  if (restart_interval > 0 && scan_bytes <= HJ_SMALL_BATCH_BYTES_RESTARTS) return HJ_SUB_LOG2_MAX - 1;
  return HJ_SUB_LOG2_MAX;
}

// state word: p (bit position inside the image's clean scan) << 16 | c << 8 | k
HJ_HD uint64_t hj_pack(uint64_t p, int c, int k) { return (p << 16) | ((uint64_t)c << 8) | (uint64_t)k; }
HJ_HD uint64_t hj_pos(uint64_t s) { return s >> 16; }
HJ_HD int hj_slot(uint64_t s) { return (int)((s >> 8) & 255); }
HJ_HD int hj_k(uint64_t s) { return (int)(s & 255); }

// Per-subsequence result of a decode run.
struct hj_run {
  uint64_t end_state;                // state at the first symbol boundary >= stop bit
  uint32_t nblocks;                  // blocks completed in the run
};

// Bit source over plain memory (host emulation): 32 bits, MSB first, starting at bit p.
struct hj_mem_src {
  const uint8_t *s;
  HJ_HD uint32_t window32(uint32_t p) const {
    const uint8_t *b = s + (p >> 3);
    const uint64_t v = ((uint64_t)b[0] << 32) | ((uint64_t)b[1] << 24) | ((uint64_t)b[2] << 16)
     | ((uint64_t)b[3] << 8) | (uint64_t)b[4];
    return (uint32_t)(v >> (8 - (p & 7)));
  }
};

// Bit reader over the CLEAN scan bytes of an image: byte stuffing (FF 00) and
// restart markers were removed when the batch was prepared, so a position is
// simply a bit count.  It is STATELESS apart from that position: every symbol
// fetches its own 32-bit window (a code is <= 16 bits, its magnitude <= 15), which
// on the GPU is one two-dword LDS read and a funnel shift — no refill branch and no
// 64-bit buffer to keep in step across divergent lanes.  Segments are byte aligned
// and lie back to back; bits past the end of a segment are never part of a block
// that gets written.
template <class Src>
struct hj_reader {
  Src src;
  uint32_t p, stop;                  // bit positions in the image's clean scan
  HJ_HD void init(const Src &source, uint64_t pos, uint64_t stop_bit) {
    src = source; p = (uint32_t)pos; stop = (uint32_t)stop_bit;
  }
  HJ_HD bool before_stop() const { return p < stop; }
  HJ_HD bool room9() const { return p + 9u <= stop; }                     // nine more bits end at or before the stop
  HJ_HD bool room(int n) const { return p + (uint32_t)n <= stop; }
  HJ_HD uint32_t window() const { return src.window32(p); }
  HJ_HD void skip(int n) { p += (uint32_t)n; }
  HJ_HD uint64_t tell() const { return p; }
};

// Which reader a bit source is read with: the one above unless the source brings its own
// (the LDS source of the kernels does: huff_kernels.hip).
template <class Src, class = void>
struct hj_reader_of { typedef hj_reader<Src> type; };
template <class Src>
struct hj_reader_of<Src, typename Src::has_reader> { typedef typename Src::reader type; };

// Look up the code at the top of window `w` — in DC table tbl & 1 if `isdc`, else in AC table
// tbl >> 1: its entry in the low half, for an AC code the pack (or 0) in the high half.
HJ_HD uint32_t hj_lookup(const hj_tables *T, int isdc, int tbl, uint32_t w) {
  const uint32_t idx = w >> (32 - HJ_FAST_BITS);
  uint32_t e = isdc ? (uint32_t)T->dc[tbl & 1][idx] : T->ac[tbl >> 1][idx];
  if ((e & 31u) == 0u) e = T->l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];   // (a long code has no pack)
  return e;
}
// The same tables as the kernels keep them in LDS: DC entries widened to 32 bits and laid out
// in front of the AC ones, so that a symbol's lookup is ONE 32-bit read from
// (table number << 9 | index) whatever its kind — no DC / AC case in the per-symbol path.
struct hj_ltables {
  uint32_t tab[4][1 << HJ_FAST_BITS];        // dc[0], dc[1], ac[0], ac[1]
  uint16_t l2[HJ_L2_BLOCKS*128];
};
// WIDE packs (round 5), for batches that leave most of the device empty — a lone frame waits for nine runs one
// after the other at the speed of ONE lane, so what counts there is symbols per step, not LDS per workgroup: the AC
// tables are indexed with the next 12 bits and their packs hold what those hold — 1.89 whole symbols per step on the
// bench's frames instead of 1.42 (counted: tools/archive/r5_packsim.py), a quarter fewer steps per run.  32 KB per
// image on top of hj_tables (built by the host, hj_prepare_head), 40 KB in LDS as hj_ltables_wide: one or two
// workgroups per CU instead of three, which is why batches that fill the device keep the 9-bit tables.  A pack is
// taken under the same two conditions as ever (no symbol but the last completes the block; all of it lies before
// the run's stop — the loops now ask for HJ_WIDE_BITS bits of room), so a run ends in the same state whichever
// tables decoded it: the host's walk over unsettled stretches (hj_walk_unsettled, 9-bit tables) and the write pass
// need not know.
#define HJ_WIDE_BITS 12
struct hj_wide_ac {
  uint32_t ac[2][1 << HJ_WIDE_BITS];         // low half: the ENTRY of the first symbol (= hj_tables::ac[..][index >> 3]), high half: the PACK
};
struct hj_ltables_wide {
  uint32_t dc[2][1 << HJ_FAST_BITS];
  uint32_t ac[2][1 << HJ_WIDE_BITS];
  uint16_t l2[HJ_L2_BLOCKS*128];
};
static_assert(sizeof(hj_ltables_wide) == 4u*((2u << HJ_FAST_BITS) + (2u << HJ_WIDE_BITS)) + 2u*HJ_L2_BLOCKS*128u, "dc and ac back to back");
// bits of look-ahead a table's packs may take: what a run asks for before it takes one whole
template <class Tab> struct hj_pack_bits { static constexpr int value = HJ_FAST_BITS; };
template <> struct hj_pack_bits<hj_ltables_wide> { static constexpr int value = HJ_WIDE_BITS; };
HJ_HD uint32_t hj_lookup(const hj_ltables_wide *T, int isdc, int tbl, uint32_t w) {
  // (dc and ac lie back to back: one read from a selected index)
  const uint32_t idx = isdc ? ((uint32_t)(tbl & 1) << HJ_FAST_BITS) | (w >> (32 - HJ_FAST_BITS))
   : (2u << HJ_FAST_BITS) + (((uint32_t)(tbl >> 1) << HJ_WIDE_BITS) | (w >> (32 - HJ_WIDE_BITS)));
  uint32_t e = (&T->dc[0][0])[idx];
  if ((e & 31u) == 0u) e = T->l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
  return e;
}
HJ_HD uint32_t hj_lookup(const hj_ltables *T, int isdc, int tbl, uint32_t w) {
  const uint32_t t = isdc ? (uint32_t)(tbl & 1) : 2u + (uint32_t)(tbl >> 1);
  uint32_t e = (&T->tab[0][0])[(t << HJ_FAST_BITS) | (w >> (32 - HJ_FAST_BITS))];
  if ((e & 31u) == 0u) e = T->l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
  return e;
}
#define HJ_PACK(bits, adv, prefix) ((uint32_t)((bits) | ((adv) << 4) | ((prefix) << 11)))   /* the high half */
#define HJ_P_BITS(e32) ((int)(((e32) >> 16) & 15u))       /* 0: no pack */
#define HJ_P_ADV(e32) ((int)(((e32) >> 20) & 127u))
#define HJ_P_PREFIX(e32) ((int)((e32) >> 27))
#define HJ_ENTRY(len, s, adv1) ((uint16_t)(((len) + (s)) | (((adv1) + 1) << 5) | ((s) << 12)))
#define HJ_ESCAPE(n) ((uint16_t)(((n) + 1) << 5))         /* level-2 block n */
#define HJ_IS_ESCAPE(e) (((e) & 31u) == 0u)
#define HJ_ESCAPE_BLOCK(e) ((int)((e) >> 5) - 1)
#define HJ_E_TOT(e) ((int)((e) & 31u))
#define HJ_E_ADV(e) ((int)(((e) >> 5) & 127u))
#define HJ_E_S(e) ((int)(((e) >> 12) & 15u))
#define HJ_E_LEN(e) (HJ_E_TOT(e) - HJ_E_S(e))

// The `s` magnitude bits that follow a `len`-bit code in window `w`, extended (T.81 F.2.2.1).
HJ_HD int hj_value(uint32_t w, int len, int s) {
  if (!s) return 0;
  const int v = (int)((w << len) >> (32 - s));
  return v - ((v >> (s - 1)) ? 0 : (1 << s) - 1);
}

// Decode from `start` until the first symbol boundary whose bit position is >= stop_bit:
// the synchronisation rounds' run.  Produces the hj_run (end state, blocks completed), no
// coefficient values at all.  `T` = the image's tables (hj_tables on the host, the LDS copy
// hj_ltables on the GPU); the table choice of MCU slot c sits in bits [2c, 2c+1] of a register
// (an indexed private array would live in scratch).  Every lane of a wave executes every
// instruction of a divergent loop, so the per-symbol instruction count is what the rounds cost.
// `last`: the subsequence is the last one of its segment, so `stop_bit` is also where the
// segment's data ends.  A symbol that reaches past it borrows bits of the next segment (or the
// pad): a block it would complete does not count — the data ended early, as the host stage says
// of such a stream (entropy.c, xjpeg.c:593-629).
// LITE: only the end state is wanted (r.nblocks is left 0) — the very first run of a
// subsequence starts from a GUESS (bit 0 of the subsequence, slot 0), so everything but where it
// ends is meaningless.
// The table choice of every MCU slot, two bits each (a kernel works it out once, not per run: the
// loop is a chain of dependent LDS reads).
HJ_HD uint32_t hj_slot_tables(const hj_image &im) {
  uint32_t bits = 0;
  for (int q = 0; q < im.nslots; q++) bits |= (uint32_t)im.comp_tbl[im.slot_comp[q]] << (2*q);
  return bits;
}
// The lookup with the table NUMBER already chosen (0, 1: the DC tables, 2, 3: the AC tables).
HJ_HD uint32_t hj_lookup_t(const hj_tables *T, uint32_t t, uint32_t w) {
  const uint32_t idx = w >> (32 - HJ_FAST_BITS);
  uint32_t e = t < 2u ? (uint32_t)T->dc[t][idx] : T->ac[t - 2u][idx];
  if ((e & 31u) == 0u) e = T->l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
  return e;
}
HJ_HD uint32_t hj_lookup_t(const hj_ltables *T, uint32_t t, uint32_t w) {
  uint32_t e = (&T->tab[0][0])[(t << HJ_FAST_BITS) | (w >> (32 - HJ_FAST_BITS))];
  if ((e & 31u) == 0u) e = T->l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
  return e;
}
HJ_HD uint32_t hj_lookup_t(const hj_ltables_wide *T, uint32_t t, uint32_t w) {
  const uint32_t idx = t < 2u ? (t << HJ_FAST_BITS) | (w >> (32 - HJ_FAST_BITS))
   : (2u << HJ_FAST_BITS) + (((t - 2u) << HJ_WIDE_BITS) | (w >> (32 - HJ_WIDE_BITS)));
  uint32_t e = (&T->dc[0][0])[idx];
  if ((e & 31u) == 0u) e = T->l2[(((e >> 5) - 1u) << 7) | ((w >> 16) & 127u)];
  return e;
}
// The table numbers of every MCU slot as two words of 2-bit fields — the DC table's number (0, 1) and the AC table's
// (2, 3) of slot c at bits [2c, 2c + 1] — so that a symbol's table is one select and one bit-field extract (round 5:
// seven vector instructions of the run's ~41 per step were spent on getting it out of hj_slot_tables' bits).
struct hj_slot_words { uint32_t dc, ac; };
HJ_HD hj_slot_words hj_slot_table_words(const hj_image &im) {
  hj_slot_words W;
  W.dc = 0; W.ac = 0;
  for (int q = 0; q < im.nslots; q++) {
    const uint32_t two = im.comp_tbl[im.slot_comp[q]];
    W.dc |= (two & 1u) << (2*q);
    W.ac |= (2u | ((two >> 1) & 1u)) << (2*q);
  }
  return W;
}
HJ_HD uint32_t hj_field2(uint32_t word, uint32_t at) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ubfe(word, at, 2u);
#else
  return (word >> at) & 3u;
#endif
}
template <class Src, bool LITE = false, class Tab = hj_tables>
HJ_HD hj_run hj_sync_decode(const Src &src, const hj_image &im, const Tab *T,
 uint64_t start, uint64_t stop_bit, bool last, const hj_slot_words &W) {
  const uint32_t c2end = 2u*(uint32_t)im.nslots;
  typename hj_reader_of<Src>::type br;
  hj_run r;
  int k = hj_k(start);
  uint32_t c2 = 2u*(uint32_t)hj_slot(start);               // twice the MCU slot: where its fields lie in wdc / wac
  uint32_t nblocks = 0;
  const uint32_t wdc = W.dc, wac = W.ac;
  br.init(src, hj_pos(start), stop_bit);
  while (br.before_stop()) {
    const uint32_t w = br.window();
    const uint32_t e = hj_lookup_t(T, hj_field2(k == 0 ? wdc : wac, c2), w);
    // several AC symbols at once, unless one of them (other than the last) ends the block — or
    // the run: a run ends at the FIRST symbol boundary at or past its stop bit whichever way it
    // got there (runs that have fallen into step must hand on identical states), so a pack is
    // only taken whole if all of it lies before the stop
    const bool packed = HJ_P_BITS(e) != 0 && k + HJ_P_PREFIX(e) < 64 && br.room(hj_pack_bits<Tab>::value);
    br.skip(packed ? HJ_P_BITS(e) : HJ_E_TOT(e));
    const int kn = k + (packed ? HJ_P_ADV(e) : HJ_E_ADV(e));   // DC: 1; AC: past the run(s); EOB: >= 64
    const int done = kn >= 64;
    if (!LITE) nblocks += (uint32_t)done;
    c2 = done ? (c2 + 2u == c2end ? 0u : c2 + 2u) : c2;
    k = done ? 0 : kn;
  }
  // (k == 0 with a block counted: the run's final symbol completed it)
  if (last && k == 0 && nblocks > 0 && br.tell() > stop_bit) nblocks--;   // ...with bits the segment does not have
  r.nblocks = nblocks;
  r.end_state = hj_pack(br.tell(), (int)(c2 >> 1), k);
  return r;
}

template <class Src, bool LITE = false, class Tab = hj_tables>
HJ_HD hj_run hj_sync_decode(const Src &src, const hj_image &im, const Tab *T,
 uint64_t start, uint64_t stop_bit, bool last = false) {
  return hj_sync_decode<Src, LITE, Tab>(src, im, T, start, stop_bit, last, hj_slot_table_words(im));
}

// The final pass as the HOST states it (tools/huff_emul.cpp, tests/test_huff_emul.py): every
// coefficient goes through `out`, DC values integrated from the predictors handed in, which are
// handed back as they stand at the end of the run (the emulation walks the lanes of a segment in
// order).  The kernel (huff_kernels.hip: hj_write) decodes the same symbols but leaves the DC
// DIFFERENCES and lets hj_dc_scan integrate them; the GPU parity tests hold the two against the
// same oracle planes.
//   out.put(natural_index, value)       into the lane's block buffer
//   out.flush_complete(waiting, slot, head)   the lane holds a block whose last coefficient was
//       decoded here (`head`: its first one was too, so the buffer holds the whole block)
//   out.flush_partial(slot, head)       end of the run, block unfinished: its coefficients
//       so far (a later lane holds the rest)
template <class Src, class Out>
HJ_HD int hj_write_decode(const Src &src, const hj_image &im, const hj_tables *T,
 const uint8_t *dezz, uint64_t start, uint64_t stop_bit, uint32_t max_blocks,
 int &pred0, int &pred1, int &pred2, Out &out) {
  uint32_t slot_comp_bits = 0, slot_tbl_bits = 0;
  for (int q = 0; q < im.nslots; q++) {
    slot_comp_bits |= (uint32_t)im.slot_comp[q] << (2*q);
    slot_tbl_bits |= (uint32_t)im.comp_tbl[im.slot_comp[q]] << (2*q);
  }
  const int nslots = im.nslots;
  typename hj_reader_of<Src>::type br;
  int k = hj_k(start), c = hj_slot(start), error = 0;
  bool head = k == 0;
  uint32_t n = 0;
  br.init(src, hj_pos(start), stop_bit);
  int comp = (int)((slot_comp_bits >> (2*c)) & 3u);
  int tbl = (int)((slot_tbl_bits >> (2*c)) & 3u);
  while (br.before_stop() && n < max_blocks) {
    const uint32_t w = br.window();
    const int isdc = k == 0;
    const uint32_t e = hj_lookup(T, isdc, tbl, w) & 0xffffu;       // (every value is wanted: no packs)
    const int s = HJ_E_S(e), len = HJ_E_TOT(e) - s, adv1 = HJ_E_ADV(e) - 1;
    int v = hj_value(w, len, s);
    br.skip(len + s);
    if (isdc) {
      pred0 += comp == 0 ? v : 0;
      pred1 += comp == 1 ? v : 0;
      pred2 += comp == 2 ? v : 0;
      v = (int16_t)(comp == 0 ? pred0 : comp == 1 ? pred1 : pred2);    // wraps like xjpeg.c:480
    }
    const int kn = k + adv1 + 1;                           // one past this coefficient's zig-zag index
    if (len > 16) error = 1;                               // a bit pattern that is no code
    if (kn > 64 && adv1 != 63) error = 1;                  // an AC run past coefficient 63
    else if (isdc || s) out.put(dezz[kn - 1], v);
    if (kn >= 64) {                                        // block complete
      out.flush_complete(true, c, head);
      n++;
      c = c + 1 == nslots ? 0 : c + 1;
      comp = (int)((slot_comp_bits >> (2*c)) & 3u);
      tbl = (int)((slot_tbl_bits >> (2*c)) & 3u);
      head = true;
      k = 0;
    }
    else k = kn;
  }
  if (k != 0 && n < max_blocks) out.flush_partial(c, head);     // a later lane finishes this block
  return error;
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
