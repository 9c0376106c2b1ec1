#!/usr/bin/env python3
"""Static instruction budget of a kernel from the compiler's ISA listing (jpeg_gpu_amd/build/*.s): the kernel's
instructions by class, mapped to the stage of the block decode each class can only come from.  The block-decode
kernels are straight-line code (every loop unrolled, no data-dependent branch on the main path), so the static count
IS the count a wave executes — compare with SQ_INSTS_VALU per wave of the PMC passes.

    python tools/isa_budget.py idct_kernels _Z19jga_idct_rgb_kernelILi1ELi1ELb1EEv11jga_kparams [more symbols]
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# mnemonic prefix -> (class, the stage it belongs to)
CLASSES = [
    (("v_pk_mul_lo_u16",), "dequantise (level x q, int16 wrap)", "dequant"),
    (("v_cvt_f32_i32", "v_cvt_f32_i16", "v_bfe_i32", "v_ashrrev_i32", "v_cvt_f32_u32"), "coefficient -> float (sign extension + convert)", "convert"),
    (("v_mul_f32", "v_pk_mul_f32", "v_mul_legacy"), "multiplies: two scale factors per coefficient + 5 per 1-D transform + 4 per chroma sample", "scale + transform + colour"),
    (("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32"), "adds: 29 per 1-D transform, +0.5 per column, 4 per pixel + 3 per chroma sample in the colour stage", "transform + colour"),
    (("v_floor_f32",), "floor of the column pass (dct.c:118) and of G", "transform + colour"),
    (("v_max3_f32", "v_max_f32"), "max |t| for the (short) wrap test", "wrap test"),
    (("v_med3_f32",), "clamp to [-128, 127] (the level shift's clamp)", "clamp"),
    (("v_cvt_pk_u8_f32",), "float -> u8 with saturation, packed", "pack"),
    (("v_perm_b32", "v_alignbit_b32", "v_alignbyte_b32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_and_b32", "v_or_b32", "v_bfi_b32", "v_bfe_u32"),
     "bit shuffling: RGB interleave, unpacking, address bits", "pack / addresses"),
    (("v_mov_b32", "v_accvgpr", "v_cndmask_b32", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap"), "moves / selects", "bookkeeping"),
    (("v_add_u32", "v_add_co", "v_addc_co", "v_sub_u32", "v_subrev_u32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshl_add_u32", "v_add_lshl_u32",
      "v_add3_u32", "v_mad_i32_i24", "v_mul_i32_i24", "v_lshl_add_u64", "v_ashrrev_i64", "v_lshlrev_b64", "v_mad_i64_i32", "v_sub_co", "v_subb_co", "v_min_", "v_max_i32", "v_max_u32"),
     "integer arithmetic: addresses, indices", "addresses"),
    (("v_cmp", "v_cmpx"), "compares", "bookkeeping"),
    (("ds_",), "LDS reads / writes", "publish / stage"),
    (("global_load", "buffer_load", "flat_load"), "global loads", "load"),
    (("global_store", "buffer_store", "flat_store"), "global stores", "store"),
    (("s_waitcnt", "s_nop", "s_barrier"), "waits / barriers", "sync"),
    (("s_",), "scalar instructions", "scalar"),
]


def classify(m):
    for prefixes, what, stage in CLASSES:
        if any(m.startswith(p) for p in prefixes):
            return what, stage
    return ("other vector" if m.startswith("v_") else "other"), "other"


def kernel_body(path, symbol):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol + ":"))
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return body


def main():
    stem = sys.argv[1]
    path = os.path.join(ROOT, "jpeg_gpu_amd", "build", "%s-hip-amdgcn-amd-amdhsa-gfx950.s" % stem)
    for sym in sys.argv[2:]:
        count = collections.Counter()
        by_mn = collections.defaultdict(collections.Counter)
        for l in kernel_body(path, sym):
            mt = re.match(r"^\s+([a-z_0-9]+)\b", l)
            if not mt or l.lstrip().startswith((";", ".")):
                continue
            m = mt.group(1)
            what, stage = classify(m)
            count[(what, stage)] += 1
            by_mn[(what, stage)][m] += 1
        valu = sum(n for (w, s), n in count.items() if any(k.startswith("v_") for k in by_mn[(w, s)]))
        total = sum(count.values())
        print("== %s: %d instructions, %d of them vector ALU" % (sym, total, valu))
        for (what, stage), n in sorted(count.items(), key=lambda kv: -kv[1]):
            top = ", ".join("%s %d" % kv for kv in by_mn[(what, stage)].most_common(4))
            print("  %5d  %-16s %s   [%s]" % (n, stage, what, top))
        text = "\n".join(kernel_body(path, sym))
        for key in ("vgpr_count", "sgpr_count", "lds_size", "scratch"):
            pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
