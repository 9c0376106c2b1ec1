"""Generate the committed golden vectors from the COMPILED REFERENCE.

Run in the build container (needs /root/reference, via oracle/_ref, and Pillow):

    python tests/golden/make_golden.py          # the three files below
    python tests/golden/make_golden.py rare     # jpegs_rare.npz + jpegs_mcu18.npz only

Writes (all small, committed):
  idct_blocks.npz   4096 input blocks + the reference glj_real_idct8x8 output
                    (src/dct.c:100-121), in five value regimes
  jpegs.npz         tiny baseline JPEGs (Pillow/libjpeg-turbo made AND made by
                    our synthetic writer) with, from the reference xjpeg decoder:
                    QUANT planes, DCT planes, Y/Cb/Cr planes, PACK words + index
  layout.json       image_init results for the BASELINE.json geometries
                    
  jpegs_rare.npz    (rare) the sampling factors the reference accepts beyond the usual six
                    (src/xjpeg.c:384-391: any of 1, 2, 4): luma 4x2, 2x4, 1x4, and Cb / Cr
                    decimated differently from each other — same arrays
                    as jpegs.npz, loaded together with it by tests/conftest.py
  jpegs_mcu18.npz   (rare) luma 4x4: 18 blocks per MCU, more than T.81 B.2.3 allows and
                    libjpeg takes, but the reference decodes it
The files are DATA (inputs and expected outputs); no reference source is stored.
"""
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from jpeg_gpu_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def lcg_blocks(n, lo, hi, seed):
    """IEEE-1180 style LCG (test/dct.c:70-81): x = x*1103515245 + 12345."""
    x = seed
    out = np.empty(n * 64, np.int64)
    span = hi - lo + 1
    for i in range(n * 64):
        x = (x * 1103515245 + 12345) & 0xFFFFFFFF
        v = (x & 0x7FFFFFFE) / float(0x7FFFFFFF)
        out[i] = int(v * span) + lo
    return out.reshape(n, 64).astype(np.int16)


def reference_outputs(R, files):
    store = {}
    for name, data in files.items():
        info, quant = R.decode(data, oracle.QUANT)
        _, dct = R.decode(data, oracle.DCT)
        _, planes = R.decode(data, oracle.YUV)
        _, (pack, index) = R.decode(data, oracle.PACK)
        store[name + ".jpg"] = np.frombuffer(data, np.uint8)
        store[name + ".quant"] = quant
        store[name + ".dct"] = dct
        store[name + ".pack"] = pack
        store[name + ".index"] = index
        for i, p in enumerate(planes):
            store[name + ".plane%d" % i] = p
        store[name + ".info"] = np.frombuffer(json.dumps(info.as_dict()).encode(), np.uint8)
    return store


def rare():
    R = oracle.Reference()
    files = {
        "synth_4x2_72x40_q85": synth.synthetic_jpeg(72, 40, (4, 2), quality=85, seed=21),
        "synth_4x2_dri2_35x33": synth.synthetic_jpeg(35, 33, (4, 2), quality=70, restart_interval=2, seed=22),
        "synth_2x4_40x72_q85": synth.synthetic_jpeg(40, 72, (2, 4), quality=85, seed=23),
        "synth_2x4_dri_row_50x70": synth.synthetic_jpeg(50, 70, (2, 4), quality=60, restart_interval=-1, seed=24),
        "synth_1x4_24x64_q90": synth.synthetic_jpeg(24, 64, (1, 4), quality=90, seed=25),
        "synth_1x4_dqt16_17x37": synth.synthetic_jpeg(17, 37, (1, 4), quality=50, seed=26, flags=synth.DQT16),
        # Cb and Cr decimated differently (res/unyuv.fs.glsl has u_xdec/u_ydec and v_xdec/v_ydec)
        "synth_y2x2_cb2x1_cr1x1_70x50": synth.synthetic_jpeg(70, 50, ((2, 2), (2, 1), (1, 1)), quality=85, seed=29),
        "synth_y4x1_cb1x1_cr2x1_dri4_90x30": synth.synthetic_jpeg(90, 30, ((4, 1), (1, 1), (2, 1)), quality=70,
                                                                 restart_interval=4, seed=30),
        "synth_y2x2_cb1x2_cr2x1_41x39": synth.synthetic_jpeg(41, 39, ((2, 2), (1, 2), (2, 1)), quality=90, seed=31),
    }
    np.savez_compressed(os.path.join(HERE, "jpegs_rare.npz"), **reference_outputs(R, files))
    files = {
        "synth_4x4_64x64_q85": synth.synthetic_jpeg(64, 64, (4, 4), quality=85, seed=27),
        "synth_4x4_dri3_75x45": synth.synthetic_jpeg(75, 45, (4, 4), quality=75, restart_interval=3, seed=28),
    }
    np.savez_compressed(os.path.join(HERE, "jpegs_mcu18.npz"), **reference_outputs(R, files))
    for f in ("jpegs_rare.npz", "jpegs_mcu18.npz"):
        print("%8d %s" % (os.path.getsize(os.path.join(HERE, f)), f))


def main():
    R = oracle.Reference()
    rng = np.random.default_rng(20260927)
    # ---- blocks -----------------------------------------------------------
    parts = [
        lcg_blocks(1024, -256, 255, 1),
        lcg_blocks(512, -5, 5, 1),
        lcg_blocks(512, -300, 300, 1),
        rng.integers(-2048, 2048, (512, 64)).astype(np.int16),
        rng.integers(-32768, 32768, (512, 64)).astype(np.int16),   # exercises the (short) wrap
    ]
    sparse = np.zeros((1024, 64), np.int16)
    for b in sparse:
        k = rng.integers(1, 12)
        b[rng.integers(0, 64, k)] = rng.integers(-1024, 1025, k)
    parts.append(sparse)
    blocks = np.concatenate(parts)
    np.savez_compressed(os.path.join(HERE, "idct_blocks.npz"), inp=blocks,
                        out=R.idct_blocks(blocks))
    # ---- jpegs -------------------------------------------------------------
    from PIL import Image

    def pil_jpeg(w, h, mode, **kw):
        yy, xx = np.mgrid[0:h, 0:w]
        ch = [np.clip(127 + 80 * np.sin(xx / (37 + 11 * c)) * np.cos(yy / (53 + 7 * c)) +
                      rng.normal(0, 12, (h, w)), 0, 255).astype(np.uint8) for c in range(3)]
        im = Image.fromarray(np.stack(ch, -1)) if mode == "RGB" else Image.fromarray(ch[0])
        b = io.BytesIO()
        im.save(b, "JPEG", **kw)
        return b.getvalue()

    files = {
        "pil_grey_64x48_q90": pil_jpeg(64, 48, "L", quality=90),
        "pil_444_40x24_q90": pil_jpeg(40, 24, "RGB", quality=90, subsampling="4:4:4"),
        "pil_422_50x30_q85": pil_jpeg(50, 30, "RGB", quality=85, subsampling="4:2:2"),
        "pil_420_100x75_q90": pil_jpeg(100, 75, "RGB", quality=90, subsampling="4:2:0"),
        "pil_420_opt_61x47_q60": pil_jpeg(61, 47, "RGB", quality=60, subsampling="4:2:0",
                                          optimize=True),
        "pil_420_dri_96x80_q95": pil_jpeg(96, 80, "RGB", quality=95, subsampling="4:2:0",
                                          restart_marker_rows=1),
        "synth_420_128x64_q90": synth.synthetic_jpeg(128, 64, "420", seed=7),
        "synth_420_dri3_72x40": synth.synthetic_jpeg(72, 40, "420", restart_interval=3, seed=8),
        "synth_422_dqt16_48x32": synth.synthetic_jpeg(48, 32, "422", seed=9, flags=synth.DQT16),
        "synth_411_64x16": synth.synthetic_jpeg(64, 16, "411", seed=10),
        "synth_440_24x48": synth.synthetic_jpeg(24, 48, "440", seed=11),
        "synth_grey_33x17": synth.synthetic_jpeg(33, 17, "grey", seed=12),
    }
    store = reference_outputs(R, files)
    np.savez_compressed(os.path.join(HERE, "jpegs.npz"), **store)
    # ---- layout --------------------------------------------------------------
    geoms = {
        "512x512 grey": (512, 512, [(1, 1)]),
        "1080p 4:2:0": (1920, 1080, [(2, 2), (1, 1), (1, 1)]),
        "1080p 4:2:2": (1920, 1080, [(2, 1), (1, 1), (1, 1)]),
        "4K 4:4:4": (3840, 2160, [(1, 1)] * 3),
        "4K 4:2:0": (3840, 2160, [(2, 2), (1, 1), (1, 1)]),
        "8K 4:2:0": (7680, 4320, [(2, 2), (1, 1), (1, 1)]),
        "100x75 4:2:0": (100, 75, [(2, 2), (1, 1), (1, 1)]),
        "32x24 4:2:0 (test/image.c)": (32, 24, [(2, 2), (1, 1), (1, 1)]),
        "97x33 4:1:1": (97, 33, [(4, 1), (1, 1), (1, 1)]),
        "24x48 4:4:0": (24, 48, [(1, 2), (1, 1), (1, 1)]),
    }
    lay = {k: dict(R.layout(w, h, s).as_dict(), samp=s) for k, (w, h, s) in geoms.items()}
    with open(os.path.join(HERE, "layout.json"), "w") as f:
        json.dump(lay, f, indent=1, sort_keys=True)
    for f in os.listdir(HERE):
        print("%8d %s" % (os.path.getsize(os.path.join(HERE, f)), f))


if __name__ == "__main__":
    rare() if sys.argv[1:] == ["rare"] else main()
