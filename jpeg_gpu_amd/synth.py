"""Synthetic baseline-JPEG inputs (ctypes binding of libjga_synth.so).

The reference ships no fixtures; tests and bench.py make their
inputs with csrc/synth_encode.c.  Input generator only: not on the decode path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libjga_synth.so")
if not os.path.exists(_PATH):
    raise ImportError("%s missing: run `python -m jpeg_gpu_amd.build`" % _PATH)
S = C.CDLL(_PATH)

DQT16, NO_JFIF, SPLIT_DHT, FLAT_AC, SWAP_AC = 1, 2, 4, 8, 16
# luma sampling factors (hs, vs) by name
SAMPLING = {"444": (1, 1), "422": (2, 1), "420": (2, 2), "440": (1, 2), "411": (4, 1),
            "grey": (1, 1)}

S.jgs_encode_synthetic.restype = C.c_long
S.jgs_encode_synthetic.argtypes = [C.c_int] * 7 + [C.c_uint, C.c_int, C.c_void_p, C.c_long]
S.jgs_encode_pixels.restype = C.c_long
S.jgs_encode_pixels.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_long]
S.jgs_encode_levels.restype = C.c_long
S.jgs_encode_levels.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_int,
                                                               C.c_void_p, C.c_long]
S.jgs_coef_shorts.restype = C.c_longlong
S.jgs_coef_shorts.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]
S.jgs_quality_tables.argtypes = [C.c_int, C.c_void_p]
S.jgs_synthetic_pixels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint]


S.jgs_set_chroma_factors.argtypes = [C.c_int] * 4


def _samp(sampling, ncomps):
    """(hs, vs) of the luma plane, with the chroma factors of the next frame set to match:
    a name, (hs, vs) — chroma 1x1 — or ((hs, vs), (hs, vs), (hs, vs)) for Y, Cb, Cr."""
    chroma = (1, 1, 1, 1)
    if ncomps == 1:
        luma = (1, 1)
    elif isinstance(sampling, str):
        luma = SAMPLING[sampling]
    elif isinstance(sampling[0], (tuple, list)):
        luma = tuple(sampling[0])
        chroma = tuple(sampling[1]) + tuple(sampling[2])
    else:
        luma = tuple(sampling)
    if S.jgs_set_chroma_factors(*chroma):
        raise ValueError("sampling factors must be 1, 2 or 4")
    return luma


def synthetic_jpeg(width, height, sampling="420", quality=90, restart_interval=0, seed=1234,
                   flags=0):
    """SURVEY.md §8(d) recipe.  restart_interval: MCUs, 0 = none, -1 = one MCU row."""
    ncomps = 1 if sampling == "grey" else 3
    hs, vs = _samp(sampling, ncomps)
    buf = np.empty(width * height * ncomps * (3 if flags & FLAT_AC else 1) + (1 << 16), np.uint8)
    n = S.jgs_encode_synthetic(width, height, ncomps, hs, vs, quality, restart_interval,
                               seed, flags, buf.ctypes.data, buf.size)
    if n <= 0:
        raise ValueError("jgs_encode_synthetic failed: %d" % n)
    return buf[:n].tobytes()


def synthetic_pixels(width, height, ncomps=3, seed=1234):
    px = np.empty((height, width, ncomps) if ncomps == 3 else (height, width), np.uint8)
    S.jgs_synthetic_pixels(px.ctypes.data, width, height, ncomps, seed)
    return px


def encode_pixels(pixels, sampling="420", quality=90, restart_interval=0, flags=0):
    pixels = np.ascontiguousarray(pixels, np.uint8)
    ncomps = 3 if pixels.ndim == 3 else 1
    h, w = pixels.shape[:2]
    hs, vs = _samp(sampling, ncomps)
    buf = np.empty(w * h * ncomps * 2 + (1 << 16), np.uint8)
    n = S.jgs_encode_pixels(pixels.ctypes.data, w, h, ncomps, hs, vs, quality,
                            restart_interval, flags, buf.ctypes.data, buf.size)
    if n <= 0:
        raise ValueError("jgs_encode_pixels failed: %d" % n)
    return buf[:n].tobytes()


def photo_like_pixels(width, height, seed=1):
    """Photograph-like content: three random-phase fields with a power-law spectrum (amplitude
    ~ f^-1.5: large smooth structures, little fine detail — softer than the 1/f of a sharp photograph, which at this
    contrast would cost 0.55 B/px), a common luminance field plus weaker chroma fields, sigma 45 around mid-grey,
    and one grey level of sensor-like grain.  At q90 4:2:0 it compresses to 0.14 B/px at 3840x2160 and 0.17 at
    1920x1080 — where photographs lie (0.1-0.25) — against the SURVEY recipe's 0.37 (its N(0, 12) noise alone is
    3 bits per pixel)."""
    r = np.random.default_rng(seed)
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.rfftfreq(width)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    amp = (f ** -1.5).astype(np.float32)
    amp[0, 0] = 0.0

    def field():
        ph = r.random(amp.shape, dtype=np.float32) * np.float32(2 * np.pi)
        x = np.fft.irfft2(amp * np.exp(1j * ph), s=(height, width)).astype(np.float32)
        return x / x.std()
    y, u, v = field(), field(), field()
    img = np.empty((height, width, 3), np.float32)
    img[..., 0] = 128 + 45 * (y + 0.35 * v)
    img[..., 1] = 128 + 45 * (y - 0.105 * u - 0.175 * v)
    img[..., 2] = 128 + 45 * (y + 0.35 * u)
    img += r.normal(0, 1.0, img.shape).astype(np.float32)
    return np.clip(img + 0.5, 0, 255).astype(np.uint8)


def photo_like_jpeg(width, height, sampling="420", quality=90, restart_interval=0, seed=1):
    px = photo_like_pixels(width, height, seed)
    if sampling == "grey":
        px = np.ascontiguousarray(px[..., 1])
    return encode_pixels(px, sampling, quality, restart_interval)


def coef_shorts(width, height, sampling="420"):
    ncomps = 1 if sampling == "grey" else 3
    hs, vs = _samp(sampling, ncomps)
    return int(S.jgs_coef_shorts(width, height, ncomps, hs, vs, None, None, None))


def quality_tables(quality):
    q = np.zeros((3, 64), np.uint16)
    S.jgs_quality_tables(quality, q.ctypes.data)
    return q


def encode_levels(levels, width, height, sampling="420", qtab=None, restart_interval=0,
                  flags=0):
    """Entropy-code quantised levels given in the packed coefficient layout."""
    ncomps = 1 if sampling == "grey" else 3
    hs, vs = _samp(sampling, ncomps)
    levels = np.ascontiguousarray(levels, np.int16)
    assert levels.size == coef_shorts(width, height, sampling)
    if qtab is None:
        qtab = quality_tables(90)
    qtab = np.ascontiguousarray(qtab, np.uint16).reshape(3, 64)
    buf = np.empty(levels.size * 4 + (1 << 16), np.uint8)
    n = S.jgs_encode_levels(levels.ctypes.data, width, height, ncomps, hs, vs,
                            qtab.ctypes.data, restart_interval, flags, buf.ctypes.data,
                            buf.size)
    if n <= 0:
        raise ValueError("jgs_encode_levels failed: %d" % n)
    return buf[:n].tobytes()
