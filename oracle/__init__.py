"""oracle — CPU checkers for the JPEG block-decode path.  TEST INFRASTRUCTURE ONLY.

Two ctypes-wrapped libraries:

* ``liboracle.so``  — our C restatement of the reference's CPU path (oracle.c).
* ``_ref/libjpeggpu_ref.so`` — the reference's own sources compiled by
  oracle/Makefile (present when built in a container that has /root/reference;
  the prebuilt .so travels to the GPU box).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this package.  The product (jpeg_gpu_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

QUANT, DCT, YUV = 1, 2, 3
PACK = 0


class Info(C.Structure):
    """Mirror of orc_info / ref_info (identical layout in both libraries)."""
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("ncomps", C.c_int),
        ("restart_interval", C.c_int), ("nhmb", C.c_int), ("nvmb", C.c_int),
        ("coef_shorts", C.c_longlong),
        ("hsamp", C.c_int * 3), ("vsamp", C.c_int * 3),
        ("hblocks", C.c_int * 3), ("vblocks", C.c_int * 3),
        ("xdec", C.c_int * 3), ("ydec", C.c_int * 3),
        ("cstride", C.c_int * 3), ("tq", C.c_int * 3),
        ("coef_off", C.c_longlong * 3),
        ("quant", (C.c_ushort * 64) * 3),
    ]

    def plane_shape(self, i):
        return (self.vblocks[i] * 8, self.hblocks[i] * 8)

    def qtab(self):
        return np.array([[self.quant[p][k] for k in range(64)] for p in range(3)],
                        dtype=np.uint16)

    def as_dict(self):
        n = self.ncomps
        return dict(width=self.width, height=self.height, ncomps=n,
                    restart_interval=self.restart_interval, nhmb=self.nhmb,
                    nvmb=self.nvmb, coef_shorts=self.coef_shorts,
                    hsamp=list(self.hsamp)[:n], vsamp=list(self.vsamp)[:n],
                    hblocks=list(self.hblocks)[:n], vblocks=list(self.vblocks)[:n],
                    xdec=list(self.xdec)[:n], ydec=list(self.ydec)[:n],
                    cstride=list(self.cstride)[:n],
                    coef_off=list(self.coef_off)[:n])


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference exists, _ref/."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:     # one rank per GPU may call this
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True,
                           stdout=subprocess.DEVNULL if quiet else None)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte)) if a is not None else None


class _Lib:
    def __init__(self, path):
        self.lib = C.CDLL(path)


class Oracle(_Lib):
    """The restatement (oracle.c)."""

    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        super().__init__(path)
        L = self.lib
        L.orc_idct8x8_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.orc_parse.argtypes = [C.c_void_p, C.c_long, C.POINTER(Info), C.POINTER(C.c_char_p)]
        L.orc_decode.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(Info), C.POINTER(C.c_char_p)]
        L.orc_coef_to_planes.argtypes = [C.POINTER(Info), C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_planes_to_rgb.argtypes = [C.POINTER(Info), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.orc_decode_rgb.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p,
                                     C.POINTER(Info)]
        L.orc_constants.argtypes = [C.c_void_p]
        L.orc_dezigzag.argtypes = [C.c_void_p]
        L.orc_unpack_blocks.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_long, C.c_void_p]
        L.orc_unpack_blocks.restype = None

    def constants(self):
        out = np.zeros(12, np.float32)
        self.lib.orc_constants(out.ctypes.data)
        return out

    def dezigzag(self):
        out = np.zeros(64, np.int32)
        self.lib.orc_dezigzag(out.ctypes.data)
        return out

    def idct_blocks(self, blocks):
        blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 64)
        out = np.empty_like(blocks)
        self.lib.orc_idct8x8_blocks(out.ctypes.data, blocks.ctypes.data, len(blocks))
        return out

    def unpack_blocks(self, pack, starts):
        """PACK words + block start indices -> (nblocks, 64) int16, natural order."""
        pack = np.ascontiguousarray(pack).view(np.uint16)
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        out = np.zeros((len(starts), 64), np.int16)
        self.lib.orc_unpack_blocks(pack.ctypes.data, len(pack), starts.ctypes.data, len(starts),
                                   out.ctypes.data)
        return out

    def parse(self, data):
        info = Info()
        err = C.c_char_p()
        buf = bytes(data)
        if self.lib.orc_parse(buf, len(buf), C.byref(info), C.byref(err)):
            raise ValueError((err.value or b"?").decode())
        return info

    def decode(self, data, out):
        """-> (info, coef) for QUANT/DCT, (info, [planes]) for YUV."""
        buf = bytes(data)
        info = self.parse(buf)
        err = C.c_char_p()
        if out in (QUANT, DCT):
            coef = np.zeros(info.coef_shorts, np.int16)
            rc = self.lib.orc_decode(buf, len(buf), out, coef.ctypes.data, None, None, None,
                                     C.byref(info), C.byref(err))
            res = coef
        else:
            planes = [np.zeros(info.plane_shape(i), np.uint8) for i in range(info.ncomps)]
            ptrs = [p.ctypes.data for p in planes] + [None] * (3 - len(planes))
            rc = self.lib.orc_decode(buf, len(buf), out, None, ptrs[0], ptrs[1], ptrs[2],
                                     C.byref(info), C.byref(err))
            res = planes
        if rc:
            raise ValueError((err.value or b"?").decode())
        return info, res

    def coef_to_planes(self, info, coef, dequant=True):
        coef = np.ascontiguousarray(coef, np.int16)
        planes = [np.zeros(info.plane_shape(i), np.uint8) for i in range(info.ncomps)]
        ptrs = [p.ctypes.data for p in planes] + [None] * (3 - len(planes))
        self.lib.orc_coef_to_planes(C.byref(info), coef.ctypes.data, int(dequant), *ptrs)
        return planes

    def planes_to_rgb(self, info, planes):
        n = info.ncomps
        rgb = np.zeros((info.height, info.width, n) if n == 3 else (info.height, info.width),
                       np.uint8)
        planes = [np.ascontiguousarray(p, np.uint8) for p in planes]
        ptrs = [p.ctypes.data for p in planes] + [None] * (3 - len(planes))
        self.lib.orc_planes_to_rgb(C.byref(info), ptrs[0], ptrs[1], ptrs[2], rgb.ctypes.data)
        return rgb

    def decode_rgb(self, data, scratch=None, rgb=None):
        """Whole CPU path (the cpu_baseline 'port')."""
        buf = bytes(data)
        info = self.parse(buf)
        n = info.ncomps
        need = sum(info.hblocks[i] * info.vblocks[i] * 64 for i in range(n))
        if scratch is None or scratch.size < need:
            scratch = np.empty(need, np.uint8)
        if rgb is None:
            rgb = np.empty((info.height, info.width, n) if n == 3 else
                           (info.height, info.width), np.uint8)
        if self.lib.orc_decode_rgb(buf, len(buf), scratch.ctypes.data, rgb.ctypes.data,
                                   C.byref(info)):
            raise ValueError("orc_decode_rgb failed")
        return info, rgb


class Reference(_Lib):
    """The reference's own code (oracle/_ref).  ``Reference.available()`` first."""
    PATH = os.path.join(_HERE, "_ref", "libjpeggpu_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        super().__init__(self.PATH)
        L = self.lib
        L.ref_idct8x8_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.ref_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_longlong), C.POINTER(Info)]
        L.ref_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(Info)]
        if hasattr(L, "ref_frames_yuv"):          # (a _ref built before this entry existed)
            L.ref_frames_yuv.argtypes = [C.c_char_p, C.c_int, C.c_int]

    def frames_yuv(self, data, frames):
        """`frames` passes of the reference's own per-frame loop (reset -> header -> YUV)."""
        if self.lib.ref_frames_yuv(bytes(data), len(data), int(frames)):
            raise ValueError("reference decode failed")

    def idct_blocks(self, blocks):
        blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 64)
        out = np.empty_like(blocks)
        self.lib.ref_idct8x8_blocks(out.ctypes.data, blocks.ctypes.data, len(blocks))
        return out

    def parse(self, data):
        buf = bytes(data)
        info = Info()
        if self.lib.ref_decode(buf, len(buf), -1, None, None, None, None, None, None,
                               C.byref(info)):
            raise ValueError("reference failed to parse header")
        return info

    def decode(self, data, out):
        buf = bytes(data)
        info = self.parse(buf)
        if out in (QUANT, DCT):
            coef = np.zeros(info.coef_shorts, np.int16)
            rc = self.lib.ref_decode(buf, len(buf), out, coef.ctypes.data, None, None, None,
                                     None, None, C.byref(info))
            res = coef
        elif out == PACK:
            coef = np.zeros(info.coef_shorts, np.int16)
            nblk = sum((info.hblocks[i] << info.xdec[i]) * info.cstride[i]
                       for i in range(info.ncomps))
            index = np.zeros(nblk, np.int32)
            nwords = C.c_longlong()
            rc = self.lib.ref_decode(buf, len(buf), out, coef.ctypes.data, None, None, None,
                                     index.ctypes.data, C.byref(nwords), C.byref(info))
            res = (coef[:nwords.value].copy(), index)
        else:
            planes = [np.zeros(info.plane_shape(i), np.uint8) for i in range(info.ncomps)]
            ptrs = [p.ctypes.data for p in planes] + [None] * (3 - len(planes))
            rc = self.lib.ref_decode(buf, len(buf), out, None, ptrs[0], ptrs[1], ptrs[2],
                                     None, None, C.byref(info))
            res = planes
        if rc:
            raise ValueError("reference decode failed")
        return info, res

    def layout(self, width, height, samp):
        n = len(samp)
        hs = (C.c_int * 3)(*([s[0] for s in samp] + [0] * (3 - n)))
        vs = (C.c_int * 3)(*([s[1] for s in samp] + [0] * (3 - n)))
        info = Info()
        if self.lib.ref_layout(width, height, n, hs, vs, C.byref(info)):
            raise ValueError("ref_layout failed")
        return info
