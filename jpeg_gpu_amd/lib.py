"""ctypes binding of libjpeg_gpu_amd.so — every call goes through the C-ABI."""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JGA_LIB_PATH") or os.path.join(_HERE, "libjpeg_gpu_amd.so")   # (override: A/B builds)

# Symbols include/jpeg_gpu_amd.h declares (checked by tests/test_layout_abi.py).
EXPORTED = [
    "HIPJPEG_DECODE_CTX_VTBL", "JGA_LIBJPEG_DECODE_CTX_VTBL", "jga_libjpeg_available", "jga_version", "jga_last_error", "jga_image_init",
    "jga_image_zero", "jga_image_clear", "jga_geom_from_header", "jga_block_offset",
    "jga_parse_header", "jga_band_plan", "jga_band_file", "jga_entropy_decode", "jga_entropy_decode_pack",
    "jga_device_count", "jga_idct_rgb_batch", "jga_idct_yuv_batch", "jga_idct_rgb_batch_dc", "jga_idct_yuv_batch_dc", "jga_huff_decode_split", "jga_kernel_name",
    "jga_index_count", "jga_unpack_batch", "jga_yuv_rgb_batch",
    "jga_device_malloc", "jga_device_free", "jga_host_malloc_pinned",
    "jga_host_free_pinned", "jga_memcpy_h2d", "jga_memcpy_d2h", "jga_upload_staged", "jga_download_staged", "jga_device_memset",
    "jga_stream_sync", "jga_set_device", "jga_device_pci_bus_id", "jga_stream_create", "jga_stream_destroy",
    "jga_time_idct_batch", "jga_pipeline_create", "jga_pipeline_run", "jga_pipeline_plan",
    "jga_pipeline_destroy", "jga_huff_create", "jga_huff_destroy", "jga_huff_prepare",
    "jga_huff_decode", "jga_huff_prepare_verdict", "jga_huff_upload_bytes", "jga_huff_last_rounds", "jga_huff_last_assisted", "jga_huff_image_errors", "jga_huff_image_error", "jga_huff_qtabs",
    "jga_huff_set_threads", "jga_huff_set_device_unstuff", "jga_huff_set_inputs_pinned", "jga_huff_set_blocking_waits", "jga_huff_set_copy_stream", "jga_huff_set_device_shared", "jga_huff_set_upload_gate", "jga_huff_wait_upload",
    "jga_host_register", "jga_host_unregister",
    "jga_pipeline_plan_cfg", "jga_pipeline_register_input", "jga_pipeline_forget_input", "jga_pipeline_counters",
    "jga_huff_set_input_flags", "jga_huff_host_bytes", "jga_huff_set_option", "jga_plugin_configure",
    "jga_time_device_copy", "jga_time_kernel_copy",
    "jga_huff_decode_split_begin", "jga_huff_decode_split_end", "jga_huff_qtabs_device", "jga_huff_set_upload_poll", "jga_huff_image_copied",
]


class JgaError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension has not been built "
            "(run `python -m jpeg_gpu_amd.build`). There is no CPU fallback." % LIB_PATH)
    # If torch is (going to be) in this process its bundled HIP runtime must be
    # the one we bind to; it shares our DT_NEEDED soname (libamdhip64.so.7).
    if "torch" in sys.modules:
        pass
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


L = _load()
os.environ.setdefault("JGA_QUIET", "1")   # errors are raised, not printed

_vp, _i, _ll = C.c_void_p, C.c_int, C.c_longlong
_G = C.POINTER(abi.jga_geom)
L.jga_version.restype = C.c_char_p
L.jga_last_error.restype = C.c_char_p
L.jga_kernel_name.restype = C.c_char_p
L.jga_kernel_name.argtypes = [_G, _i]
L.jga_image_init.argtypes = [C.POINTER(abi.image), C.POINTER(abi.jpeg_header)]
L.jga_image_zero.argtypes = [C.POINTER(abi.image)]
L.jga_image_zero.restype = None
L.jga_image_clear.argtypes = [C.POINTER(abi.image)]
L.jga_image_clear.restype = None
L.jga_geom_from_header.argtypes = [_G, C.POINTER(abi.jpeg_header)]
L.jga_block_offset.argtypes = [_G, _i, _i, _i]
L.jga_block_offset.restype = _ll
L.jga_parse_header.argtypes = [C.c_char_p, _i, C.POINTER(abi.jpeg_header)]
L.jga_entropy_decode.argtypes = [C.c_char_p, _i, _G, _vp, _i]
L.jga_band_plan.argtypes = [C.c_char_p, C.c_long, _i, C.POINTER(abi.jga_band)]
L.jga_band_file.argtypes = [C.c_char_p, C.c_long, C.POINTER(abi.jga_band), _vp, C.c_long]
L.jga_band_file.restype = C.c_long
L.jga_entropy_decode_pack.argtypes = [C.c_char_p, _i, _G, _vp, _ll, _vp,
                                      C.POINTER(_ll), C.POINTER(_ll)]
L.jga_idct_rgb_batch.argtypes = [_G, _i, _vp, _ll, _vp, _i, _vp, _ll, _vp]
L.jga_idct_yuv_batch.argtypes = [_G, _i, _vp, _ll, _vp, _i, _vp, _ll, _vp]
if hasattr(L, "jga_idct_rgb_batch_dc"):          # (absent from older A/B builds given through JGA_LIB_PATH)
    L.jga_idct_rgb_batch_dc.argtypes = [_G, _i, _vp, _ll, _vp, _ll, _vp, _i, _vp, _ll, _vp]
    L.jga_idct_yuv_batch_dc.argtypes = [_G, _i, _vp, _ll, _vp, _ll, _vp, _i, _vp, _ll, _vp]
L.jga_yuv_rgb_batch.argtypes = [_G, _i, _vp, _ll, _vp, _ll, _vp]
L.jga_index_count.argtypes = [_G]
L.jga_index_count.restype = _ll
L.jga_unpack_batch.argtypes = [_G, _i, _vp, _ll, _ll, _vp, _ll, _vp, _ll, _vp]
L.jga_time_idct_batch.argtypes = [_G, _i, _vp, _ll, _vp, _i, _vp, _ll, _i, _i, _vp,
                                  C.POINTER(C.c_float)]
L.jga_device_malloc.argtypes = [C.c_size_t]
L.jga_device_malloc.restype = _vp
L.jga_device_free.argtypes = [_vp]
L.jga_device_free.restype = None
L.jga_host_malloc_pinned.argtypes = [C.c_size_t]
L.jga_host_malloc_pinned.restype = _vp
L.jga_host_free_pinned.argtypes = [_vp]
L.jga_host_free_pinned.restype = None
L.jga_memcpy_h2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
L.jga_memcpy_d2h.argtypes = [_vp, _vp, C.c_size_t, _vp]
L.jga_upload_staged.argtypes = [_vp, _vp, C.c_size_t]
L.jga_download_staged.argtypes = [_vp, _vp, C.c_size_t]
L.jga_device_memset.argtypes = [_vp, _i, C.c_size_t, _vp]
L.jga_stream_sync.argtypes = [_vp]
L.jga_set_device.argtypes = [_i]
L.jga_device_pci_bus_id.argtypes = [_i, C.c_char_p, _i]
L.jga_stream_create.restype = _vp
L.jga_stream_destroy.argtypes = [_vp]
L.jga_stream_destroy.restype = None
L.jga_pipeline_create.argtypes = [C.POINTER(abi.jga_pipeline_config)]
L.jga_pipeline_create.restype = _vp
L.jga_pipeline_run.argtypes = [_vp, C.POINTER(abi.jga_job), _i]
L.jga_pipeline_destroy.argtypes = [_vp]
L.jga_pipeline_destroy.restype = None
L.jga_pipeline_plan.argtypes = [_i, _i, C.POINTER(abi.jga_job), _i, C.POINTER(_i)]
L.jga_pipeline_plan_cfg.argtypes = [C.POINTER(abi.jga_pipeline_config), C.POINTER(abi.jga_job), _i, C.POINTER(_i)]
L.jga_pipeline_register_input.argtypes = [_vp, _vp, _i]
L.jga_pipeline_forget_input.argtypes = [_vp, _vp]
L.jga_pipeline_counters.argtypes = [_vp, C.POINTER(_ll), _i]
L.jga_huff_set_input_flags.argtypes = [_vp, C.c_char_p, _i]
L.jga_huff_set_input_flags.restype = None
L.jga_huff_host_bytes.argtypes = [_vp]
L.jga_huff_host_bytes.restype = _ll
L.jga_huff_set_option.argtypes = [_vp, _i, _i]
L.jga_plugin_configure.argtypes = [C.POINTER(abi.jga_plugin_config)]
L.jga_time_device_copy.argtypes = [_vp, _vp, C.c_size_t, _i, _vp, C.POINTER(C.c_float)]
L.jga_time_kernel_copy.argtypes = [_vp, _vp, C.c_size_t, _i, _i, _vp, C.POINTER(C.c_float)]

L.jga_huff_create.argtypes = [_i, _ll]
L.jga_huff_create.restype = _vp
L.jga_huff_destroy.argtypes = [_vp]
L.jga_huff_destroy.restype = None
L.jga_huff_prepare.argtypes = [_vp, C.POINTER(C.c_char_p), C.POINTER(_i), _i, _G, _vp]
L.jga_huff_decode.argtypes = [_vp, _vp, _ll, _vp]
if hasattr(L, "jga_huff_decode_split"):
    L.jga_huff_decode_split.argtypes = [_vp, _vp, _ll, _vp, _ll, _vp]
L.jga_huff_decode_split_begin.argtypes = [_vp, _vp, _ll, _vp, _ll, _vp]
L.jga_huff_decode_split_end.argtypes = [_vp, C.POINTER(_i)]
L.jga_huff_qtabs_device.argtypes = [_vp]
L.jga_huff_qtabs_device.restype = _vp
L.jga_huff_prepare_verdict.argtypes = [_vp, _i]
L.jga_huff_upload_bytes.argtypes = [_vp]
L.jga_huff_upload_bytes.restype = _ll
L.jga_huff_last_rounds.argtypes = [_vp]
L.jga_huff_last_assisted.argtypes = [_vp]
L.jga_huff_image_errors.argtypes = [_vp]
L.jga_huff_image_error.argtypes = [_vp, C.c_int]
L.jga_huff_set_threads.argtypes = [_vp, _i]
L.jga_huff_set_threads.restype = None
L.jga_huff_set_device_unstuff.argtypes = [_vp, _i]
L.jga_huff_set_device_unstuff.restype = None
L.jga_huff_set_inputs_pinned.argtypes = [_vp, _i]
L.jga_huff_set_inputs_pinned.restype = None
L.jga_huff_set_blocking_waits.argtypes = [_vp, _i]
L.jga_huff_set_blocking_waits.restype = None
if hasattr(L, "jga_huff_set_device_shared"):
    L.jga_huff_set_device_shared.argtypes = [_vp, _i]
    L.jga_huff_set_device_shared.restype = None
if hasattr(L, "jga_huff_set_copy_stream"):
    L.jga_huff_set_copy_stream.argtypes = [_vp, _vp]
    L.jga_huff_set_copy_stream.restype = None
L.jga_host_register.argtypes = [_vp, C.c_size_t]
L.jga_host_unregister.argtypes = [_vp]
L.jga_huff_qtabs.argtypes = [_vp]
L.jga_huff_qtabs.restype = C.POINTER(C.c_ushort)

VTBL = abi.jpeg_decode_ctx_vtbl.in_dll(L, "HIPJPEG_DECODE_CTX_VTBL")
LIBJPEG_VTBL = abi.jpeg_decode_ctx_vtbl.in_dll(L, "JGA_LIBJPEG_DECODE_CTX_VTBL")


def check(rc):
    if rc != 0:
        raise JgaError((L.jga_last_error() or b"?").decode())


def version():
    return L.jga_version().decode()


def device_count():
    return L.jga_device_count()


def device_pci_bus_id(dev):
    buf = C.create_string_buffer(32)
    check(L.jga_device_pci_bus_id(dev, buf, len(buf)))
    return buf.value.decode().lower()


# ---- host stage ---------------------------------------------------------------

def parse_header(data):
    h = abi.jpeg_header()
    check(L.jga_parse_header(bytes(data), len(data), C.byref(h)))
    return h


def band_plan(data, count):
    """The frame `data` in up to `count` bands of MCU rows that decode independently
    (include/jpeg_gpu_amd.h: jga_band_plan): a list of abi.jga_band."""
    data = bytes(data)
    bands = (abi.jga_band * count)()
    n = L.jga_band_plan(data, len(data), count, bands)
    if n < 0:
        raise JgaError(L.jga_last_error().decode())
    return [bands[i] for i in range(n)]


def band_file(data, band):
    """Band `band` of the frame as a JPEG file of its own (jga_band_file)."""
    data = bytes(data)
    n = L.jga_band_file(data, len(data), C.byref(band), None, 0)
    if n < 0:
        raise JgaError(L.jga_last_error().decode())
    out = C.create_string_buffer(n)
    n = L.jga_band_file(data, len(data), C.byref(band), out, n)
    if n < 0:
        raise JgaError(L.jga_last_error().decode())
    return out.raw[:n]


def geom_from_header(h):
    g = abi.jga_geom()
    check(L.jga_geom_from_header(C.byref(g), C.byref(h)))
    return g


def geom_of(data):
    h = parse_header(data)
    return h, geom_from_header(h)


def qtab_of(h):
    """(3,64) uint16: quantisation table of each PLANE, natural order."""
    q = np.zeros((3, 64), np.uint16)
    for p in range(h.ncomps):
        q[p] = np.ctypeslib.as_array(h.comp[p].quant.contents.tbl)
    return q


def entropy_decode(data, g=None, dequant=False, out=None):
    if g is None:
        _, g = geom_of(data)
    if out is None:
        out = np.zeros(g.coef_shorts, np.int16)
    check(L.jga_entropy_decode(bytes(data), len(data), C.byref(g), out.ctypes.data,
                               int(dequant)))
    return out


def entropy_decode_pack(data, g=None):
    if g is None:
        _, g = geom_of(data)
    pack = np.zeros(g.coef_shorts, np.int16)
    nblk = sum((g.plane[i].hblocks << g.plane[i].xdec) * g.plane[i].cstride
               for i in range(g.nplanes))
    index = np.zeros(nblk, np.int32)
    n = C.c_longlong()
    per = (C.c_longlong * 3)()
    check(L.jga_entropy_decode_pack(bytes(data), len(data), C.byref(g), pack.ctypes.data,
                                    pack.size, index.ctypes.data, C.byref(n), per))
    return pack[:n.value].copy(), index, list(per)


# ---- device stage ---------------------------------------------------------------

_NAMED_COPIES = os.environ.get("JGA_TOOLING_NAMED_COPIES") == "1"


class DeviceBuffer:
    """hipMalloc'd bytes owned through the C-ABI helpers."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = L.jga_device_malloc(self.nbytes)
        if not self.ptr:
            raise JgaError((L.jga_last_error() or b"hipMalloc failed").decode())

    def upload(self, arr, offset=0, stream=None):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        # (through the library's pinned bounce buffer: a copy that NAMED this short-lived array would leave the runtime's
        # read-only pinning of its pages cached for whoever gets them next — csrc/device_api.cpp)
        check(L.jga_stream_sync(stream))
        if _NAMED_COPIES:                                    # (A/B of the fault's cause only: tools/sessions/r6_s11.sh)
            check(L.jga_memcpy_h2d(self.ptr + offset, arr.ctypes.data, arr.nbytes, stream))
            check(L.jga_stream_sync(stream))
            return
        check(L.jga_upload_staged(self.ptr + offset, arr.ctypes.data, arr.nbytes))

    def download(self, nbytes=None, offset=0, dtype=np.uint8, stream=None):
        nbytes = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(nbytes, np.uint8)
        check(L.jga_stream_sync(stream))
        if _NAMED_COPIES:
            check(L.jga_memcpy_d2h(out.ctypes.data, self.ptr + offset, nbytes, stream))
            check(L.jga_stream_sync(stream))
            return out.view(dtype)
        check(L.jga_download_staged(out.ctypes.data, self.ptr + offset, nbytes))
        return out.view(dtype)

    def fill(self, value, stream=None):
        check(L.jga_device_memset(self.ptr, value, self.nbytes, stream))
        check(L.jga_stream_sync(stream))

    def free(self):
        if self.ptr:
            L.jga_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _align(n, a=256):
    return (int(n) + a - 1) // a * a


def idct_batch(g, coefs, qtabs, rgb=True, dequant=True):
    """Run the device stage on host arrays (upload -> kernel -> download).

    coefs: (n, coef_shorts) int16; qtabs: (n, 3, 64) uint16.
    Returns (n, rgb_bytes) or (n, yuv_bytes) uint8.
    """
    coefs = np.ascontiguousarray(coefs, np.int16).reshape(-1, g.coef_shorts)
    n = coefs.shape[0]
    qtabs = np.ascontiguousarray(qtabs, np.uint16).reshape(n, 3, 64)
    out_bytes = g.rgb_bytes if rgb else g.yuv_bytes
    cstride = _align(g.coef_shorts * 2) // 2
    ostride = _align(out_bytes)
    d_coef = DeviceBuffer(cstride * 2 * n)
    d_q = DeviceBuffer(qtabs.nbytes)
    d_out = DeviceBuffer(ostride * n)
    try:
        d_out.fill(0xA5)
        for i in range(n):
            d_coef.upload(coefs[i], offset=i * cstride * 2)
        d_q.upload(qtabs)
        fn = L.jga_idct_rgb_batch if rgb else L.jga_idct_yuv_batch
        check(fn(C.byref(g), n, d_coef.ptr, cstride, d_q.ptr, int(dequant), d_out.ptr,
                 ostride, None))
        check(L.jga_stream_sync(None))
        raw = d_out.download().reshape(n, ostride)
        return raw[:, :out_bytes].copy()
    finally:
        d_coef.free()
        d_q.free()
        d_out.free()


def split_planes(g, yuv):
    """Views of the padded planes inside one concatenated YUV buffer."""
    out = []
    for p in range(g.nplanes):
        pl = g.plane[p]
        n = pl.hblocks * pl.vblocks * 64
        out.append(yuv[pl.data_off:pl.data_off + n].reshape(pl.vblocks * 8, pl.hblocks * 8))
    return out


# ---- plugin (vtable) ------------------------------------------------------------

class Decoder:
    """Drives a plugin table (default HIPJPEG_DECODE_CTX_VTBL) exactly as the reference's main() drives a
    decoder (src/jpeg_gpu.c:612-613, 637, 701-704, 1215, 1231-1237)."""

    def __init__(self, data, vtbl=None):
        self.vtbl = VTBL if vtbl is None else vtbl      # (LIBJPEG_VTBL: the comparison backend)
        self._data = np.frombuffer(bytes(data), np.uint8).copy()
        self._info = abi.jpeg_info(len(self._data), self._data.ctypes.data)
        self.ctx = self.vtbl.decode_alloc(C.byref(self._info))
        if not self.ctx:
            raise MemoryError("decode_alloc failed: " + (L.jga_last_error() or b"").decode())
        self.header = abi.jpeg_header()
        self.img = None

    def read_header(self):
        check(self.vtbl.decode_header(self.ctx, C.byref(self.header)))
        return self.header

    def init_image(self):
        self.img = abi.image()
        check(L.jga_image_init(C.byref(self.img), C.byref(self.header)))
        L.jga_image_zero(C.byref(self.img))
        return self.img

    def decode(self, out):
        check(self.vtbl.decode_image(self.ctx, C.byref(self.img), out))

    def reset(self, data=None):
        if data is not None:
            self._data = np.frombuffer(bytes(data), np.uint8).copy()
            self._info = abi.jpeg_info(len(self._data), self._data.ctypes.data)
        self.vtbl.decode_reset(self.ctx, C.byref(self._info))

    # views into the image buffers
    def planes(self):
        out = []
        for i in range(self.img.nplanes):
            p = self.img.plane[i]
            a = np.ctypeslib.as_array(C.cast(p.data, C.POINTER(C.c_ubyte)),
                                      (p.height, p.width))
            out.append(a.copy())
        return out

    def pixels(self):
        n = self.img.nplanes
        shape = (self.img.height, self.img.width, 3) if n == 3 else \
            (self.img.height, self.img.width)
        a = np.ctypeslib.as_array(C.cast(self.img.pixels, C.POINTER(C.c_ubyte)), shape)
        return a.copy()

    def coef(self):
        g = geom_from_header(self.header)
        a = np.ctypeslib.as_array(C.cast(self.img.coef, C.POINTER(C.c_short)),
                                  (g.coef_shorts,))
        return a.copy()

    def close(self):
        if self.ctx:                       # the reference's order: decode_free, then image_clear
            self.vtbl.decode_free(self.ctx)
            self.ctx = None
        if self.img is not None:
            L.jga_image_clear(C.byref(self.img))
            self.img = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class PinnedBytes:
    """A copy of `data` in pinned host memory (hipHostMalloc), as a uint8 numpy view: an ingest
    buffer the DMA engine can read directly."""

    def __init__(self, data):
        n = len(data)
        self.ptr = L.jga_host_malloc_pinned(max(n, 1))
        if not self.ptr:
            raise JgaError((L.jga_last_error() or b"hipHostMalloc failed").decode())
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_ubyte)), (n,))
        self.array[:] = np.frombuffer(bytes(data), np.uint8)

    def free(self):
        if self.ptr:
            self.array = None
            L.jga_host_free_pinned(self.ptr)
            self.ptr = None


# ---- pipeline -------------------------------------------------------------------

class Pipeline:
    @staticmethod
    def config(device=0, nthreads=0, out=abi.JPEG_DECODE_RGB, copy_back=False,
               max_coef_shorts=0, max_out_bytes=0, transport=0, batch=0, depth=0, unstuff=0, **more):
        """A jga_pipeline_config; `more` = any of its other fields by name (spin_waits, trace, input_cache_mb,
        input_cache_sight)."""
        cfg = abi.jga_pipeline_config(C.sizeof(abi.jga_pipeline_config), C.sizeof(abi.jga_job),
                                      device, nthreads, depth, out, int(copy_back),
                                      max_coef_shorts, max_out_bytes, int(transport), int(batch),
                                      int(unstuff))
        names = {f[0] for f in abi.jga_pipeline_config._fields_}
        for k, v in more.items():
            if k not in names or k == "reserved_":
                raise TypeError("jga_pipeline_config has no field %r" % k)
            setattr(cfg, k, int(v))
        return cfg

    def __init__(self, *a, **k):
        cfg = self.config(*a, **k)
        self.ptr = L.jga_pipeline_create(C.byref(cfg))
        if not self.ptr:
            raise JgaError((L.jga_last_error() or b"pipeline_create failed").decode())
        self.copy_back = bool(cfg.copy_back)

    def counters(self):
        """{registered, registered_MB, jobs_in_place, jobs_copied, evicted, register_us, ..., stale} since create."""
        v = (_ll * 9)()
        L.jga_pipeline_counters(self.ptr, v, 9)
        return dict(zip(("registered", "registered_MB", "jobs_in_place", "jobs_copied", "evicted", "register_us",
                         "cleanup_on_device", "host_bytes", "stale"), [int(x) for x in v]))

    def register_input(self, array):
        check(L.jga_pipeline_register_input(self.ptr, array.ctypes.data, array.size))

    def forget_input(self, array):
        check(L.jga_pipeline_forget_input(self.ptr, array.ctypes.data))

    @staticmethod
    def make_jobs(jpegs, host_outs=None, dev_outs=None, pinned=False, outs_pinned=False):
        """The jga_job array for a run (built outside any timed region).  The returned object
        keeps the input buffers alive.  pinned: the files lie in pinned memory; outs_pinned: the
        host_outs do (PinnedBytes / PinnedArray)."""
        n = len(jpegs)
        views = {}
        jobs = (abi.jga_job * n)()
        for i, j in enumerate(jpegs):
            v = views.get(id(j))
            if v is None:
                # (a numpy array is used where it lies — e.g. a PinnedBytes buffer; bytes are viewed)
                v = views[id(j)] = j if isinstance(j, np.ndarray) else np.frombuffer(bytes(j), np.uint8)
            jobs[i].jpeg = v.ctypes.data
            jobs[i].size = v.size
            jobs[i].host_out = host_outs[i].ctypes.data if host_outs is not None else None
            jobs[i].dev_out = dev_outs[i] if dev_outs is not None else None
            jobs[i].pinned = int(bool(pinned)) | (2 if outs_pinned and host_outs is not None else 0)
        jobs._keep = (list(views.values()), list(jpegs), host_outs)
        return jobs

    def run_jobs(self, jobs):
        return L.jga_pipeline_run(self.ptr, jobs, len(jobs))

    def run(self, jpegs, host_outs=None, dev_outs=None):
        jobs = self.make_jobs(jpegs, host_outs, dev_outs)
        return self.run_jobs(jobs), jobs

    def close(self):
        if self.ptr:
            L.jga_pipeline_destroy(self.ptr)
            self.ptr = None


# ---- GPU entropy stage ------------------------------------------------------------

class HuffBatch:
    """jga_huff_*: Huffman decode of a same-geometry batch on the GPU."""

    def __init__(self, max_images, max_scan_bytes, device_unstuff=None):
        self.ptr = L.jga_huff_create(max_images, max_scan_bytes)
        if not self.ptr:
            raise JgaError((L.jga_last_error() or b"jga_huff_create failed").decode())
        if device_unstuff is not None:
            L.jga_huff_set_device_unstuff(self.ptr, int(device_unstuff))
        self.n = 0
        self.geom = None

    def set_option(self, option, value):
        check(L.jga_huff_set_option(self.ptr, int(option), int(value)))

    def prepare(self, jpegs, stream=None):
        n = len(jpegs)
        self._keep = [bytes(j) for j in jpegs]
        arr = (C.c_char_p * n)(*self._keep)
        return self._prepare(arr, n, stream)

    def prepare_at(self, addresses, sizes, stream=None):
        """prepare() on files given by address (e.g. PinnedBytes buffers): no copy is made here."""
        n = len(addresses)
        self._keep = None
        self._addr = (C.c_void_p * n)(*[int(a) for a in addresses])
        return self._prepare(C.cast(self._addr, C.POINTER(C.c_char_p)), n, stream, sizes)

    def _prepare(self, arr, n, stream, sizes=None):
        sizes = (C.c_int * n)(*(sizes if sizes is not None else [len(j) for j in self._keep]))
        g = abi.jga_geom()
        check(L.jga_huff_prepare(self.ptr, arr, sizes, n, C.byref(g), stream))
        self.n, self.geom = n, g
        return g

    def qtabs(self):
        p = L.jga_huff_qtabs(self.ptr)
        return np.ctypeslib.as_array(p, (self.n, 3, 64)).copy()

    def decode(self, d_coef_ptr, coef_stride, stream=None):
        check(L.jga_huff_decode(self.ptr, d_coef_ptr, coef_stride, stream))
        return L.jga_huff_last_rounds(self.ptr)

    def decode_split(self, d_coef_ptr, coef_stride, d_dc_ptr, dc_stride, stream=None):
        """Planes with DC differences + the DC values by buffer slot in d_dc (for *_batch_dc)."""
        check(L.jga_huff_decode_split(self.ptr, d_coef_ptr, coef_stride, d_dc_ptr, dc_stride, stream))
        return L.jga_huff_last_rounds(self.ptr)

    def decode_split_begin(self, d_coef_ptr, coef_stride, d_dc_ptr, dc_stride, stream=None):
        """The decode's first half: everything queued on `stream`, nothing waited for."""
        check(L.jga_huff_decode_split_begin(self.ptr, d_coef_ptr, coef_stride, d_dc_ptr, dc_stride, stream))

    def decode_split_end(self):
        """The second half: waits for the stream; -> (sync rounds, did work queued in between see the final planes)."""
        valid = C.c_int(0)
        check(L.jga_huff_decode_split_end(self.ptr, C.byref(valid)))
        return L.jga_huff_last_rounds(self.ptr), bool(valid.value)

    def qtabs_device(self):
        return L.jga_huff_qtabs_device(self.ptr)

    def assisted(self):
        return L.jga_huff_last_assisted(self.ptr)

    def upload_bytes(self):
        return L.jga_huff_upload_bytes(self.ptr)

    def close(self):
        if self.ptr:
            L.jga_huff_destroy(self.ptr)
            self.ptr = None


def block_slots(g):
    """Per plane: (index positions, coefficient offsets in shorts) of its real blocks, raster."""
    out, base = [], 0
    for p in range(g.nplanes):
        pl = g.plane[p]
        by, bx = np.mgrid[0:pl.vblocks, 0:pl.hblocks]
        rs = (g.w0 // 8) * 64
        off = (pl.coef_off + rs * (by >> pl.xdec) + (rs >> pl.xdec) * (by & ((1 << pl.xdec) - 1))
               + bx * 64)
        out.append(((base + by * pl.hblocks + bx).ravel(), off.ravel().astype(np.int64)))
        base += (pl.hblocks << pl.xdec) * pl.cstride
    return out


def real_coef_mask(g):
    """(coef_shorts,) bool: positions of the coefficient buffer that belong to a real block (the
    packed layout has holes between plane rows)."""
    real = np.zeros(g.coef_shorts, bool)
    for _, off in block_slots(g):
        real[off[:, None] + np.arange(64)] = True
    return real


def gpu_unpack(g, packs, indexes, pack_words=None):
    """jga_unpack_batch on same-geometry images -> (n, coef_shorts) int16 (unwritten slots 0)."""
    n = len(packs)
    nidx = int(L.jga_index_count(C.byref(g)))
    pstride = _align(max(len(p) for p in packs) * 2 + 2, 4) // 2
    cstride = _align(g.coef_shorts * 2) // 2
    d_pack, d_idx = DeviceBuffer(max(pstride, 2) * 2 * n), DeviceBuffer(nidx * 4 * n)
    d_coef = DeviceBuffer(cstride * 2 * n)
    try:
        hp = np.zeros((n, pstride), np.uint16)
        hi = np.zeros((n, nidx), np.int32)
        for i in range(n):
            hp[i, :len(packs[i])] = np.asarray(packs[i]).view(np.uint16)
            hi[i, :len(indexes[i])] = indexes[i]
        d_pack.upload(hp)
        d_idx.upload(hi)
        d_coef.fill(0)
        words = pstride if pack_words is None else int(pack_words)
        check(L.jga_unpack_batch(C.byref(g), n, d_pack.ptr, pstride, words, d_idx.ptr, nidx,
                                 d_coef.ptr, cstride, None))
        check(L.jga_stream_sync(None))
        raw = d_coef.download(dtype=np.int16).reshape(n, cstride)
        return raw[:, :g.coef_shorts].copy()
    finally:
        d_pack.free(); d_idx.free(); d_coef.free()


def gpu_entropy_decode(jpegs, device_unstuff=None, shared=0):
    """Decode the scans of same-geometry JPEGs on the GPU -> (geom, (n, coef_shorts) int16, rounds).
    shared=1: the batch is told that other decodes share the device (jga_huff_set_device_shared) — whatever its size
    it then takes the kernels of a batch that fills the device: the list rounds from the second launch on."""
    hb = HuffBatch(len(jpegs), sum(len(j) for j in jpegs) + 4096, device_unstuff)
    try:
        g = hb.prepare(jpegs)
        if shared:
            L.jga_huff_set_device_shared(hb.ptr, 1)
        stride = _align(g.coef_shorts * 2) // 2
        d = DeviceBuffer(stride * 2 * len(jpegs))
        try:
            # poison the planes: every coefficient of every real block must be WRITTEN by the
            # decode (it clears nothing but the blocks it assembles from pieces)
            d.upload(np.full(stride * len(jpegs), 0x5A5A, np.int16))
            rounds = hb.decode(d.ptr, stride)
            gpu_entropy_decode.assisted = hb.assisted()          # (for tests: host-walked subsequences)
            raw = d.download(dtype=np.int16).reshape(len(jpegs), stride)
            out = raw[:, :g.coef_shorts].copy()
            out[:, ~real_coef_mask(g)] = 0                        # layout holes: no block lives there
            return g, out, rounds
        finally:
            d.free()
    finally:
        hb.close()
