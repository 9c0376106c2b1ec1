"""ctypes mirror of include/jpeg_gpu_amd.h (the C-ABI boundary).

Field for field the reference's data model (src/image.h:25-51,
src/jpeg_info.h:37-71, src/jpeg_wrap.h:22-51); sizes are asserted against the
x86-64 numbers of SURVEY.md §8b.
"""
import ctypes as C

NCOMPS_MAX = 3
NQUANT_MAX = 4
NPLANES_MAX = 3

(JPEG_SUBSAMP_UNKNOWN, JPEG_SUBSAMP_444, JPEG_SUBSAMP_422, JPEG_SUBSAMP_420,
 JPEG_SUBSAMP_440, JPEG_SUBSAMP_411, JPEG_SUBSAMP_MONO) = range(7)
SUBSAMP_NAMES = ["Unknown", "4:4:4", "4:2:2", "4:2:0", "4:4:0", "4:1:1", "Mono"]

(JPEG_DECODE_PACK, JPEG_DECODE_QUANT, JPEG_DECODE_DCT, JPEG_DECODE_YUV,
 JPEG_DECODE_RGB) = range(5)
DECODE_OUT_NAMES = ["pack", "quant", "dct", "yuv", "rgb"]


class jpeg_quant(C.Structure):
    _fields_ = [("valid", C.c_int), ("bits", C.c_ubyte), ("tbl", C.c_ushort * 64)]


class jpeg_component(C.Structure):
    _fields_ = [("hblocks", C.c_int), ("vblocks", C.c_int), ("hsamp", C.c_int),
                ("vsamp", C.c_int), ("quant", C.POINTER(jpeg_quant))]


class jpeg_header(C.Structure):
    _fields_ = [("bits", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("ncomps", C.c_int), ("subsamp", C.c_int), ("restart_interval", C.c_int),
                ("comp", jpeg_component * NCOMPS_MAX), ("quant", jpeg_quant * NQUANT_MAX)]


class jpeg_info(C.Structure):
    _fields_ = [("size", C.c_int), ("buf", C.c_void_p)]


class image_plane(C.Structure):
    _fields_ = [("bitdepth", C.c_int), ("xdec", C.c_ubyte), ("ydec", C.c_ubyte),
                ("xstride", C.c_int), ("ystride", C.c_int), ("width", C.c_ushort),
                ("height", C.c_ushort), ("data", C.c_void_p), ("coef", C.c_void_p),
                ("cstride", C.c_int), ("packed", C.c_int), ("index", C.c_void_p)]


class image(C.Structure):
    _fields_ = [("width", C.c_ushort), ("height", C.c_ushort), ("nplanes", C.c_int),
                ("plane", image_plane * NPLANES_MAX), ("coef", C.c_void_p),
                ("packed", C.c_int), ("index", C.c_void_p), ("pixels", C.c_void_p)]


alloc_func = C.CFUNCTYPE(C.c_void_p, C.POINTER(jpeg_info))
header_func = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(jpeg_header))
image_func = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(image), C.c_int)
reset_func = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(jpeg_info))
free_func = C.CFUNCTYPE(None, C.c_void_p)


class jpeg_decode_ctx_vtbl(C.Structure):
    _fields_ = [("decode_alloc", alloc_func), ("decode_header", header_func),
                ("decode_image", image_func), ("decode_reset", reset_func),
                ("decode_free", free_func)]


class jga_plane_geom(C.Structure):
    _fields_ = [("hblocks", C.c_int), ("vblocks", C.c_int), ("xdec", C.c_int),
                ("ydec", C.c_int), ("cstride", C.c_int), ("qidx", C.c_int),
                ("coef_off", C.c_longlong), ("data_off", C.c_longlong)]


class jga_geom(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("nplanes", C.c_int),
                ("subsamp", C.c_int), ("w0", C.c_int), ("nhmb", C.c_int), ("nvmb", C.c_int),
                ("restart_interval", C.c_int), ("coef_shorts", C.c_longlong),
                ("coef_blocks", C.c_longlong), ("yuv_bytes", C.c_longlong),
                ("rgb_bytes", C.c_longlong), ("plane", jga_plane_geom * NPLANES_MAX)]


class jga_pipeline_config(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("job_size", C.c_int),
                ("device", C.c_int), ("nthreads", C.c_int), ("depth", C.c_int),
                ("out", C.c_int), ("copy_back", C.c_int),
                ("max_coef_shorts", C.c_longlong), ("max_out_bytes", C.c_longlong),
                ("transport", C.c_int), ("batch", C.c_int), ("unstuff", C.c_int),
                ("spin_waits", C.c_int), ("trace", C.c_int),
                ("input_cache_mb", C.c_int), ("input_cache_sight", C.c_int), ("reserved_", C.c_int * 6)]


class jga_job(C.Structure):
    _fields_ = [("jpeg", C.c_void_p), ("size", C.c_int), ("host_out", C.c_void_p),
                ("dev_out", C.c_void_p), ("status", C.c_int), ("width", C.c_int),
                ("height", C.c_int), ("nplanes", C.c_int), ("h2d_bytes", C.c_longlong),
                ("pinned", C.c_int), ("host_bytes", C.c_longlong)]


class jga_plugin_config(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("register_buffers", C.c_int), ("host_entropy", C.c_int),
                ("copy_team", C.c_int), ("reserved_", C.c_int)]


class jga_band(C.Structure):
    _fields_ = [("index", C.c_int), ("count", C.c_int), ("mcu_row0", C.c_int), ("mcu_rows", C.c_int),
                ("y0", C.c_int), ("rows", C.c_int), ("first_interval", C.c_int), ("reserved_", C.c_int),
                ("scan_off", C.c_long), ("scan_bytes", C.c_long)]


JGA_HUFF_OPT_SUB_BYTES, JGA_HUFF_OPT_ASSIST_AFTER, JGA_HUFF_OPT_SPECULATE, JGA_HUFF_OPT_TRACE = 1, 2, 3, 5


# SURVEY.md §8b ABI numbers (x86-64 SysV)
ABI_SIZES = {image_plane: 56, image: 208, jpeg_quant: 136, jpeg_component: 24,
             jpeg_header: 640, jpeg_info: 16, jpeg_decode_ctx_vtbl: 40}
for _t, _n in ABI_SIZES.items():
    assert C.sizeof(_t) == _n, (_t, C.sizeof(_t), _n)
