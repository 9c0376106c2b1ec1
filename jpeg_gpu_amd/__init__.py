"""jpeg_gpu_amd — MI355X-native JPEG block-decode path (host-side Python mirror).

The product is the C-ABI library ``libjpeg_gpu_amd.so`` (include/jpeg_gpu_amd.h):
host entropy stage in C, HIP kernels for gfx950, the ``HIPJPEG_DECODE_CTX_VTBL``
decoder plugin and a pipelined batch decoder.  This package only binds it with
ctypes for tests and bench orchestration; it contains no decode logic and never
imports ``oracle``.  If the library (i.e. the HIP extension) is missing, import
of :mod:`jpeg_gpu_amd.lib` fails loudly — there is no CPU fallback.
"""
from . import abi  # noqa: F401
from .abi import (JPEG_DECODE_PACK, JPEG_DECODE_QUANT, JPEG_DECODE_DCT,  # noqa: F401
                  JPEG_DECODE_YUV, JPEG_DECODE_RGB)

__version__ = "0.1"
