#!/usr/bin/env python3
"""What registering a caller's pageable buffer costs against the host copy it replaces:
hipHostRegister / hipHostUnregister per buffer, the memcpy into pinned staging, and the DMA rate out of a
registered malloc buffer against one from hipHostMalloc.  Prints a table; run on the GPU box."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import lib          # noqa: E402

lib.check(lib.L.jga_set_device(0))
st = lib.L.jga_stream_create()
for mb in (0.25, 0.77, 3.07, 12.3):
    n = int(mb * 1e6)
    bufs = [np.random.default_rng(i).integers(0, 255, n, dtype=np.uint8) for i in range(24)]
    dev = lib.DeviceBuffer(n)
    pin = lib.PinnedBytes(bytes(n))
    t_reg, t_unreg = [], []
    for b in bufs:
        t0 = time.perf_counter()
        lib.check(lib.L.jga_host_register(b.ctypes.data, n))
        t_reg.append(time.perf_counter() - t0)
    t_dma = []
    for b in bufs:                      # DMA out of registered pageable memory
        t0 = time.perf_counter()
        lib.check(lib.L.jga_memcpy_h2d(dev.ptr, b.ctypes.data, n, st))
        lib.check(lib.L.jga_stream_sync(st))
        t_dma.append(time.perf_counter() - t0)
    for b in bufs:
        t0 = time.perf_counter()
        lib.check(lib.L.jga_host_unregister(b.ctypes.data))
        t_unreg.append(time.perf_counter() - t0)
    t_cpy, t_pin = [], []
    for b in bufs:                      # what it replaces: memcpy into pinned staging (+ the DMA from there)
        t0 = time.perf_counter()
        C.memmove(pin.array.ctypes.data, b.ctypes.data, n)
        t_cpy.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        lib.check(lib.L.jga_memcpy_h2d(dev.ptr, pin.array.ctypes.data, n, st))
        lib.check(lib.L.jga_stream_sync(st))
        t_pin.append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    print("%6.2f MB: register %7.0f us (first %7.0f)  unregister %6.0f us | memcpy to pinned %6.0f us | "
          "DMA registered %6.0f us (%.1f GB/s)  DMA hipHostMalloc %6.0f us (%.1f GB/s)"
          % (mb, med(t_reg[1:]), t_reg[0] * 1e6, med(t_unreg), med(t_cpy), med(t_dma), n / med(t_dma) / 1e3,
             med(t_pin), n / med(t_pin) / 1e3), flush=True)
    dev.free()
    pin.free()
