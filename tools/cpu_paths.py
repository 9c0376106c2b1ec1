"""CPU decode paths timed on the box's host cores (bench.py `cpu_baseline`, and the per-config
CPU rates of tools/configs_bench.py): north_star's "xjpeg/libjpeg-turbo CPU path timed on the same
box's host cores in the same run (core count stated)".

One frame loop per GRANTED CPU (more loops than the cgroup grants only get the group throttled:
16 loops 2.0 Gpixel/s, 64 loops 1.6, 256 loops 1.5 on a 16-CPU grant), every loop on its own
file, warm, best of `rounds`, the reference's per-frame call order (reset -> header -> decode,
src/jpeg_gpu.c:1231-1237).  Checker-side code: it imports `oracle` (the compiled reference and
the port), which the product never does."""
import os
import threading
import time


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "?"


def all_core_rate(make_worker, threads, frames, rounds, pixels_per_frame):
    """`threads` frame loops side by side, each `frames` frames per round, started together;
    best round of `rounds` (the first one also warms caches and buffers).  make_worker(i)
    returns the loop of thread i as a callable taking the frame count.  -> (pixels/s, seconds)."""
    loops = [make_worker(i) for i in range(threads)]
    closers = [getattr(l, "close", None) for l in loops]
    best = None
    for _ in range(rounds):
        gate = threading.Barrier(threads + 1)
        done = []

        def body(loop):
            gate.wait()
            loop(frames)
            done.append(time.perf_counter())
        ts = [threading.Thread(target=body, args=(l,)) for l in loops]
        for t in ts:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in ts:
            t.join()
        dt = max(done) - t0
        best = dt if best is None or dt < best else best
    for c in closers:
        if c:
            c()
    return threads * frames * pixels_per_frame / best, best


def granted_threads(cpus=None, quota=None):
    """(threads to run, visible CPUs): all of `cpus` (default: the current affinity mask), or the
    cgroup grant when that is smaller."""
    ncpu = len(cpus) if cpus else len(os.sched_getaffinity(0))
    return (max(1, min(ncpu, int(quota + 0.5))) if quota else ncpu), ncpu


def time_paths(jpegs, width, height, threads, frames, rounds, port=False, single=True):
    """Time the CPU paths on `jpegs` (loop i takes file i mod len): the reference's own xjpeg.c +
    dct.c compiled from its sources (oracle/_ref, YUV stage — it has no CPU RGB stage),
    libjpeg-turbo through LIBJPEG_DECODE_CTX_VTBL (RGB), and with `port` the oracle's restatement
    of the whole path (RGB).  Rates in Mpixel/s."""
    import numpy as np
    import oracle
    from jpeg_gpu_amd import abi, lib
    px = width * height
    res = {}

    def entry(rate, dt, one, note):
        e = {"value": round(rate / 1e6, 1), "seconds": round(dt, 3), "note": note}
        if one is not None:
            e["single_core_value"] = round(one / 1e6, 1)
        return e

    if oracle.Reference.available() and hasattr(oracle.Reference().lib, "ref_frames_yuv"):
        ref = oracle.Reference()
        rate, dt = all_core_rate(lambda i: (lambda n, d=jpegs[i % len(jpegs)]: ref.frames_yuv(d, n)),
                                 threads, frames, rounds, px)
        one = None
        if single:
            t0 = time.perf_counter()
            ref.frames_yuv(jpegs[0], 2)
            one = 2 * px / (time.perf_counter() - t0)
        res["reference_xjpeg_yuv"] = entry(
            rate, dt, one, "the reference's xjpeg.c + dct.c compiled unmodified (oracle/_ref): Huffman + "
            "dequantise + float IDCT + clamp into Y/Cb/Cr planes (it has no CPU RGB stage)")
    if lib.L.jga_libjpeg_available():
        def lj_worker(i):
            d = lib.Decoder(jpegs[i % len(jpegs)], lib.LIBJPEG_VTBL)
            d.read_header()
            d.init_image()

            def loop(n):
                for _ in range(n):
                    d.reset()
                    d.read_header()
                    d.decode(abi.JPEG_DECODE_RGB)
            loop.close = d.close                    # image buffers per loop
            return loop
        rate, dt = all_core_rate(lj_worker, threads, frames, rounds, px)
        one = None
        if single:
            d1 = lj_worker(0)
            d1(1)
            t0 = time.perf_counter()
            d1(2)
            one = 2 * px / (time.perf_counter() - t0)
            d1.close()
        res["libjpeg_turbo_rgb"] = entry(
            rate, dt, one, "system libjpeg.so.8 (libjpeg-turbo) through LIBJPEG_DECODE_CTX_VTBL: ISLOW "
            "IDCT, plain upsampling, RGB out (src/jpeg_wrap.c:196-222); a different integer IDCT, so a "
            "speed reference, not the parity oracle")
    else:
        res["libjpeg_turbo_rgb"] = {"value": None, "note": "libjpeg.so.8 not installed on this box"}
    if port:
        orc = oracle.Oracle()
        info = orc.parse(jpegs[0])
        need = sum(info.hblocks[i] * info.vblocks[i] * 64 for i in range(info.ncomps))
        nc = 3 if info.ncomps == 3 else 1

        def port_worker(i):
            sc = np.empty(need, np.uint8)
            out = np.empty((height, width, 3) if nc == 3 else (height, width), np.uint8)
            d = jpegs[i % len(jpegs)]

            def loop(n):
                for _ in range(n):
                    orc.decode_rgb(d, sc, out)
            return loop
        rate, dt = all_core_rate(port_worker, threads, frames, rounds, px)
        one = None
        if single:
            p1 = port_worker(0)
            t0 = time.perf_counter()
            p1(2)
            one = 2 * px / (time.perf_counter() - t0)
        res["oracle_port_rgb"] = entry(rate, dt, one, "oracle/oracle.c restatement of the whole path incl. "
                                       "upsample + RGB")
    return res
