"""Why do N Python threads of C decode loops scale so badly on the GPU box's host?  Threads vs
forked processes, frames per call, thread counts.  (Diagnosis for bench.py's cpu_baseline.)"""
import multiprocessing as mp
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from jpeg_gpu_amd import synth

W, H = 3840, 2160
jpegs = [synth.synthetic_jpeg(W, H, "420", 90, seed=100 + i) for i in range(8)]
ref = oracle.Reference()
orc = oracle.Oracle()


def ref_loop(i, frames):
    ref.frames_yuv(jpegs[i % 8], frames)


def port_loop(i, frames, _buf={}):
    info = orc.parse(jpegs[0])
    need = sum(info.hblocks[k] * info.vblocks[k] * 64 for k in range(info.ncomps))
    sc, out = np.empty(need, np.uint8), np.empty((H, W, 3), np.uint8)
    for _ in range(frames):
        orc.decode_rgb(jpegs[i % 8], sc, out)


def with_threads(fn, n, frames):
    gate = threading.Barrier(n + 1)
    def body(i):
        gate.wait()
        fn(i, frames)
    ts = [threading.Thread(target=body, args=(i,)) for i in range(n)]
    for t in ts: t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in ts: t.join()
    return n * frames * W * H / (time.perf_counter() - t0) / 1e6


def proc_body(fn, i, frames, gate):
    gate.wait()
    fn(i, frames)


def with_procs(fn, n, frames):
    ctx = mp.get_context("fork")
    gate = ctx.Barrier(n + 1)
    ps = [ctx.Process(target=proc_body, args=(fn, i, frames, gate)) for i in range(n)]
    for p in ps: p.start()
    gate.wait()
    t0 = time.perf_counter()
    for p in ps: p.join()
    return n * frames * W * H / (time.perf_counter() - t0) / 1e6


ncpu = len(os.sched_getaffinity(0))
print("cpus", ncpu)
t0 = time.perf_counter(); ref_loop(0, 3); print("ref single %.1f Mpix/s" % (3 * W * H / (time.perf_counter() - t0) / 1e6))
t0 = time.perf_counter(); port_loop(0, 3); print("port single %.1f Mpix/s" % (3 * W * H / (time.perf_counter() - t0) / 1e6))
for name, fn in (("ref", ref_loop), ("port", port_loop)):
    for n in (16, 64, 128, ncpu):
        for frames in (2, 8):
            a = with_threads(fn, n, frames)
            b = with_procs(fn, n, frames)
            print("%-4s n=%3d frames=%d  threads %8.1f  procs %8.1f Mpix/s" % (name, n, frames, a, b), flush=True)
