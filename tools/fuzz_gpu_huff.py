"""Corrupt scan bytes at random and push the files through the GPU entropy stage: it must
return (error or result), never hang; where both stages accept a file, count agreement.
Usage: fuzz_gpu_huff.py [seed] [n] [wide]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jpeg_gpu_amd import lib, synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
stats = {"gpu_err": 0, "host_err": 0, "both_ok_equal": 0, "both_ok_diff": 0, "gpu_ok_host_err": 0, "gpu_err_host_ok": 0}
t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    if len(sys.argv) > 3:             # wide variant: every sampling, larger frames, more restart patterns
        samp = ["420", "444", "grey", "422", "440", "411", (4, 2), (2, 4), (1, 4), ((2, 2), (2, 1), (1, 1)),
                ((4, 1), (1, 1), (2, 1)), ((2, 2), (1, 2), (2, 1))][it % 12]
        ri = [0, 1, 2, 7, -1][it % 5]
        d = bytearray(synth.synthetic_jpeg(300 + (it * 37) % 400, 150 + (it * 23) % 300, samp,
                                           quality=30 + (it * 13) % 66, restart_interval=ri, seed=it))
    else:
        samp = ["420", "444", "grey", "422"][it % 4]
        ri = [0, 3, -1][it % 3]
        d = bytearray(synth.synthetic_jpeg(200 + it % 37, 120 + it % 23, samp, quality=70, restart_interval=ri, seed=it))
    sos = d.find(b"\xff\xda")
    lo = sos + 14
    for _ in range(int(rng.integers(1, 6))):
        pos = int(rng.integers(lo, len(d) - 2))
        mode = int(rng.integers(0, 3))
        if mode == 0: d[pos] = int(rng.integers(0, 256))
        elif mode == 1: d[pos] ^= 1 << int(rng.integers(0, 8))
        else: del d[pos]
    d = bytes(d)
    try:
        _, g = lib.geom_of(d)
    except lib.JgaError:
        continue
    h = None
    try:
        h = lib.entropy_decode(d, g)
    except lib.JgaError:
        stats["host_err"] += 1
    try:
        _, c, _ = lib.gpu_entropy_decode([d], device_unstuff=bool(it & 1))
        if h is None: stats["gpu_ok_host_err"] += 1
        elif np.array_equal(c[0], h): stats["both_ok_equal"] += 1
        else: stats["both_ok_diff"] += 1
    except lib.JgaError:
        stats["gpu_err"] += 1
        if h is not None: stats["gpu_err_host_ok"] += 1
print(stats, "%.1f s" % (time.time() - t0))
