"""Soak: random sizes / samplings / qualities / restart intervals through the GPU entropy stage
(batches of mixed content, same geometry) against the host stage.  Usage: soak_gpu_huff.py [N] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jpeg_gpu_amd import lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); bad = 0; blocks = 0; irregular = 0
for it in range(n):
    samp = ["420", "444", "grey", "422", "440", "411"][int(rng.integers(0, 6))]
    w, h = int(rng.integers(1, 1400)), int(rng.integers(1, 900))
    ri = int(rng.choice([0, 0, -1, 1, 2, 7, 33]))
    nb = int(rng.integers(1, 5))
    # header variants (16-bit DQT, no JFIF, one DHT per table) and flat-AC content (long runs of
    # identical short blocks: the streams that do not self-synchronise)
    flags = int(rng.choice([0, 0, synth.DQT16, synth.NO_JFIF, synth.SPLIT_DHT, synth.FLAT_AC,
                            synth.FLAT_AC | synth.SPLIT_DHT]))
    datas = [synth.synthetic_jpeg(w, h, samp, quality=int(rng.integers(5, 100)), restart_interval=ri,
                                  seed=int(rng.integers(0, 1 << 30)), flags=flags) for _ in range(nb)]
    try:
        g, coefs, _ = lib.gpu_entropy_decode(datas)
    except lib.JgaError as e:
        if "too irregular" in str(e):             # such tables take the host stage (by design)
            irregular += 1
            continue
        raise
    for d, c in zip(datas, coefs):
        if not np.array_equal(c, lib.entropy_decode(d, g)):
            bad += 1
            print("MISMATCH", samp, w, h, ri)
    blocks += nb * g.coef_blocks
print("soak: %d batches (%d with tables for the host stage), %d blocks, %d mismatches, %.1f s" % (n, irregular, blocks, bad, time.time() - t0))
sys.exit(1 if bad else 0)
