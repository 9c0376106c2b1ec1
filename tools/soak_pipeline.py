"""Soak: random images through the pipeline with every transport, against the oracle's RGB.
Usage: soak_pipeline.py [N] [seed]   (the oracle is the checker here, as in tests/)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from jpeg_gpu_amd import abi, lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
orc = oracle.Oracle()
specs = []
for it in range(n):
    samp = ["420", "444", "grey", "422", "440", "411"][int(rng.integers(0, 6))]
    specs.append((int(rng.integers(1, 900)), int(rng.integers(1, 600)), samp, int(rng.integers(5, 100)),
                  int(rng.choice([0, 0, -1, 3])), int(rng.integers(0, 1 << 30))))
specs += specs[: n // 3]                       # repeats: same-geometry groups for transport 2
datas = [synth.synthetic_jpeg(w, h, s, quality=q, restart_interval=ri, seed=sd) for (w, h, s, q, ri, sd) in specs]
want = [orc.decode_rgb(d)[1].reshape(-1) for d in datas]
t0 = time.time(); bad = 0
pins = [lib.PinnedBytes(d) for d in datas]
for transport, unstuff, pinned in ((0, 0, False), (1, 0, False), (2, 1, False), (2, 2, False), (2, 0, True)):
    outs = [np.zeros(x.size, np.uint8) for x in want]
    pl = lib.Pipeline(device=0, nthreads=8, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=transport,
                      batch=5, depth=3, unstuff=unstuff)
    jobs = lib.Pipeline.make_jobs([p.array for p in pins] if pinned else datas, host_outs=outs, pinned=pinned)
    rc = pl.run_jobs(jobs)
    pl.close()
    mism = sum(not np.array_equal(o, x) for o, x in zip(outs, want))
    print("transport %d unstuff %d pinned %d: rc %d, %d/%d images differ" % (transport, unstuff, pinned, rc, mism, len(datas)))
    bad += mism + (rc != 0)
print("%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
