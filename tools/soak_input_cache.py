"""Soak of round 4's host-side routes against the oracle's RGB: the input cache under pressure (a cache too
small for the working set, eight lanes acquiring / registering / evicting side by side, both sight policies,
buffers forgotten and re-registered between runs), short runs (files fetched by the device), mixed pinned / pageable
jobs.  Usage: soak_input_cache.py [rounds] [seed]   (the oracle is the checker here, as in tests/)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from jpeg_gpu_amd import abi, lib, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
orc = oracle.Oracle()
geoms = [(1280, 720, "420"), (960, 540, "444"), (1920, 1080, "420"), (640, 480, "422")]
files, want = [], []
for gi, (w, h, s) in enumerate(geoms):
    for k in range(10):
        d = synth.synthetic_jpeg(w, h, s, quality=int(rng.integers(75, 98)), restart_interval=int(rng.choice([0, 0, -1, 16])),
                                 seed=int(rng.integers(0, 1 << 30)))
        files.append(np.frombuffer(d, np.uint8).copy())
        want.append(orc.decode_rgb(d)[1].reshape(-1))
pins = [lib.PinnedBytes(f.tobytes()) for f in files]
total_mb = sum(f.size for f in files) / 2**20
print("%d files, %.1f MB" % (len(files), total_mb), flush=True)
bad = 0
t0 = time.time()
for cfg in (dict(input_cache_mb=max(2, int(total_mb / 3))), dict(input_cache_mb=max(2, int(total_mb / 3)), input_cache_sight=2),
            dict(input_cache_mb=int(total_mb) + 8, unstuff=2), dict(unstuff=1), dict(input_cache_mb=4, spin_waits=1)):
    pl = lib.Pipeline(device=0, nthreads=16, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=8, depth=8,
                      unstuff=cfg.pop("unstuff", 2), **cfg)
    mism = 0
    for r in range(rounds):
        n = int(rng.choice([3, 17, 60, 200]))
        idx = rng.integers(0, len(files), n)
        pinned_mask = rng.random(n) < 0.25
        outs = [np.zeros(want[i].size, np.uint8) for i in idx]
        jobs = lib.Pipeline.make_jobs([pins[i].array if p else files[i] for i, p in zip(idx, pinned_mask)], host_outs=outs)
        for j, p in zip(jobs, pinned_mask):
            j.pinned = int(p)
        rc = pl.run_jobs(jobs)
        mism += sum(not np.array_equal(o, want[i]) for o, i in zip(outs, idx)) + (rc != 0)
        if r % 7 == 3 and pl.counters()["registered_MB"]:
            for i in rng.integers(0, len(files), 5):
                lib.L.jga_pipeline_forget_input(pl.ptr, files[int(i)].ctypes.data)
    print(cfg, "->", pl.counters(), "mismatches", mism, flush=True)
    pl.close()
    bad += mism
print("%.1f s, %d bad" % (time.time() - t0, bad))
sys.exit(1 if bad else 0)
