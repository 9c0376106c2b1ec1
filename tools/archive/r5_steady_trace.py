"""Per-group phase times of a long 1080p run (JGA_PIPE_TRACE through the tuning build): where a lane's cycle goes."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, time
sys.path.insert(0, %r)
from jpeg_gpu_amd import abi, lib, synth
w, h, group = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
files = [synth.synthetic_jpeg(w, h, "420", quality=90, seed=1234 + i) for i in range(8)]
n = int(sys.argv[4])
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=group, depth=8)
jobs = lib.Pipeline.make_jobs([files[i %% 8] for i in range(n)])
pl.run_jobs(jobs)
t0 = time.perf_counter(); pl.run_jobs(jobs); dt = time.perf_counter() - t0
print("RATE %%.1f Gpixel/s" %% (n * w * h / dt / 1e9))
pl.close()
''' % ROOT
for w, h, group, n in ((1920, 1080, 32, 16384), (1920, 1080, 16, 16384), (3840, 2160, 32, 4096)):
    env = dict(os.environ, JGA_PIPE_TRACE="1", JGA_LIB_PATH=os.path.join(ROOT, "jpeg_gpu_amd", "libjpeg_gpu_amd_tuning.so"))
    r = subprocess.run([sys.executable, "-c", code, str(w), str(h), str(group), str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    err = r.stderr
    # the second run's lines only: after the second "run: ... jobs in" line
    idx = [m.start() for m in re.finditer(r"run: \d+ jobs in", err)]
    part = err[idx[1]:] if len(idx) > 1 else err
    prep = [float(x) for x in re.findall(r"prepare: host ([0-9.]+) ms", part)]
    grp = re.findall(r"lane group of (\d+): began at ([0-9.]+) ms; prepare \+ wait for a device slot ([0-9.]+) ms \(([0-9.]+) of this thread's CPU\), entropy decode queued ([0-9.]+) ms \(([0-9.]+)\), block decode queued \+ the group's one wait ([0-9.]+) ms", part)
    print("== %dx%d, groups of %d frame equivalents, %d images: %s" % (w, h, group, n, r.stdout.strip()))
    if prep:
        prep.sort(); print("   prepare (host parse + clean-up + wait for the link turn): median %.2f ms, p90 %.2f, n %d" % (prep[len(prep)//2], prep[int(len(prep)*0.9)], len(prep)))
    if grp:
        import statistics as st
        a = [float(g[2]) for g in grp]; c = [float(g[3]) for g in grp]; q = [float(g[4]) for g in grp]; wv = [float(g[6]) for g in grp]
        print("   per group: prepare + device-slot wait %.2f ms (%.2f of the thread's CPU), decode queued %.2f, the wait %.2f; files per group %s" % (st.median(a), st.median(c), st.median(q), st.median(wv), grp[len(grp)//2][0]))
    else:
        print(err[-1500:])
