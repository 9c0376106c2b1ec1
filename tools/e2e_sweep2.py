"""transport 2 end to end (JPEG bytes in host RAM -> RGB in HBM) over host threads and lanes,
64 distinct files.  Usage: python tools/e2e_sweep2.py [nimages]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from jpeg_gpu_amd import abi, lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
W, H = 3840, 2160
with ThreadPoolExecutor(16) as ex:
    distinct = list(ex.map(lambda s: synth.synthetic_jpeg(W, H, "420", 90, seed=1234 + s), range(64)))
PINNED = os.environ.get("PINNED", "0") == "1"
if PINNED:
    _pins = [lib.PinnedBytes(d) for d in distinct]
    distinct = [p.array for p in _pins]
cfgs = [(48, 6, t) for t in (6, 12, 24, 48, 96, 192)] + [(48, 4, 24), (48, 8, 48), (24, 8, 48), (32, 6, 48), (64, 6, 48), (96, 4, 48)]
if os.environ.get("SWEEP_CFGS"):
    cfgs = [tuple(int(v) for v in c.split(",")) for c in os.environ["SWEEP_CFGS"].split()]
for batch, depth, nthr in cfgs:
    pl = lib.Pipeline(device=0, nthreads=nthr, out=abi.JPEG_DECODE_RGB, transport=2, batch=batch, depth=depth,
                      unstuff=int(os.environ.get('UNSTUFF', '0')))
    cyc = lambda k, o=0: [distinct[(o + i) % 64] for i in range(k)]
    pl.run_jobs(lib.Pipeline.make_jobs(cyc(batch * depth), pinned=PINNED))
    jobs = lib.Pipeline.make_jobs(cyc(n, 3), pinned=PINNED)
    import resource
    def task_times():
        out = {}
        for t in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
                out[t] = (open("/proc/self/task/%s/comm" % t).read().strip(), (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"))
            except OSError:
                pass
        return out
    best = 1e9
    tt0 = task_times()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    for _ in range(3):
        t0 = time.perf_counter(); rc = pl.run_jobs(jobs); best = min(best, time.perf_counter() - t0)
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    if os.environ.get("RUSAGE"):
        tt1 = task_times()
        per = sorted(((tt1[t][1] - tt0.get(t, (None, 0))[1], tt1[t][0], t) for t in tt1), reverse=True)[:5]
        print("   threads that outlive the run (runtime helpers, main), CPU us per image:",
              ", ".join("%s %.0f" % (nm, dt / (3 * n) * 1e6) for dt, nm, t in per), flush=True)
    pl.close()
    cpu = " | host CPU per image: user %.0f us, sys %.0f us, %d vol / %d invol switches per image x100" % (
        (ru1.ru_utime - ru0.ru_utime) / (3 * n) * 1e6, (ru1.ru_stime - ru0.ru_stime) / (3 * n) * 1e6,
        (ru1.ru_nvcsw - ru0.ru_nvcsw) * 100 // (3 * n), (ru1.ru_nivcsw - ru0.ru_nivcsw) * 100 // (3 * n)) \
        if os.environ.get("RUSAGE") else ""
    print("batch %2d lanes %d threads %3d: %6.1f ms for %d images = %6.1f Gpixel/s (rc %d)%s" % (
        batch, depth, nthr, best * 1e3, n, n * W * H / best / 1e9, rc, cpu), flush=True)
