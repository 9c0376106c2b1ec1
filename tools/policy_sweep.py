#!/usr/bin/env python3
"""Does the entropy stage's plan (csrc/huff_api.cpp: choose_rounds, hj_choose_sub_log2) hold off the workloads it was
tuned on?  For every workload — geometry x sampling x batch size x quality x content — the batch is decoded under the
default plan and under every FORCED alternative (tuning build: JGA_HUFF_LIST / ITERS / BY_BLOCK / SUB / NO_WIDE), device
only (scan bytes resident), best of a few repetitions, planes compared with the host stage once per workload; prints one
row per workload and the rows where the default is more than 5 % off the best forced plan.

    JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so python tools/policy_sweep.py [--quick] [--shared] > sweep.txt

--shared: tell the batch that other decodes share the device (a pipeline's lanes: jga_huff_set_device_shared 2).
One process: the knobs are re-read between plans (jga_huff_reload_tuning)."""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("JGA_LIB_PATH", os.path.join(ROOT, "jpeg_gpu_amd", "libjpeg_gpu_amd_tuning.so"))
import numpy as np  # noqa: E402
from jpeg_gpu_amd import lib, synth  # noqa: E402

QUICK = "--quick" in sys.argv
SHARED = "--shared" in sys.argv
GEOMS = [("720p", 1280, 720), ("1080p", 1920, 1080), ("2.7K", 2704, 1520), ("4K", 3840, 2160), ("8K", 7680, 4320)]
SAMPLINGS = ["420", "422", "444", "grey"]
BATCHES = [1, 3, 8, 32, 128]
PIXEL_CAP = 48 * 3840 * 2160                     # the largest batch any lane of the pipeline sees

# name -> environment of the forced plan ({} = the default)
PLANS = [("default", {}),
         ("lists", {"JGA_HUFF_LIST": "1"}), ("no-lists", {"JGA_HUFF_LIST": "0"}),
         ("iters2", {"JGA_HUFF_ITERS": "2,2,6"}), ("iters4", {"JGA_HUFF_ITERS": "4,4,6"}), ("iters6", {"JGA_HUFF_ITERS": "6,6,6"}),
         ("by-block", {"JGA_HUFF_BY_BLOCK": "100000000"}), ("by-sub", {"JGA_HUFF_BY_BLOCK": "0"}),
         ("sub64", {"JGA_HUFF_SUB": "64"}), ("sub128", {"JGA_HUFF_SUB": "128"}),
         ("no-12bit", {"JGA_HUFF_NO_WIDE": "1"})]
KNOBS = sorted({k for _, e in PLANS for k in e})


def workloads():
    out = []
    for gname, w, h in GEOMS:
        for samp in SAMPLINGS:
            for n in BATCHES:
                if n * w * h > PIXEL_CAP:
                    continue
                out.append((gname, w, h, samp, n, 90, "recipe", 0))
        for q in (50, 97):                       # qualities, photograph-like content, restart intervals: 4:2:0 only
            for n in (1, 8, 32):
                if n * w * h <= PIXEL_CAP:
                    out.append((gname, w, h, "420", n, q, "recipe", 0))
        for n in (1, 8, 32):
            if n * w * h <= PIXEL_CAP:
                out.append((gname, w, h, "420", n, 90, "photo", 0))
                out.append((gname, w, h, "420", n, 90, "recipe", -1))
    if QUICK:
        out = [x for x in out if x[0] in ("1080p", "4K") and x[3] in ("420", "444") and x[4] in (1, 8, 32)]
    return out


_files = {}


def files_of(w, h, samp, q, content, ri, n):
    key = (w, h, samp, q, content, ri)
    want = min(n, 4)
    have = _files.setdefault(key, [])
    if len(have) < want:
        def make(i):
            if content == "photo":
                return synth.photo_like_jpeg(w, h, samp, q, ri, seed=1 + i)
            return synth.synthetic_jpeg(w, h, samp, quality=q, restart_interval=ri, seed=1234 + i)
        with ThreadPoolExecutor(max_workers=4) as ex:
            have.extend(ex.map(make, range(len(have), want)))
    return have[:want]


def time_plan(env, jobs, reps, check_against=None):
    """-> (best ms of jga_huff_decode_split, rounds); a fresh batch object per plan (JGA_HUFF_SUB is read at create)."""
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    lib.L.jga_huff_reload_tuning()
    n = len(jobs)
    hb = lib.HuffBatch(n, sum(map(len, jobs)) + 4096 * n)
    if SHARED:
        lib.L.jga_huff_set_device_shared(hb.ptr, 2)
    g = hb.prepare(jobs)
    lib.check(lib.L.jga_stream_sync(None))
    stride = (g.coef_shorts * 2 + 255) // 256 * 128
    dcs = (g.coef_shorts // 64 + 127) // 128 * 128
    d_coef, d_dc = lib.DeviceBuffer(stride * 2 * n), lib.DeviceBuffer(dcs * 2 * n)
    best, rounds = 1e9, 0
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        rounds = hb.decode_split(d_coef.ptr, stride, d_dc.ptr, dcs)
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    ok = None
    if check_against is not None:
        hb.decode(d_coef.ptr, stride)
        got = d_coef.download(g.coef_shorts * 2, dtype=np.int16)
        m = lib.real_coef_mask(g)
        ok = bool(np.array_equal(got[m], check_against(g)[m]))
    hb.close()
    d_coef.free()
    d_dc.free()
    return best * 1e3, rounds, ok


def main():
    if not hasattr(lib.L, "jga_huff_reload_tuning"):
        raise SystemExit("policy_sweep.py: this library has no jga_huff_reload_tuning")
    lib.check(lib.L.jga_set_device(0))
    rows, bad = [], []
    print("# entropy-stage policy sweep (%s), ms per decode, device only; default plan against every forced one"
          % ("device shared" if SHARED else "batch alone on the device"), flush=True)
    print("# workload | default | " + " | ".join(p for p, _ in PLANS[1:]) + " | best | default/best", flush=True)
    for gname, w, h, samp, n, q, content, ri in workloads():
        fs = files_of(w, h, samp, q, content, ri, n)
        jobs = [fs[i % len(fs)] for i in range(n)]
        reps = 3 if n * w * h >= 8 * 3840 * 2160 else 6
        res = {}
        for name, env in PLANS:
            chk = (lambda g: lib.entropy_decode(jobs[0], g)) if name in ("default", "lists", "by-block") else None
            try:
                ms, rounds, ok = time_plan(env, jobs, reps, chk)
            except lib.JgaError as e:
                res[name] = None
                print("#   %s: %s" % (name, str(e)[:100]), flush=True)
                continue
            if ok is False:
                raise SystemExit("policy_sweep.py: plan %s decodes %s wrongly" % (name, (gname, samp, n, q, content, ri)))
            res[name] = ms
        valid = {k: v for k, v in res.items() if v is not None}
        best = min(valid, key=valid.get)
        ratio = res["default"] / valid[best]
        label = "%s %s x%d q%d %s%s" % (gname, samp, n, q, content, " dri" if ri else "")
        print("%-34s | %.3f | %s | %s %.3f | %.3f%s" % (
            label, res["default"], " | ".join("%.3f" % res[p] if res[p] is not None else "-" for p, _ in PLANS[1:]),
            best, valid[best], ratio, "  <-- off by more than 5 %" if ratio > 1.05 else ""), flush=True)
        rows.append((label, res, best, ratio))
        if ratio > 1.05:
            bad.append((label, res["default"], best, valid[best], ratio))
    print("# %d workloads; default within 5 %% of the best forced plan in %d; median default/best %.3f, worst %.3f"
          % (len(rows), len(rows) - len(bad), sorted(r[3] for r in rows)[len(rows) // 2], max(r[3] for r in rows)))
    for label, d, best, b, r in bad:
        print("# OFF: %-34s default %.3f, %s %.3f (x%.3f)" % (label, d, best, b, r))


if __name__ == "__main__":
    main()
