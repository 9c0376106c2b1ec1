"""Every BASELINE.json config on one MI355X, each beside its CPU path (bench.py's `configs` object;
run alone: `python tools/configs_bench.py` prints the same object, one config per line).

For each config (BASELINE.json `configs` 2-5; the headline 4K 4:2:0 entry is filled in by bench.py
from its own legs):
  to_rgb_hbm      JPEG file bytes in host RAM -> RGB8 in HBM through the product's batch decoder
                  (jga_pipeline, transport 2): `latency_ms` of ONE frame with nothing else in flight
                  (best of a few), and `Mpixel_s` of a stream of such frames (config 4: the batch
                  itself, all 1024 files on one GPU, and rank 3's 128-file shard of an 8-GPU job)
  to_host_pixels  the plugin's semantics (src/jpeg_gpu.c:1231-1237: reset -> header -> decode_image(RGB),
                  pixels copied back into img->pixels): ms per frame and Mpixel/s
  device          scan bytes resident in HBM -> GPU Huffman + fused kernel (no host, no PCIe): ms; the
                  fused kernel alone on resident planes: ms, GB/s of algorithmic bytes, fraction of 8 TB/s
  bit_exact_vs_oracle   every distinct file's pipeline output against the oracle's pixels
  cpu             the reference's xjpeg + dct.c (compiled from its sources) and libjpeg-turbo on the same
                  file(s) and the same granted cores, Mpixel/s
"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0


def _kernel_alone(lib, np, jpegs, n, reps):
    """Coefficient planes of n images resident -> RGB, HIP events on the launch stream."""
    hdr, g = lib.geom_of(jpegs[0])
    cs = (g.coef_shorts * 2 + 255) // 256 * 128
    os_ = (g.rgb_bytes + 255) // 256 * 256
    dc, do, dq = lib.DeviceBuffer(cs * 2 * n), lib.DeviceBuffer(os_ * n), lib.DeviceBuffer(384 * n)
    coefs = [lib.entropy_decode(j, g) for j in jpegs[:min(len(jpegs), 6)]]
    for i in range(n):
        dc.upload(coefs[i % len(coefs)], offset=i * cs * 2)
    dq.upload(np.tile(lib.qtab_of(hdr).reshape(-1), n))
    ms = C.c_float()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:                       # clocks
        lib.check(lib.L.jga_time_idct_batch(C.byref(g), n, dc.ptr, cs, dq.ptr, 1, do.ptr, os_, 1, 5, None, C.byref(ms)))
    lib.check(lib.L.jga_time_idct_batch(C.byref(g), n, dc.ptr, cs, dq.ptr, 1, do.ptr, os_, 1, reps, None, C.byref(ms)))
    dc.free(); do.free(); dq.free()
    ab = n * (g.coef_blocks * 128 + g.rgb_bytes)
    return {"kernel": lib.L.jga_kernel_name(C.byref(g), 1).decode(), "kernel_ms": round(ms.value, 4),
            "kernel_GBps": round(ab / ms.value / 1e6, 1),
            "kernel_hbm_frac": round(ab / ms.value / 1e6 / HBM_PEAK_GBPS, 4), "images_per_launch": n}


def _device_only(lib, jpegs, n, reps=5):
    """Scan bytes resident in HBM -> GPU Huffman -> fused kernel -> RGB in HBM, best of reps."""
    _, g = lib.geom_of(jpegs[0])
    cs = (g.coef_shorts * 2 + 255) // 256 * 128
    os_ = (g.rgb_bytes + 255) // 256 * 256
    dc, do, dq = lib.DeviceBuffer(cs * 2 * n), lib.DeviceBuffer(os_ * n), lib.DeviceBuffer(384 * n)
    jobs = [jpegs[i % len(jpegs)] for i in range(n)]
    hb = lib.HuffBatch(n, sum(map(len, jobs)) + 4096 * n)
    hb.prepare(jobs)
    lib.check(lib.L.jga_stream_sync(None))
    dq.upload(hb.qtabs())
    best, rounds = 1e9, 0
    dcs = (g.coef_shorts // 64 + 127) // 128 * 128
    ddc = lib.DeviceBuffer(dcs * 2 * n)
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        rounds = hb.decode_split(dc.ptr, cs, ddc.ptr, dcs)      # (as the pipeline does: DC values beside the planes)
        lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), n, dc.ptr, cs, ddc.ptr, dcs, dq.ptr, 1, do.ptr, os_, None))
        lib.check(lib.L.jga_stream_sync(None))
        best = min(best, time.perf_counter() - t0)
    hb.close()
    dc.free(); do.free(); dq.free(); ddc.free()
    return {"ms": round(best * 1e3, 3), "images": n, "sync_rounds": rounds,
            "Mpixel_s": round(n * g.width * g.height / best / 1e6, 1)}


def _device_split(lib, jpegs, n, reps=5):
    """As _device_only for a batch of n, the two stages timed apart (host clock around synchronous calls, mean of
    reps after one warm-up): the GPU entropy stage (jga_huff_decode_split) and the fused block-decode kernel."""
    _, g = lib.geom_of(jpegs[0])
    cs = (g.coef_shorts * 2 + 255) // 256 * 128
    os_ = (g.rgb_bytes + 255) // 256 * 256
    dc, do = lib.DeviceBuffer(cs * 2 * n), lib.DeviceBuffer(os_ * n)
    jobs = [jpegs[i % len(jpegs)] for i in range(n)]
    hb = lib.HuffBatch(n, sum(map(len, jobs)) + 4096 * n)
    hb.prepare(jobs)
    lib.check(lib.L.jga_stream_sync(None))
    dcs = (g.coef_shorts // 64 + 127) // 128 * 128
    ddc = lib.DeviceBuffer(dcs * 2 * n)
    dq = lib.DeviceBuffer(384 * n)
    dq.upload(hb.qtabs())
    th = ti = 0.0
    rounds = 0
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        rounds = hb.decode_split(dc.ptr, cs, ddc.ptr, dcs)
        t1 = time.perf_counter()
        lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), n, dc.ptr, cs, ddc.ptr, dcs, dq.ptr, 1, do.ptr, os_, None))
        lib.check(lib.L.jga_stream_sync(None))
        t2 = time.perf_counter()
        if rep:
            th += t1 - t0
            ti += t2 - t1
    hb.close()
    dc.free(); do.free(); dq.free(); ddc.free()
    return {"images": n, "huffman_ms": round(th / reps * 1e3, 3), "idct_rgb_ms": round(ti / reps * 1e3, 3),
            "sync_rounds": rounds, "scan_MB": round(sum(map(len, jobs)) / 1e6, 1)}


def _settle(lib):
    """Before a leg that times single milliseconds: let the leg before it finish dying.  Freeing a job's buffers (6 GB
    of outputs after config 4's 1024-file job) leaves the driver work that runs on for ~0.1-0.2 s and shares the copy
    engines: the 128-file shard timed right behind it measures 4.3-4.4 ms median, after 0.3 s of nothing 3.6
    (tools/archive/r5_shard_aging.py, profiles/r5_short_runs.md)."""
    lib.check(lib.L.jga_stream_sync(None))
    time.sleep(0.3)


def _plugin(lib, abi, jpeg, reps):
    """reset -> header -> decode_image(RGB) per frame, pixels in img->pixels (host)."""
    with lib.Decoder(jpeg) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)
        px = d.header.width * d.header.height
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            d.reset()
            d.read_header()
            d.decode(abi.JPEG_DECODE_RGB)
            ts.append(time.perf_counter() - t0)
        # (the median: every process has ONE frame of 8-10 ms somewhere in its first thirty — the runtime bringing
        # something up — and a mean over twenty frames that holds it reads +0.4 ms: profiles/r5_plugin_copy_modes.md)
        ts.sort()
        dt = ts[len(ts) // 2]
    return {"ms_per_frame": round(dt * 1e3, 3), "ms_best": round(ts[0] * 1e3, 3), "ms_worst": round(ts[-1] * 1e3, 3),
            "Mpixel_s": round(px / dt / 1e6, 1),
            "note": "plugin decode_image(RGB): GPU entropy stage + fused kernel + D2H into img->pixels; median of %d frames" % reps}


def _pipeline_stream(lib, abi, np, orc, jpegs, order, nthreads, group, lanes, reps=2, pinned=False, times=None):
    """`order` = indices into jpegs, one job each.  Every output is kept (slices of one device
    buffer) and compared with the oracle's pixels of its file afterwards.  -> (best seconds, ok);
    `times` (a list) receives every repetition's seconds."""
    _, g = lib.geom_of(jpegs[0])
    n = len(order)
    ostride = (g.rgb_bytes + 255) // 256 * 256
    out = lib.DeviceBuffer(ostride * n)
    pins = [lib.PinnedBytes(j) for j in jpegs] if pinned else None
    src = [p.array for p in pins] if pinned else jpegs
    pl = lib.Pipeline(device=0, nthreads=nthreads, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2,
                      batch=group, depth=lanes)
    jobs = lib.Pipeline.make_jobs([src[i] for i in order], dev_outs=[out.ptr + k * ostride for k in range(n)],
                                  pinned=pinned)
    if reps >= 10:
        _settle(lib)
    ok = pl.run_jobs(jobs) == 0                                   # warm: every lane sized for its groups
    # (a short job timed many times: a fresh pipeline's first runs pay for the runtime bringing up its copy engines
    # and queues — ~9 ms per lane a few times, tools/archive/r3_shard_runs.py — which a median of 15 must not hold)
    for _ in range(6 if reps >= 10 else 0):
        ok = pl.run_jobs(jobs) == 0 and ok
    best = 1e9
    for _ in range(reps):
        lib.check(lib.L.jga_stream_sync(None))
        t0 = time.perf_counter()
        ok = pl.run_jobs(jobs) == 0 and ok
        dt = time.perf_counter() - t0
        best = min(best, dt)
        if times is not None:
            times.append(dt)
    pl.close()
    ok = ok and all(j.status == 0 for j in jobs)
    with ThreadPoolExecutor(max_workers=max(1, min(16, nthreads))) as ex:
        want = list(ex.map(lambda j: orc.decode_rgb(j)[1].reshape(-1), jpegs))
    for k, i in enumerate(order):
        if not ok:
            break
        ok = bool(np.array_equal(out.download(g.rgb_bytes, offset=k * ostride), want[i]))
    out.free()
    if pins:
        for p in pins:
            p.free()
    return best, ok, sum(j.h2d_bytes for j in jobs) // n


LINK_GBPS = 56.0          # what the host-to-device link sustains from pinned memory on these boxes (tools/h2d_probe.py)


def _pipeline_steady(lib, abi, np, orc, jpegs, nthreads, group, lanes, seconds=0.6, pinned=False, keep=True):
    """A stream long enough for a steady state (a timed region of >= 0.5 s in ONE jga_pipeline_run), with every output
    still checked: job i decodes file i % nfiles into slot i % R of a ring of R slots, R a multiple of nfiles, so a
    slot only ever receives the same file and the ring's final contents are compared with the oracle's pixels."""
    _, g = lib.geom_of(jpegs[0])
    nf = len(jpegs)
    px = g.width * g.height
    ostride = (g.rgb_bytes + 255) // 256 * 256
    ring = nf * max(1, min(256, (12 << 30) // ostride) // nf)
    out = lib.DeviceBuffer(ostride * ring)
    pins = [lib.PinnedBytes(j) for j in jpegs] if pinned else None
    src = [p.array for p in pins] if pinned else jpegs
    pl = lib.Pipeline(device=0, nthreads=nthreads, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2,
                      batch=group, depth=lanes)
    mk = lambda n: lib.Pipeline.make_jobs([src[i % nf] for i in range(n)],
                                          dev_outs=[out.ptr + (i % ring) * ostride for i in range(n)] if keep else None,
                                          pinned=pinned)
    cal = mk(max(ring, 4 * lanes))
    pl.run_jobs(cal)                                                 # (lanes size their buffers)
    t0 = time.perf_counter()
    ok = pl.run_jobs(cal) == 0
    rate = len(cal) / (time.perf_counter() - t0)                     # images / s, short-run regime: an underestimate
    n = int(rate * seconds * 1.6) // ring * ring + ring
    for attempt in range(3):
        jobs = mk(n)
        # (the same job list once untimed: a LONG run sizes every lane's buffers for full groups — pinned blobs of
        # ~200 MB per lane, a few tenths of a second of page pinning the first time — where the short calibration runs
        # above only sized them for theirs)
        ok = pl.run_jobs(jobs) == 0 and ok
        lib.check(lib.L.jga_stream_sync(None))
        t0 = time.perf_counter()
        ok = pl.run_jobs(jobs) == 0 and ok
        dt = time.perf_counter() - t0
        if dt >= min(seconds, 0.5) or not ok:
            break
        # (light content: the long run is much faster than the short calibration said — photograph-like 1080p files
        # ran 0.40 s where 0.6 was asked for: again, with as many jobs as THIS rate needs)
        n = int(n / dt * seconds * 1.15) // ring * ring + ring
    pl.close()
    ok = ok and all(j.status == 0 for j in jobs)
    with ThreadPoolExecutor(max_workers=max(1, min(16, nthreads))) as ex:
        want = list(ex.map(lambda j: orc.decode_rgb(j)[1].reshape(-1), jpegs))
    for k in range(ring if keep else 0):
        if not ok:
            break
        ok = bool(np.array_equal(out.download(g.rgb_bytes, offset=k * ostride), want[k % nf]))
    out.free()
    if pins:
        for p in pins:
            p.free()
    h2d = sum(j.h2d_bytes for j in jobs) / n
    ceiling = LINK_GBPS * 1e9 / (h2d / px) / 1e6                     # Mpixel/s the link allows at this many bytes per pixel
    return {"images": n, "seconds": round(dt, 3), "Mpixel_s": round(n * px / dt / 1e6, 1),
            "h2d_GBps": round(n * h2d / dt / 1e9, 1), "h2d_bytes_per_pixel": round(h2d / px, 4),
            "link_ceiling_Mpixel_s": round(ceiling, 1), "of_link_ceiling": round(n * px / dt / 1e6 / ceiling, 3),
            "outputs_checked": ring, "bit_exact_vs_oracle": ok}


def _pipeline_latency(lib, abi, jpeg, nthreads, reps=8):
    """One frame through the batch decoder with nothing else in flight: wall time of
    jga_pipeline_run (host parse, upload, GPU entropy stage, fused kernel, pixels in HBM)."""
    _, g = lib.geom_of(jpeg)
    out = lib.DeviceBuffer(g.rgb_bytes)
    pl = lib.Pipeline(device=0, nthreads=max(1, min(nthreads, 4)), out=abi.JPEG_DECODE_RGB, copy_back=False,
                      transport=2, batch=1, depth=1)
    jobs = lib.Pipeline.make_jobs([jpeg], dev_outs=[out.ptr])
    pl.run_jobs(jobs)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        pl.run_jobs(jobs)
        best = min(best, time.perf_counter() - t0)
    pl.close()
    out.free()
    return best


def run_configs(nthreads, cpu_threads, lanes=8, group=32, cpu_frames=2, cpu_rounds=2, quick=False, log=None,
                cpu_affinity=None):
    """-> {"config2_...": {...}, ...} (BASELINE.json configs 2, 3, 4, 5).  nthreads: host threads of
    the pipeline; cpu_threads: frame loops of the CPU paths (the granted CPUs)."""
    import numpy as np
    import oracle
    import cpu_paths
    from jpeg_gpu_amd import abi, lib, shard, synth
    orc = oracle.Oracle()
    say = log or (lambda *a: None)
    out = {}

    # The CPU paths run AFTER every GPU leg (round 5): they move the process onto all granted CPUs and back, and the
    # kernel then migrates heap pages between NUMA nodes — pages the device has mapped through registrations and
    # the runtime's pinning cache.  bench.py's configs leg (which passes cpu_affinity) died twice in three runs with
    # "Memory access fault by GPU ... Write access to a read-only page" on a host heap address in the legs right
    # behind a CPU leg; tools/configs_bench.py alone (no affinity change) never did in a dozen runs.
    deferred = []

    def cpu_rates_later(entry, files, w, h, frames):
        deferred.append((entry, files, w, h, frames))
        return None

    def cpu_rates(files, w, h, frames):
        """The CPU paths on the whole CPU grant (the pipeline legs run on the rank's own cores)."""
        mine = os.sched_getaffinity(0)
        if cpu_affinity:
            os.sched_setaffinity(0, cpu_affinity)
        try:
            return dict(cpu_paths.time_paths(files, w, h, cpu_threads, frames, cpu_rounds, single=False),
                        cores=cpu_threads)
        finally:
            os.sched_setaffinity(0, mine)

    def single(key, what, w, h, samp, ri, stream_n, kernel_n, photo=False, batch_n=0):
        t_cfg = time.perf_counter()
        n_files = 2 if quick else 4
        if photo:
            with ThreadPoolExecutor(max_workers=n_files) as ex:
                files = list(ex.map(lambda i: synth.photo_like_jpeg(w, h, samp, 90, ri, seed=1 + i), range(n_files)))
        else:
            files = [synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234 + i, restart_interval=ri)
                     for i in range(n_files)]
        px = w * h
        lat = _pipeline_latency(lib, abi, files[0], nthreads)
        order = [i % n_files for i in range(stream_n)]
        dt, ok, h2d = _pipeline_stream(lib, abi, np, orc, files, order, nthreads, group, lanes)
        steady = _pipeline_steady(lib, abi, np, orc, files, nthreads, group, lanes, seconds=0.15 if quick else 0.6)
        ok = ok and steady["bit_exact_vs_oracle"]
        e = {"what": what, "file_bytes": len(files[0]),
             "to_rgb_hbm": {"latency_ms": round(lat * 1e3, 3), "latency_Mpixel_s": round(px / lat / 1e6, 1),
                            "stream_images": stream_n, "Mpixel_s": round(stream_n * px / dt / 1e6, 1),
                            "h2d_bytes_per_image": int(h2d),
                            # (the figure above is a SHORT job: a few groups per lane, latency-bound; this one a
                            # timed region of >= 0.5 s)
                            "steady": steady},
             "to_host_pixels": _plugin(lib, abi, files[0], 3 if quick else 10),
             "device": dict(_kernel_alone(lib, np, files, kernel_n, 5 if quick else 20),
                            one_frame=_device_only(lib, files, 1, 2 if quick else 5)),
             "bit_exact_vs_oracle": ok,
             "cpu": None}
        cpu_rates_later(e, files, w, h, cpu_frames)
        if batch_n:
            # which of {link, entropy stage, block decode} bounds a stream of such files: per image, the link's
            # time for its bytes at LINK_GBPS against the two device stages' times in a batch of batch_n (the
            # stages share the one device: their sum is the device's time per image)
            sp = _device_split(lib, files, 2 * n_files if quick else batch_n, 2 if quick else 5)
            nb = sp["images"]
            e["device"].update(sp, batch=nb)
            link_ms = h2d / (LINK_GBPS * 1e6)
            hu, bd = sp["huffman_ms"] / nb, sp["idct_rgb_ms"] / nb
            e["per_image_ms"] = {"link": round(link_ms, 4), "entropy_stage": round(hu, 4), "block_decode": round(bd, 4)}
            e["bound_by"] = ("link" if link_ms > hu + bd else "entropy stage" if hu > bd else "block decode")
            e["device_ceiling_Mpixel_s"] = round(px / (hu + bd) / 1e3, 1)
            e["bytes_per_pixel"] = round(len(files[0]) / px, 4)
        if ri:
            # (SURVEY §8e's note) the same frame in 8 bands of MCU rows, one per GPU: what ONE of them does —
            # finds its band (host: a pass over the file for the restart markers), writes it as a file of
            # its own and decodes that; band 3 of 8 here.  No collective: the rows stay on their GPUs
            t0 = time.perf_counter()
            for _ in range(3):
                y0, rows, bf = shard.band_of_rank(files[0], 3, 8)
            cut = (time.perf_counter() - t0) / 3
            whole_rgb = orc.decode_rgb(files[0])[1] if not quick else None
            blat = _pipeline_latency(lib, abi, bf, nthreads)
            e["band_3_of_8"] = {"rows": [y0, rows], "file_bytes": len(bf), "host_cut_ms": round(cut * 1e3, 3),
                                "to_rgb_hbm": {"latency_ms": round(blat * 1e3, 3)},
                                "to_host_pixels": _plugin(lib, abi, bf, 3 if quick else 10),
                                "device": {"one_frame": _device_only(lib, [bf], 1, 2 if quick else 5)}}
            if whole_rgb is not None:
                with lib.Decoder(bf) as d:
                    d.read_header()
                    d.init_image()
                    d.decode(abi.JPEG_DECODE_RGB)
                    e["band_3_of_8"]["bit_exact_vs_oracle_rows_of_the_frame"] = bool(
                        (d.pixels() == whole_rgb[y0:y0 + rows]).all())
        out[key] = e
        say("configs: %s done in %.1f s" % (key, time.perf_counter() - t_cfg))

    single("config2_1080p_420_one_image", "1920x1080 4:2:0 q90, one image (BASELINE.json configs[1])",
           1920, 1080, "420", 0, 16 if quick else 256, 4 if quick else 128)
    single("config3_4k_444", "3840x2160 4:4:4 q90 (configs[2]: no-upsample path)",
           3840, 2160, "444", 0, 8 if quick else 96, 2 if quick else 24)

    # config 4: a batch of 1024 x 1080p 4:2:0 sharded over 8 GPUs — here ONE GPU takes all of it, and
    # rank 3's contiguous 128-file shard (what one GPU of the 8 gets)
    t_cfg = time.perf_counter()
    n4 = 64 if quick else 1024
    distinct = 8 if quick else 16
    files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(distinct)]
    px = 1920 * 1080
    e4 = {"what": "batch of %d x 1920x1080 4:2:0 q90 (configs[3]; %d distinct files repeated)" % (n4, distinct),
          "file_bytes": len(files[0])}
    oks = []
    for name, order in (("all_%d_on_one_gpu" % n4, list(range(n4))),
                        ("rank3_shard_of_8", list(shard.shard_range(n4, 3, 8)))):
        for kind, pinned in (("pageable_files", False), ("pinned_files", True)):
            say("configs: config4 %s %s ..." % (name, kind))
            ts = []
            # (a short job's time moves by +-10 % from run to run: the shard is timed 15 times and the MEDIAN quoted
            # — `ms` — with the best beside it; rounds 1-4 quoted the best of three)
            dt, ok, h2d = _pipeline_stream(lib, abi, np, orc, files, [i % distinct for i in order], nthreads,
                                           group, lanes, reps=3 if len(order) > 256 or quick else 15, pinned=pinned, times=ts)
            ts.sort()
            med = ts[len(ts) // 2]
            oks.append(ok)
            e4.setdefault("to_rgb_hbm", {}).setdefault(name, {})[kind] = {
                "images": len(order), "ms": round(med * 1e3, 2), "ms_best": round(dt * 1e3, 2), "runs": len(ts),
                "Mpixel_s": round(len(order) * px / med / 1e6, 1), "h2d_bytes_per_image": int(h2d),
                "h2d_GBps": round(len(order) * h2d / med / 1e9, 1)}
    say("configs: config4 plugin ...")
    e4["to_host_pixels"] = _plugin(lib, abi, files[0], 3 if quick else 10)
    say("configs: config4 device only ...")
    e4["device"] = dict(_kernel_alone(lib, np, files, n4, 3 if quick else 5),
                        whole_batch=_device_only(lib, files, n4, 1 if quick else 3),
                        shard_128=_device_only(lib, files, min(n4, 128), 1 if quick else 3))
    e4["bit_exact_vs_oracle"] = all(oks)
    e4["cpu"] = None
    cpu_rates_later(e4, files, 1920, 1080, max(cpu_frames, 4))
    out["config4_batch_1080p_420"] = e4
    say("configs: config4 done in %.1f s" % (time.perf_counter() - t_cfg))

    single("config5_8k_420_dri", "7680x4320 4:2:0 q90, DRI = one MCU row (configs[4]: GPU-parallel Huffman variant)",
           7680, 4320, "420", -1, 4 if quick else 32, 2 if quick else 8)
    # photograph-like content (README.md:18-22 of the reference: the gain depends on "how much it
    # is compressed"): a power-law-spectrum synthetic at q90, 0.14-0.17 B/px — the link has room there and the
    # device's entropy stage + block decode decide the rate
    single("photo_like_4k_420", "3840x2160 4:2:0 q90, photograph-like content (f^-1.5 spectrum + grain, ~0.14 B/px)",
           3840, 2160, "420", 0, 8 if quick else 192, 2 if quick else 48, photo=True, batch_n=48)
    single("photo_like_1080p_420", "1920x1080 4:2:0 q90, photograph-like content (~0.17 B/px)",
           1920, 1080, "420", 0, 16 if quick else 512, 4 if quick else 128, photo=True, batch_n=128)
    t_cpu = time.perf_counter()
    for entry, files, w, h, frames in deferred:
        entry["cpu"] = cpu_rates(files, w, h, frames)
    say("configs: CPU paths of the %d configs in %.1f s" % (len(deferred), time.perf_counter() - t_cpu))
    return out


def _child_main(spec_path):
    """bench.py's `configs` leg as a process of its own (bench.py --> python tools/configs_bench.py --child SPEC.json):
    what it measures is supplementary to the headline, and a fault of the device or the runtime in one of its many
    short legs must not take the bench line with it.  Prints ONE JSON line: {"configs": {...}, "headline": {...}}."""
    import json
    spec = json.load(open(spec_path))
    from jpeg_gpu_amd import abi, lib
    if os.environ.get("JGA_CONFIGS_FAULT"):                     # tests: a process that dies the way a GPU fault kills one
        os.abort()
    lib.check(lib.L.jga_set_device(int(spec.get("gpu", 0))))
    say = lambda *a: print(*a, file=sys.stderr, flush=True)
    res = {"configs": run_configs(spec["nthreads"], spec["cpu_threads"], lanes=spec["lanes"], group=spec["group"],
                                  quick=spec["quick"], log=say, cpu_affinity=spec.get("cpu_affinity")), "headline": {}}
    if spec.get("headline_file"):                # one frame of the headline's geometry alone, like the single-image configs
        try:
            data = open(spec["headline_file"], "rb").read()
            res["headline"]["latency_ms"] = round(_pipeline_latency(lib, abi, data, spec["nthreads"]) * 1e3, 3)
            res["headline"]["plugin"] = _plugin(lib, abi, data, 10)
        except Exception as e:                   # (supplementary)
            say("configs child: headline latency leg failed: %s" % e)
    print(json.dumps(res), flush=True)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--child":
    _child_main(sys.argv[2])
elif __name__ == "__main__":
    import json
    import __graft_entry__
    __graft_entry__.build()
    from jpeg_gpu_amd import shard
    quota = shard.cpu_quota()
    cpus = len(os.sched_getaffinity(0))
    budget = shard.rank_cpu_budget(cpus, 1, quota)
    nthreads = min(cpus, max(8, budget + budget // 2)) if quota else max(1, min(cpus, 96))
    res = run_configs(nthreads, budget, quick="--quick" in sys.argv, log=lambda *a: print(*a, file=sys.stderr, flush=True))
    for k, v in res.items():
        print(k, json.dumps(v))
