#!/usr/bin/env python3
"""bench.py — headline benchmark of the JPEG block-decode path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE pass of the hot path (fused dequantise + 2x1-D IDCT + chroma
upsample + YCbCr->RGB kernel) over one batch of synthetic 3840x2160 4:2:0 q90
baseline JPEGs whose packed coefficient planes are ALREADY RESIDENT IN HBM when
the timed region starts; the RGB8 output stays in HBM.  Images are independent:
each rank (one per GPU) owns its own batch, there is no data-path collective
(weak scaling); the only cross-rank traffic is the timing barrier / max.

Rank 0 prints ONE JSON line: BASELINE.json's metric (Mpixel/s) as `value`, plus
  roofline      achieved algorithmic HBM GB/s of the fused kernel, measured live
                with HIP events on the launch stream over the timed region
  cpu_baseline  the CPU port of the same path (oracle/, whole decode to RGB) timed
                on this box's host cores on a bounded sample (rank 0, N=1 only)
  e2e           supplementary: JPEG bytes in host RAM -> RGB, through the
                pipelined decoder (host Huffman threads + pinned H2D + kernel);
                PCIe/host-inclusive, never `value`.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
W, H, SAMPLING, QUALITY = 3840, 2160, "420", 90


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=48, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=6, help="distinct synthetic images")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the supplementary e2e leg")
    ap.add_argument("--prewarm", type=float, default=0.5,
                    help="seconds of untimed launches before warm-up (GPU clock ramp)")
    ap.add_argument("--no-pack", action="store_true", help="skip the PACK expansion leg")
    ap.add_argument("--no-other", action="store_true", help="skip the other-kernels leg")
    ap.add_argument("--e2e-images", type=int, default=96)
    ap.add_argument("--e2e-threads", type=int, default=0, help="0 = min(cores, 48)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-gpu-entropy", action="store_true",
                    help="skip the supplementary all-on-GPU leg (Huffman on the GPU)")
    return ap.parse_args()


def make_inputs(synth, n, rank):
    seeds = [1234 + rank * 1000 + i for i in range(n)]
    with ThreadPoolExecutor(max_workers=min(n, 8)) as ex:
        return list(ex.map(lambda s: synth.synthetic_jpeg(W, H, SAMPLING, QUALITY, seed=s),
                           seeds))


def device_facts(torch, dev):
    p = torch.cuda.get_device_properties(dev)
    return {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
            "hbm_GB": round(p.total_memory / 2**30, 1),
            "clock_MHz": getattr(p, "clock_rate", 0) // 1000 or None}


def cpu_baseline(jpegs, seconds):
    """The CPU port of the whole path (oracle.orc_decode_rgb: Huffman + float IDCT
    + clamp + upsample + RGB), one image per thread on the host cores; bounded."""
    import numpy as np
    import oracle
    orc = oracle.Oracle()
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    threads = max(1, min(ncpu, 64))
    # single core first: calibrates the sample size
    info = orc.parse(jpegs[0])
    scratch = np.empty(info.hblocks[0] * info.vblocks[0] * 64 * 3, np.uint8)
    rgb = np.empty((H, W, 3), np.uint8)
    t0 = time.perf_counter()
    orc.decode_rgb(jpegs[0], scratch, rgb)
    t1 = time.perf_counter() - t0
    # ~`seconds` of total CPU work, spread over the threads
    per_thread = max(1, min(16, int(round(seconds / max(t1, 1e-3) / threads))))

    def work(i):
        sc = np.empty_like(scratch)
        out = np.empty_like(rgb)
        for k in range(per_thread):
            orc.decode_rgb(jpegs[(i + k) % len(jpegs)], sc, out)
        return per_thread

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        done = sum(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    res = {
        "value": round(done * W * H / dt / 1e6, 1), "unit": "Mpixel/s", "cores": threads,
        "kind": "port",
        "sample": "%d decodes of %dx%d 4:2:0 q90 JPEGs to RGB on %d threads "
                  "(oracle.orc_decode_rgb, %.1f s)" % (done, W, H, threads, dt),
        "single_core_value": round(W * H / t1 / 1e6, 1),
    }
    # libjpeg-turbo where the box has it (Pillow's bundled copy): a sanity line, not the oracle —
    # its integer IDCT differs from src/dct.c by +-1 on ~2 % of samples
    try:
        import io
        from PIL import Image, features
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            Image.open(io.BytesIO(jpegs[0])).convert("RGB").load()
            best = min(best, time.perf_counter() - t0)
        res["libjpeg_turbo_single_core"] = {
            "value": round(W * H / best / 1e6, 1), "unit": "Mpixel/s",
            "note": "Pillow %s (libjpeg-turbo %s), full RGB decode, best of 5" % (
                Image.__version__, features.version("libjpeg_turbo") or "?")}
    except Exception as e:  # not importable on this box
        res["libjpeg_turbo_single_core"] = {"value": None, "note": "not available (%s)" % type(e).__name__}
    if oracle.Reference.available():
        ref = oracle.Reference()
        t0 = time.perf_counter()
        ref.decode(jpegs[0], oracle.YUV)
        res["reference_yuv_single_core"] = {
            "value": round(W * H / (time.perf_counter() - t0) / 1e6, 1), "unit": "Mpixel/s",
            "note": "the reference's own xjpeg+dct.c compiled (oracle/_ref), YUV stage "
                    "(it has no CPU RGB stage), 1 decode incl. image_init"}
    return res


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log("bench.py: --gpus %d needs torch.distributed.run with %d ranks; "
                "running rank-local only" % (args.gpus, args.gpus))
        args.gpus = world

    import torch                      # first: its bundled HIP runtime must be THE runtime
    import torch.distributed as dist
    import numpy as np
    import ctypes as C
    import __graft_entry__
    __graft_entry__.build()
    from jpeg_gpu_amd import abi, lib, shard, synth

    if not torch.cuda.is_available() or lib.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    lib.check(lib.L.jga_set_device(local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    # ---- inputs: synthetic JPEGs -> host entropy stage -> coefficient planes in HBM
    t_setup = time.perf_counter()
    jpegs = make_inputs(synth, args.distinct, rank)
    hdr, g = lib.geom_of(jpegs[0])
    B = args.batch
    cstride = (g.coef_shorts * 2 + 255) // 256 * 128          # shorts, 256-B aligned
    ostride = (g.rgb_bytes + 255) // 256 * 256
    d_coef = lib.DeviceBuffer(cstride * 2 * B)
    d_q = lib.DeviceBuffer(3 * 64 * 2 * B)
    d_out = lib.DeviceBuffer(ostride * B)
    coefs = [lib.entropy_decode(j, g) for j in jpegs]
    qt = np.zeros((B, 3, 64), np.uint16)
    for i in range(B):
        d_coef.upload(coefs[i % len(coefs)], offset=i * cstride * 2)
        qt[i] = lib.qtab_of(lib.parse_header(jpegs[i % len(jpegs)]))
    d_q.upload(qt)
    stream = lib.L.jga_stream_create()
    log("rank %d: setup %.1f s, batch %d x %dx%d, coef %.1f MB + rgb %.1f MB per image"
        % (rank, time.perf_counter() - t_setup, B, W, H, g.coef_shorts * 2 / 1e6,
           g.rgb_bytes / 1e6))

    def launch(reps):
        ms = C.c_float()
        lib.check(lib.L.jga_time_idct_batch(C.byref(g), B, d_coef.ptr, cstride, d_q.ptr, 1,
                                            d_out.ptr, ostride, 1, reps, stream, C.byref(ms)))
        return ms.value

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # power state: a cold GPU needs a few hundred ms of work before its clocks settle (part of
    # setup, like uploading the inputs; the W warm-up steps and the K timed steps follow)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        launch(20)
    if args.warmup > 0:
        launch(args.warmup)
    fence()
    t0 = time.perf_counter()
    ev_ms = launch(args.steps)            # K launches, HIP events on `stream`
    fence()
    dt = time.perf_counter() - t0
    # whole-job rate = pixels of all ranks / max time over ranks (no data-path collective)
    rate, _, dt = shard.aggregate_throughput(args.batch * W * H * args.steps, dt,
                                              dist if world > 1 else None, device="cuda")
    if world > 1:
        e = torch.tensor([ev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        ev_ms = float(e.item())

    # spot-check the last step's output against the oracle (outside the timed region)
    ok = True
    if rank == 0:
        import oracle
        want = oracle.Oracle().decode_rgb(jpegs[0])[1].reshape(-1)
        got = d_out.download(g.rgb_bytes, offset=0)
        ok = bool(np.array_equal(got, want))
        if not ok:
            raise SystemExit("bench.py: device output differs from the oracle")

    value = rate / 1e6
    alg_bytes = B * (g.coef_blocks * 128 + g.rgb_bytes)
    achieved = alg_bytes / (ev_ms * 1e-3) / 1e9
    traffic, valu = None, {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            if pmc.get("batch") == B and pmc.get("workload") == "%dx%d %s" % (W, H, SAMPLING):
                traffic = pmc.get("hbm_bytes_per_launch")
                valu = {k: pmc[k] for k in ("valu_insts_per_wave", "valu_busy_4clk") if k in pmc}
        except Exception:
            traffic = None
    out = {
        "metric": "Mpixel/s end-to-end decode, 4K 4:2:0 baseline JPEG",
        "value": round(value, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "3840x2160 4:2:0 q90 baseline JPEG; step = fused dequant+IDCT+"
                        "upsample+RGB over a batch of %d images per GPU, packed int16 "
                        "coefficient planes resident in HBM -> RGB8 in HBM" % B,
            "batch_per_gpu": B, "distinct_images": len(jpegs),
            "parallelism": "image-sharded x%d, no collectives" % world,
            "kernel": lib.L.jga_kernel_name(C.byref(g), 1).decode(),
            "bit_exact_vs_oracle": ok,
            "device": device_facts(torch, local_rank),
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms_per_launch": round(ev_ms, 4),
            # the other roof, from the same PMC passes as `traffic` (profiles/): the kernel is
            # co-limited by vector-ALU issue
            **({"valu": dict(valu, source="profiles/pmc_latest.json")} if valu else {}),
        },
    }

    if rank == 0:
        # what a plain device-to-device copy of the same volume reaches on this box (SURVEY.md
        # 8d: "state the measured copy ceiling next to the spec"): bytes read + bytes written
        src = torch.empty(alg_bytes // 2, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        out["roofline"]["device_copy_GBps"] = round(2 * src.numel() * 10 / e0.elapsed_time(e1) / 1e6, 1)
        del src, dst

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(jpegs, args.cpu_seconds)

    if rank == 0 and world == 1 and not args.no_other:
        # Supplementary: the other device stages, each against its own algorithmic bytes
        # (128 B per coded block in, output bytes out), HIP-event timed.
        others = {}

        def time_stage(gg, n, dc, cs, dq, do, os_, rgb, reps=10):
            ms = C.c_float()
            for r in (2, reps):
                lib.check(lib.L.jga_time_idct_batch(C.byref(gg), n, dc, cs, dq, 1, do, os_, rgb, r,
                                                    stream, C.byref(ms)))
            return ms.value

        ys = (g.yuv_bytes + 255) // 256 * 256
        d_yuv = lib.DeviceBuffer(ys * B)
        t = time_stage(g, B, d_coef.ptr, cstride, d_q.ptr, d_yuv.ptr, ys, 0)
        ab = B * (g.coef_blocks * 128 + g.yuv_bytes)
        others["yuv_stage_420"] = {"kernel": "jga_idct_yuv_kernel", "ms": round(t, 4),
                                   "GBps": round(ab / t / 1e6, 1), "images": B}
        # pass 3 alone on those planes (wall-clock over 10 launches)
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(10):
                lib.check(lib.L.jga_yuv_rgb_batch(C.byref(g), B, d_yuv.ptr, ys, d_out.ptr, ostride,
                                                  stream))
            lib.check(lib.L.jga_stream_sync(stream))
            t = (time.perf_counter() - t0) / 10 * 1e3
        ab = B * (g.yuv_bytes + g.rgb_bytes)
        others["yuv_to_rgb_420"] = {"kernel": "jga_yuv_rgb_kernel", "ms": round(t, 4),
                                    "GBps": round(ab / t / 1e6, 1), "images": B}
        d_yuv.free()
        for name, samp, n in (("rgb_444", "444", 24), ("grey", "grey", 48)):
            data = synth.synthetic_jpeg(W, H, samp, quality=90, seed=1234)
            h2, g2 = lib.geom_of(data)
            cs2 = (g2.coef_shorts * 2 + 255) // 256 * 128
            os2 = (g2.rgb_bytes + 255) // 256 * 256
            dc2, do2 = lib.DeviceBuffer(cs2 * 2 * n), lib.DeviceBuffer(os2 * n)
            c2 = lib.entropy_decode(data, g2)
            for i in range(n):
                dc2.upload(c2, offset=i * cs2 * 2)
            dq2 = lib.DeviceBuffer(384 * n)
            dq2.upload(np.tile(lib.qtab_of(h2).reshape(-1), n))
            t = time_stage(g2, n, dc2.ptr, cs2, dq2.ptr, do2.ptr, os2, 1)
            ab = n * (g2.coef_blocks * 128 + g2.rgb_bytes)
            others[name] = {"kernel": lib.L.jga_kernel_name(C.byref(g2), 1).decode(),
                            "ms": round(t, 4), "GBps": round(ab / t / 1e6, 1), "images": n}
            dc2.free(); do2.free(); dq2.free()
        out["other_kernels"] = others

    if rank == 0 and world == 1 and not args.no_e2e:
        n = args.e2e_images
        e2e = {}
        nthr = args.e2e_threads or max(1, min(os.cpu_count() or 1, 48))
        for copy_back, transport in ((False, 0), (True, 0), (False, 1), (False, 2), (True, 2)):
            nt = max(1, min(os.cpu_count() or 1, 96)) if transport == 2 else nthr
            pl = lib.Pipeline(device=local_rank, nthreads=nt, out=abi.JPEG_DECODE_RGB,
                              copy_back=copy_back, transport=transport, batch=48, depth=6)
            # enough images for every worker to reach steady state (its two slots allocated in the
            # warm-up, then several images each); the fast path needs more to ramp
            n = args.e2e_images*24 if transport == 2 else max(args.e2e_images, 4*nthr)
            if copy_back:
                n = min(n, 288)                   # 25 MB of host pixels per image
            jobs = [jpegs[i % len(jpegs)] for i in range(n)]
            # the caller's pixel buffers exist before the clock starts (zeros: pages touched)
            outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in range(n)] if copy_back else None
            nw = 288 if transport == 2 else 2*nthr                     # warm: slots/lanes, pages
            pl.run(jobs[:nw], host_outs=outs[:nw] if outs else None)
            t0 = time.perf_counter()
            rc, done = pl.run(jobs, host_outs=outs)
            te = time.perf_counter() - t0
            pl.close()
            key = "jpeg_host_to_rgb_host" if copy_back else "jpeg_host_to_rgb_hbm"
            if transport:
                key += ("", "_pack_transport", "_gpu_entropy")[transport]
            e2e[key] = {"value": round(n * W * H / te / 1e6, 1), "unit": "Mpixel/s",
                        "images": n, "ok": rc == 0,
                        "h2d_bytes_per_image": int(sum(j.h2d_bytes for j in done) // n)}
        e2e["host_threads"] = nthr
        e2e["note"] = "JPEG bytes in host RAM -> RGB; PCIe- and host-inclusive, not `value`. " \
                      "Default/pack transports: host Huffman threads + pinned hipMemcpyAsync + " \
                      "fused kernel; gpu_entropy: host only unstuffs, 6 lanes x 48 images, 3 with kernels queued at a time"
        out["e2e"] = e2e

    if rank == 0 and world == 1 and not args.no_pack:
        # Supplementary: PACK words + block index resident in HBM ->
        # jga_unpack_kernel -> QUANT planes.  Algorithmic bytes: 2 B/word + 4 B/block read,
        # 128 B/block written.
        pw = [lib.entropy_decode_pack(j, g)[:2] for j in jpegs]
        nidx = int(lib.L.jga_index_count(C.byref(g)))
        pstride = (max(len(p) for p, _ in pw) + 127) // 128 * 128
        hp = np.zeros((B, pstride), np.uint16)
        hi = np.zeros((B, nidx), np.int32)
        for i in range(B):
            p, ix = pw[i % len(pw)]
            hp[i, :len(p)] = p.view(np.uint16)
            hi[i] = ix
        d_pack, d_idx = lib.DeviceBuffer(hp.nbytes), lib.DeviceBuffer(hi.nbytes)
        d_pack.upload(hp)
        d_idx.upload(hi)
        reps = 10
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.check(lib.L.jga_unpack_batch(C.byref(g), B, d_pack.ptr, pstride, pstride,
                                                 d_idx.ptr, nidx, d_coef.ptr, cstride, None))
            lib.check(lib.L.jga_stream_sync(None))
            tu = (time.perf_counter() - t0) / reps
        nblk = sum(g.plane[p].hblocks * g.plane[p].vblocks for p in range(g.nplanes))
        words = sum(len(pw[i % len(pw)][0]) for i in range(B))
        ub = words * 2 + B * nblk * (4 + 128)
        same = bool(np.array_equal(d_coef.download(g.coef_shorts * 2, dtype=np.int16),
                                   lib.entropy_decode(jpegs[0], g)))
        out["pack_stage"] = {
            "kernel": "jga_unpack_kernel", "ms_per_launch": round(tu * 1e3, 4),
            "achieved_GBps": round(ub / tu / 1e9, 1), "algorithmic_bytes_per_launch": ub,
            "words_per_block": round(words / (B * nblk), 2),
            "pcie_bytes_vs_dense": round((words * 2 + B * nidx * 4) / (B * g.coef_shorts * 2), 3),
            "equals_host_quant_stage": same,
        }
        d_pack.free()
        d_idx.free()

    if rank == 0 and world == 1 and not args.no_gpu_entropy:
        # Supplementary: the whole decode on the GPU.  Entropy-coded
        # bytes resident in HBM -> self-synchronising parallel Huffman decode -> the same
        # fused kernel -> RGB in HBM.  No host Huffman, 8x fewer PCIe bytes.
        jobs = [jpegs[i % len(jpegs)] for i in range(B)]
        hb = lib.HuffBatch(B, sum(map(len, jobs)) + 4096 * B)
        t0 = time.perf_counter()
        hb.prepare(jobs)
        lib.check(lib.L.jga_stream_sync(None))
        t_prep = time.perf_counter() - t0
        d_q.upload(hb.qtabs())
        reps = 5
        th = ti = 0.0
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            rounds = hb.decode(d_coef.ptr, cstride)           # synchronous (checks errors)
            t1 = time.perf_counter()
            lib.check(lib.L.jga_idct_rgb_batch(C.byref(g), B, d_coef.ptr, cstride, d_q.ptr, 1,
                                               d_out.ptr, ostride, None))
            lib.check(lib.L.jga_stream_sync(None))
            t2 = time.perf_counter()
            if rep:                                           # first repetition warms up
                th += t1 - t0
                ti += t2 - t1
        got = d_out.download(g.rgb_bytes, offset=0)
        import oracle as _o
        same = bool(np.array_equal(got, _o.Oracle().decode_rgb(jobs[0])[1].reshape(-1)))
        hb.close()
        out["gpu_entropy"] = {
            "value": round(B * W * H * reps / (th + ti) / 1e6, 1), "unit": "Mpixel/s",
            "note": "JPEG entropy-coded bytes resident in HBM -> GPU Huffman -> fused kernel "
                    "-> RGB in HBM (device-only timed region, batch of %d)" % B,
            "huffman_ms": round(th / reps * 1e3, 3), "idct_rgb_ms": round(ti / reps * 1e3, 3),
            "sync_rounds": rounds, "bit_exact_vs_oracle": same,
            "prepare_ms_host_parse_unstuff_h2d": round(t_prep * 1e3, 2),
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    lib.L.jga_stream_destroy(stream)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
