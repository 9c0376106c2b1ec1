"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle
on the same seeded inputs — bit-exact (integer/byte outputs)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SAMPLINGS = ["grey", "444", "422", "420", "440", "411"]


def oracle_outputs(orc, info, quant):
    planes = orc.coef_to_planes(info, quant, True)
    return planes, orc.planes_to_rgb(info, planes)


def run_device(gpu, g, quant, qtab, rgb, dequant=True):
    out = gpu.idct_batch(g, quant[None], qtab[None], rgb=rgb, dequant=dequant)[0]
    return out


def check_image(gpu, orc, data, dequant=True):
    import oracle
    h, g = gpu.geom_of(data)
    info = orc.parse(data)
    quant = gpu.entropy_decode(data, g, dequant=not dequant)
    q = gpu.qtab_of(h)
    want_planes, want_rgb = oracle_outputs(orc, info, orc.decode(data, oracle.QUANT)[1])
    yuv = run_device(gpu, g, quant, q, rgb=False, dequant=dequant)
    for p, (a, b) in enumerate(zip(gpu.split_planes(g, yuv), want_planes)):
        assert np.array_equal(a, b), "plane %d: %d samples differ" % (p, (a != b).sum())
    rgb = run_device(gpu, g, quant, q, rgb=True, dequant=dequant)
    assert np.array_equal(rgb, want_rgb.reshape(-1)), \
        "rgb: %d bytes differ" % (rgb != want_rgb.reshape(-1)).sum()


@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("size", [(8, 8), (17, 9), (100, 75), (640, 360), (1031, 517)])
def test_kernel_matches_oracle(gpu, orc, synth, sampling, size):
    """Every sampling x aligned/odd/sub-MCU sizes (tile tails, edge crops,
    unaligned row pitch -> byte-store path)."""
    data = synth.synthetic_jpeg(size[0], size[1], sampling, quality=90, seed=size[0])
    check_image(gpu, orc, data)


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
def test_dct_stage_input(gpu, orc, synth, sampling):
    """dequant_on_device=0: coefficients already multiplied on the host (DCT stage)."""
    data = synth.synthetic_jpeg(200, 120, sampling, quality=75, seed=2)
    check_image(gpu, orc, data, dequant=False)


def test_golden_jpegs(gpu, golden_jpegs):
    """Committed vectors from the compiled reference: Y/Cb/Cr planes bit-exact."""
    for name in golden_jpegs.names:
        data = golden_jpegs.jpeg(name)
        h, g = gpu.geom_of(data)
        quant = golden_jpegs[name + ".quant"]
        yuv = run_device(gpu, g, quant, gpu.qtab_of(h), rgb=False)
        for a, b in zip(gpu.split_planes(g, yuv), golden_jpegs.planes(name)):
            assert np.array_equal(a, b), name


def test_golden_blocks_through_kernel(gpu, golden_blocks, synth):
    """4096+ reference in/out blocks pushed through the kernel as a grey image
    with unit quantisers: plane = clamp255(idct + 128) (xjpeg.c:578)."""
    from jpeg_gpu_amd import abi
    from test_layout_abi import make_header
    inp, out = golden_blocks
    n = len(inp)
    wb = 64
    hb = (n + wb - 1) // wb
    g = gpu.geom_from_header(make_header(abi, wb * 8, hb * 8, [(1, 1)]))
    coef = np.zeros(g.coef_shorts, np.int16)
    coef[:n * 64] = inp.reshape(-1)
    q = np.ones((3, 64), np.uint16)
    yuv = run_device(gpu, g, coef, q, rgb=False)
    plane = gpu.split_planes(g, yuv)[0]
    got = plane.reshape(hb, 8, wb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)[:n]
    want = np.clip(out.astype(np.int32) + 128, 0, 255).astype(np.uint8)
    assert np.array_equal(got, want), "%d blocks differ" % (got != want).any(1).sum()


def test_extreme_coefficients(gpu, orc, synth):
    """Full int16 coefficient range with 16-bit quantisers: dequant wraps mod 2^16
    (xjpeg.c:501-503) and the (short) cast after floor wraps (dct.c:118)."""
    import oracle
    from jpeg_gpu_amd import abi
    from test_layout_abi import make_header
    rng = np.random.default_rng(5)
    for samp, name in (([(2, 2), (1, 1), (1, 1)], "420"), ([(1, 1)], "grey")):
        h = make_header(abi, 256, 64, samp)
        g = gpu.geom_from_header(h)
        coef = rng.integers(-32768, 32768, g.coef_shorts).astype(np.int16)
        coef.reshape(-1, 64)[::3, 1:] = 0            # some DC-only giants
        q = rng.integers(1, 65536, (3, 64)).astype(np.uint16)
        q[:, ::2] = rng.integers(1, 4, (3, 32))
        info = orc.parse(synth.synthetic_jpeg(256, 64, name))
        for p in range(info.ncomps):
            for k in range(64):
                info.quant[p][k] = int(q[p, k])
        want_planes, want_rgb = oracle_outputs(orc, info, coef)
        yuv = run_device(gpu, g, coef, q, rgb=False)
        for a, b in zip(gpu.split_planes(g, yuv), want_planes):
            assert np.array_equal(a, b)
        rgb = run_device(gpu, g, coef, q, rgb=True)
        assert np.array_equal(rgb, want_rgb.reshape(-1))


def test_ieee1180_on_device(gpu):
    """The reference's own IDCT accuracy test (test/dct.c:229-261) on the kernel:
    run via unit quantisers, recover idct = plane - 128 where unclamped."""
    from jpeg_gpu_amd import abi
    from test_layout_abi import make_header
    from test_oracle_pins import ieee1180_check, assert_ieee1180

    def dev_idct(blocks):
        n = len(blocks)
        wb = 50
        hb = (n + wb - 1) // wb
        g = gpu.geom_from_header(make_header(abi, wb * 8, hb * 8, [(1, 1)]))
        coef = np.zeros(g.coef_shorts, np.int16)
        # bias DC by -? no: compare in the [-128,127] window, the spec range is
        # [-256,255]; run twice with the level shift moved by +-128 via DC offset
        # is not exact, so instead check only samples that are unclamped.
        coef[:n * 64] = blocks.reshape(-1)
        plane = gpu.split_planes(g, run_device(gpu, g, coef, np.ones((3, 64), np.uint16),
                                               rgb=False))[0]
        got = plane.reshape(hb, 8, wb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)[:n]
        return got.astype(np.int32) - 128

    stats = ieee1180_check_clamped(dev_idct)
    assert_ieee1180(stats)


def ieee1180_check_clamped(idct_fn, n=1000):
    """IEEE-1180 statistics restricted to the [-128,127] output window the u8 plane
    can represent (both sides clamped identically)."""
    from test_oracle_pins import ieee1180_blocks, dct_matrix
    m = dct_matrix()
    stats = []
    for lo, hi in ((-256, 255), (-5, 5), (-300, 300)):
        for sign in (1, -1):
            px = ieee1180_blocks(n, lo, hi, sign).astype(np.float64)
            coef = np.clip(np.rint(np.einsum("ij,njk,lk->nil", m, px, m)), -2048, 2047)
            ref = np.clip(np.rint(np.einsum("ji,njk,kl->nil", m, coef, m)), -128, 127)
            got = idct_fn(coef.astype(np.int16).reshape(-1, 64)).reshape(-1, 8, 8)
            err = got - ref
            stats.append(dict(peak=np.abs(err).max(), pmse=(err ** 2).mean(0).max(),
                              omse=(err ** 2).mean(), pme=np.abs(err.mean(0)).max(),
                              ome=abs(err.mean())))
    return stats


def test_batch_of_images(gpu, orc, synth):
    """Batched launch: n images of one geometry, distinct content and tables."""
    import oracle
    datas = [synth.synthetic_jpeg(328, 200, "420", quality=60 + 5 * i, seed=i) for i in range(5)]
    h0, g = gpu.geom_of(datas[0])
    coefs = np.stack([gpu.entropy_decode(d, g) for d in datas])
    qt = np.stack([gpu.qtab_of(gpu.parse_header(d)) for d in datas])
    rgb = gpu.idct_batch(g, coefs, qt, rgb=True)
    yuv = gpu.idct_batch(g, coefs, qt, rgb=False)
    for i, d in enumerate(datas):
        info, planes = orc.decode(d, oracle.YUV)
        assert np.array_equal(rgb[i], orc.planes_to_rgb(info, planes).reshape(-1)), i
        for a, b in zip(gpu.split_planes(g, yuv[i]), planes):
            assert np.array_equal(a, b), i


def test_full_size_configs_match_oracle(gpu, orc, synth):
    """BASELINE configs at full size: 1080p 4:2:0 and 4K 4:4:4 against the oracle."""
    for w, h, s in ((1920, 1080, "420"), (3840, 2160, "444")):
        check_image(gpu, orc, synth.synthetic_jpeg(w, h, s, quality=90, seed=1234))


def test_4k_420_properties(gpu, orc, synth):
    """Headline config, size-independent properties: (i) the RGB image equals the
    RGB stage applied to the device's own planes, (ii) decoding is idempotent /
    deterministic across launches, (iii) a checksum of per-tile checksums matches
    the oracle's."""
    import oracle
    data = synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234)
    h, g = gpu.geom_of(data)
    quant = gpu.entropy_decode(data, g)
    q = gpu.qtab_of(h)
    rgb1 = run_device(gpu, g, quant, q, rgb=True)
    rgb2 = run_device(gpu, g, quant, q, rgb=True)
    assert np.array_equal(rgb1, rgb2)
    yuv = run_device(gpu, g, quant, q, rgb=False)
    info = orc.parse(data)
    assert np.array_equal(orc.planes_to_rgb(info, gpu.split_planes(g, yuv)).reshape(-1), rgb1)
    want = orc.decode_rgb(data)[1].reshape(-1)
    tiles_g = rgb1.reshape(-1, 4096).astype(np.uint64).sum(1)
    tiles_w = want.reshape(-1, 4096).astype(np.uint64).sum(1)
    assert np.array_equal(tiles_g, tiles_w)
    assert np.array_equal(rgb1, want)


def test_linearity_in_dc(gpu):
    """Domain property: adding 8*k to every DC (unit quantisers) adds k to every
    unclamped output sample (dct.c scaling S[0]*S[0] = 1/8 exactly in float32)."""
    from jpeg_gpu_amd import abi
    from test_layout_abi import make_header
    g = gpu.geom_from_header(make_header(abi, 512, 8, [(1, 1)]))
    rng = np.random.default_rng(9)
    base = np.zeros(g.coef_shorts, np.int16)
    base.reshape(-1, 64)[:, 0] = rng.integers(-50, 50, 64) * 8
    q = np.ones((3, 64), np.uint16)
    a = run_device(gpu, g, base, q, rgb=False).astype(np.int32)
    shifted = base.copy()
    shifted.reshape(-1, 64)[:, 0] += 8 * 5
    b = run_device(gpu, g, shifted, q, rgb=False).astype(np.int32)
    assert np.array_equal(b, a + 5)


def test_plugin_gpu_stages(gpu, orc, synth):
    """HIPJPEG vtable YUV + RGB through the reference's call order, with reset
    between frames and a different file of the same geometry on the same ctx."""
    import oracle
    from jpeg_gpu_amd import abi
    d1 = synth.synthetic_jpeg(322, 241, "420", seed=1)
    d2 = synth.synthetic_jpeg(322, 241, "420", seed=2, quality=70)
    with gpu.Decoder(d1) as d:
        d.read_header()
        d.init_image()
        for data in (d1, d2, d1):
            d.reset(data)
            d.read_header()
            d.decode(abi.JPEG_DECODE_YUV)
            info, planes = orc.decode(data, oracle.YUV)
            for a, b in zip(d.planes(), planes):
                assert np.array_equal(a, b)
            d.reset()
            d.read_header()
            d.decode(abi.JPEG_DECODE_RGB)
            assert np.array_equal(d.pixels(), orc.planes_to_rgb(info, planes))
    g = synth.synthetic_jpeg(77, 33, "grey")
    with gpu.Decoder(g) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)
        info, planes = orc.decode(g, oracle.YUV)
        assert np.array_equal(d.pixels(), planes[0][:33, :77])


@pytest.mark.parametrize("transport", [0, 1, 2])
def test_pipeline(gpu, orc, synth, transport):
    """Pipelined batch decoder: mixed geometries, copy-back to host, results equal
    the oracle's whole-path decode — with dense planes (0), the PACK wire format (1) or only
    the entropy-coded bytes (2: GPU entropy stage, groups of 4 with mixed geometries inside)
    crossing PCIe."""
    from jpeg_gpu_amd import abi
    specs = [(320, 200, "420"), (128, 64, "444"), (200, 100, "422"), (64, 64, "grey"),
             (320, 200, "420"), (97, 55, "420")] * 3
    datas = [synth.synthetic_jpeg(w, h, s, seed=i, restart_interval=(i % 3) * 4)
             for i, (w, h, s) in enumerate(specs)]
    outs = [np.zeros(w * h * (1 if s == "grey" else 3), np.uint8) for (w, h, s) in specs]
    pl = gpu.Pipeline(device=0, nthreads=3, out=abi.JPEG_DECODE_RGB, copy_back=True,
                      transport=transport, batch=4)
    try:
        rc, jobs = pl.run(datas, host_outs=outs)
        assert rc == 0
        for i, d in enumerate(datas):
            assert jobs[i].status == 0
            assert np.array_equal(outs[i], orc.decode_rgb(d)[1].reshape(-1)), i
        if transport >= 1:      # the compact form is what crossed PCIe
            _, g = gpu.geom_of(datas[0])
            assert 0 < jobs[0].h2d_bytes < g.coef_shorts * 2
        # a corrupt job fails alone
        bad = list(datas[:4])
        bad[1] = bad[1][:200]
        rc, jobs = pl.run(bad, host_outs=outs[:4])
        assert rc == 1 and [jobs[i].status for i in range(4)] == [0, 1, 0, 0]
    finally:
        pl.close()


def test_pipeline_gpu_entropy_group_with_a_damaged_scan(gpu, orc, synth):
    """One member of a same-geometry group has a damaged scan: the group's batch decode fails as
    a whole, its members are then decoded one by one — the damaged one alone reports failure."""
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(320, 200, "420", quality=80, seed=i) for i in range(6)]
    bad = bytearray(datas[2])
    sos = bad.find(b"\xff\xda")
    del bad[sos + 200:sos + 1200]                 # a kilobyte of entropy-coded data missing
    datas[2] = bytes(bad)
    _, g = gpu.geom_of(datas[0])
    outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in datas]
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2,
                      batch=1, depth=2)           # 320x200 frames: groups of 16 -> one group of 6
    try:
        rc, jobs = pl.run(datas, host_outs=outs)
        assert rc == 1
        assert [jobs[i].status for i in range(6)] == [0, 0, 1, 0, 0, 0]
        for i in (0, 1, 3, 4, 5):
            assert np.array_equal(outs[i], orc.decode_rgb(datas[i])[1].reshape(-1)), i
    finally:
        pl.close()


@pytest.mark.parametrize("more", [{}, {"unstuff": 1}, {"unstuff": 2, "input_cache_mb": -1}, {"spin_waits": 1}])
def test_pipeline_gpu_entropy_batches(gpu, orc, synth, more):
    """transport 2 on a stream of same-geometry images: full groups, a ragged last group,
    results left in HBM at caller-given addresses and in internal buffers; with the clean-up where the
    pipeline puts it, on the host, on the device with copy calls naming the files, and with spinning waits."""
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(640, 360, "420", quality=50 + i, seed=i, restart_interval=(i % 2) * 40)
             for i in range(37)]
    _, g = gpu.geom_of(datas[0])
    want = [orc.decode_rgb(d)[1].reshape(-1) for d in datas]
    outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in datas]
    # `batch` counts 4K frames: 640x360 frames fill a group 16 to 1 -> groups of 16, 16 and 5
    pl = gpu.Pipeline(device=0, nthreads=6, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2,
                      batch=1, depth=3, **more)
    dbuf = gpu.DeviceBuffer(gpu._align(g.rgb_bytes) * len(datas))
    try:
        rc, jobs = pl.run(datas, host_outs=outs)
        assert rc == 0
        for i in range(len(datas)):
            assert np.array_equal(outs[i], want[i]), i
        devs = [dbuf.ptr + gpu._align(g.rgb_bytes) * i for i in range(len(datas))]
        rc, jobs = pl.run(datas, dev_outs=devs)
        assert rc == 0
        got = dbuf.download().reshape(len(datas), -1)
        for i in range(len(datas)):
            assert np.array_equal(got[i, :g.rgb_bytes], want[i]), i
    finally:
        pl.close()
        dbuf.free()


def oracle_quant(orc, data):
    """The oracle's QUANT-stage planes of a file (oracle.c, pinned to the compiled reference):
    what every coefficient-producing GPU path is compared with."""
    import oracle
    return orc.decode(data, oracle.QUANT)[1]


@pytest.mark.parametrize("transport", [0, 1, 2])
def test_pipeline_pinned_destinations(gpu, orc, synth, transport):
    """copy_back into buffers the caller declares pinned (jga_job.pinned bit 1): the pixels are
    DMA'd straight into them, no staging buffer, no host memcpy; pageable and pinned
    destinations may be mixed in one run.  Same pixels."""
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(320 + 16 * (i % 2), 200, ["420", "444", "grey"][i % 3], quality=70 + i,
                                  seed=i, restart_interval=[0, -1][i % 2]) for i in range(12)]
    want = [orc.decode_rgb(d)[1].reshape(-1) for d in datas]
    pins = [gpu.PinnedBytes(bytes(w.size)) for w in want]
    outs = [p.array if i % 3 else np.zeros(want[i].size, np.uint8) for i, p in enumerate(pins)]
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True,
                      transport=transport, batch=4, depth=2)
    try:
        jobs = gpu.Pipeline.make_jobs(datas, host_outs=outs)
        for i in range(len(datas)):
            jobs[i].pinned = 2 if i % 3 else 0
        assert pl.run_jobs(jobs) == 0
        for i in range(len(datas)):
            assert np.array_equal(outs[i], want[i]), (transport, i)
    finally:
        pl.close()
        for p in pins:
            p.free()


@pytest.mark.parametrize("mode", ["pinned", "device", "host", "mixed"])
def test_pipeline_unstuff_modes_and_pinned_inputs(gpu, orc, synth, mode):
    """transport 2 with the scan clean-up on the host, on the GPU, and — for jobs whose files lie
    in pinned memory — on the GPU with the scans DMA'd straight out of the callers' buffers
    (no host copy; with four host threads `auto` means the device); a group that mixes pinned
    and pageable files has its scans copied by the host.  Same pixels every way, damaged member
    reported alone."""
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(640, 360, "420", quality=60 + i, seed=i, restart_interval=(i % 3) * 20)
             for i in range(20)]
    bad = bytearray(datas[7])
    bad[len(bad) // 2:len(bad) // 2 + 40] = b"\xff\xd9" * 20          # an EOI in mid-scan
    datas[7] = bytes(bad)
    _, g = gpu.geom_of(datas[0])
    want = [orc.decode_rgb(d)[1].reshape(-1) if i != 7 else None for i, d in enumerate(datas)]
    pins = [gpu.PinnedBytes(d) for d in datas] if mode in ("pinned", "mixed") else []
    srcs = [p.array for p in pins] if mode == "pinned" else list(datas)
    outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in datas]
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2,
                      batch=1, depth=2, unstuff={"pinned": 0, "mixed": 0, "device": 2, "host": 1}[mode])
    try:
        jobs = gpu.Pipeline.make_jobs(srcs, host_outs=outs, pinned=(mode == "pinned"))
        if mode == "mixed":
            for i in range(0, len(datas), 2):
                jobs[i].jpeg, jobs[i].pinned = pins[i].array.ctypes.data, 1
        rc = pl.run_jobs(jobs)
        assert rc == 1 and [j.status for j in jobs] == [int(i == 7) for i in range(len(datas))]
        for i in range(len(datas)):
            if i != 7:
                assert np.array_equal(outs[i], want[i]), (mode, i)
        assert not outs[7].any()              # a failed job hands out zeros, not leftovers of other images
        # (clean-up on the device: the device reads every file where it lies — pinned, or ordinary memory the input
        # cache registers; buffers under 64 KB are cheaper to copy than to register — and no host core passes over
        # it; on the host: every byte)
        for i, j in enumerate(jobs):
            if i != 7:
                copied = mode == "host" or (not (j.pinned & 1) and j.size < 65536)
                assert j.host_bytes == (j.size if copied else 0), (mode, i, j.size)
    finally:
        pl.close()
        for p in pins:
            p.free()


def test_pipeline_input_cache_registers_pageable_buffers(gpu, orc, synth):
    """jga_pipeline_config.input_cache_mb: callers' ordinary (pageable) buffers are registered with the
    device the first (or input_cache_sight-th) time a run sees them and read where they lie from then
    on, least recently used out first when the cache is full; buffers the cache does not hold (under 64 KB,
    larger than the cache, seen once under the second-sight policy) are copied by a host core;
    input_cache_mb = -1: no cache, the copies name the buffers as they are and the runtime pins them; -2: a host
    core copies every file (rounds 2-3); forget / explicit register; same pixels as the oracle's every time."""
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=40 + i, restart_interval=(i % 2) * 80)
             for i in range(10)]
    small = synth.synthetic_jpeg(1280, 720, "420", quality=5, seed=3)
    assert all(len(d) > 200_000 for d in datas) and len(small) < 64 * 1024
    _, g = gpu.geom_of(datas[0])
    want = [orc.decode_rgb(d)[1].reshape(-1) for d in datas + [small]]
    arrs = [np.frombuffer(d, np.uint8).copy() for d in datas + [small]]        # (malloc memory: pageable)
    outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in arrs]

    def run(pl, idx):
        for o in outs:
            o[:] = 0
        jobs = gpu.Pipeline.make_jobs([arrs[i] for i in idx], host_outs=[outs[i] for i in idx])
        assert pl.run_jobs(jobs) == 0
        for i in idx:
            assert np.array_equal(outs[i], want[i]), i
        return jobs
    total_mb = sum(a.size for a in arrs[:10]) / 2**20
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=2,
                      unstuff=2, input_cache_mb=int(total_mb) + 2)
    try:
        jobs = run(pl, list(range(11)))
        c = pl.counters()
        assert c["cleanup_on_device"] == 1 and c["registered"] == 10 and c["evicted"] == 0
        assert all(j.host_bytes == 0 for j in jobs[:10]) and jobs[10].host_bytes == jobs[10].size   # first sight registers; the small file is copied
        jobs = run(pl, [3, 3, 9, 0, 3])
        c2 = pl.counters()
        assert c2["registered"] == 10 and c2["jobs_in_place"] == c["jobs_in_place"] + 5
        pl.forget_input(arrs[3])
        assert pl.counters()["registered_MB"] <= c2["registered_MB"]
        run(pl, [3])
        assert pl.counters()["registered"] == 11                                # ... and is registered again when it comes back
    finally:
        pl.close()
    # a cache too small for all of them: least recently used out first, never the ones a running group holds;
    # second-sight policy: a buffer seen once is not registered
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=2,
                      unstuff=2, input_cache_mb=1, input_cache_sight=2)
    try:
        run(pl, list(range(10)))
        assert pl.counters()["registered"] == 0
        run(pl, list(range(10)))                     # second sight: registered while there is room (buffers a running
        c = pl.counters()                            # group holds are never evicted: the others go unregistered)
        assert c["registered"] >= 2 and c["registered_MB"] <= 1
        run(pl, [7, 8, 9])                           # ... and idle ones make room, least recently used first
        c = pl.counters()
        assert c["evicted"] >= 1 and c["registered_MB"] <= 1
        pl.register_input(arrs[0])                                              # explicit: at once, whatever the sight count
        n0 = pl.counters()["registered"]
        run(pl, [0])
        assert pl.counters()["registered"] == n0                                # (it was there already)
        big = np.frombuffer(synth.synthetic_jpeg(1280, 720, "444", quality=100, seed=1), np.uint8).copy()
        assert big.size > 1 << 20
        with pytest.raises(gpu.JgaError):
            pl.register_input(big)                                              # larger than the whole cache
    finally:
        pl.close()
    # the default: registrations that live as long as the groups that use them — the files are read where they lie,
    # and NOTHING of the caller's memory is still registered when the run returns (a caller is free to
    # free() its buffers the moment jga_pipeline_run is back); explicit registration needs the persistent cache
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=2, unstuff=2)
    try:
        jobs = run(pl, list(range(10)))
        c = pl.counters()
        assert c["registered"] == 10 and c["registered_MB"] == 0 and all(j.host_bytes == 0 for j in jobs)
        run(pl, list(range(10)))
        c = pl.counters()
        assert c["registered"] == 20 and c["registered_MB"] == 0 and c["evicted"] == 0
        with pytest.raises(gpu.JgaError):
            pl.register_input(arrs[0])
    finally:
        pl.close()
    # no cache: -1 names the callers' memory in the copies (no host pass), -2 copies it
    for mb, copied in ((-1, False), (-2, True)):
        pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=2,
                          unstuff=2, input_cache_mb=mb)
        try:
            jobs = run(pl, list(range(11)))
            assert pl.counters()["registered"] == 0 and all(j.host_bytes == (j.size if copied else 0) for j in jobs)
        finally:
            pl.close()
    # with the clean-up on the host the cache is not used at all (the host reads every byte anyway)
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=2,
                      unstuff=1, input_cache_mb=64)
    try:
        jobs = run(pl, list(range(10)))
        assert pl.counters()["registered"] == 0 and all(j.host_bytes == j.size for j in jobs)
    finally:
        pl.close()


def test_input_cache_never_serves_a_stale_buffer(gpu, orc, synth):
    """The reference's caller keeps its file in plain malloc memory and frees it with the jpeg_info
    (src/jpeg_info.c:31-62).  A buffer that is freed (unmapped) and handed out again AT THE SAME ADDRESS AND SIZE
    for another file must decode as the NEW file: the cache's registration was made for the old contents — and may
    name pages the buffer no longer has.  Every entry carries a fingerprint of its file that is re-read at every
    sight; a mismatch drops the registration and registers afresh (counter `stale`)."""
    import ctypes as C
    import mmap
    from jpeg_gpu_amd import abi
    a = synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=501)
    b = synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=502)
    n = (max(len(a), len(b)) + 2 * mmap.PAGESIZE) // mmap.PAGESIZE * mmap.PAGESIZE
    a, b = a + b"\0" * (n - len(a)), b + b"\0" * (n - len(b))     # (bytes after the EOI are nobody's business)
    assert a[:64] == b[:64] and a[-64:] == b[-64:] and len(a) == len(b)         # only the scans differ
    _, g = gpu.geom_of(a)
    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    MAP_FIXED = 0x10
    addr = libc.mmap(None, n, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert addr not in (None, C.c_void_p(-1).value)
    out = np.zeros(g.rgb_bytes, np.uint8)
    pl = gpu.Pipeline(device=0, nthreads=2, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=2, unstuff=2,
                      input_cache_mb=64)                     # (the persistent cache: registrations outlive their runs)
    try:
        def decode_what_lies_there():
            buf = np.ctypeslib.as_array((C.c_ubyte * n).from_address(addr))
            jobs = gpu.Pipeline.make_jobs([buf], host_outs=[out])
            assert pl.run_jobs(jobs) == 0
            return out.copy()
        C.memmove(addr, a, n)
        assert np.array_equal(decode_what_lies_there(), orc.decode_rgb(a)[1].reshape(-1))
        assert np.array_equal(decode_what_lies_there(), orc.decode_rgb(a)[1].reshape(-1))
        c = pl.counters()
        assert c["registered"] == 1 and c["jobs_in_place"] == 2 and c["stale"] == 0
        # the caller frees the buffer; the allocator hands the same range out again, backed by fresh pages
        assert libc.munmap(addr, n) == 0
        again = libc.mmap(addr, n, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | MAP_FIXED, -1, 0)
        assert again == addr
        C.memmove(addr, b, n)
        assert np.array_equal(decode_what_lies_there(), orc.decode_rgb(b)[1].reshape(-1))
        c = pl.counters()
        assert c["stale"] == 1 and c["registered"] == 2
        # ... and a buffer that is simply overwritten in place (same pages) is caught the same way
        C.memmove(addr, a, n)
        assert np.array_equal(decode_what_lies_there(), orc.decode_rgb(a)[1].reshape(-1))
        assert pl.counters()["stale"] == 2
    finally:
        pl.close()
        libc.munmap(addr, n)


def test_input_cache_with_callers_buffers_that_share_pages(gpu, orc, synth):
    """Files that lie back to back in one malloc arena (the reference's caller mallocs its file, src/jpeg_info.c:31-56:
    under glibc's mmap threshold, or above it once the threshold has grown, neighbours share the page at their seam),
    registered by the input cache one by one, in any order, by several lanes at once: every file decodes as itself,
    run after run, and a buffer is only taken as "registered by somebody else" if every page of it answers."""
    files = [synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=600 + i) for i in range(6)]
    arena = np.zeros(sum(map(len, files)) + 64, np.uint8)
    views, o = [], 13                                            # (no alignment whatsoever)
    for f in files:
        arena[o:o + len(f)] = np.frombuffer(f, np.uint8)
        views.append(arena[o:o + len(f)])
        o += len(f)
    _, g = gpu.geom_of(files[0])
    want = [orc.decode_rgb(f)[1].reshape(-1) for f in files]
    outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in files]
    from jpeg_gpu_amd import abi
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=1, depth=3, unstuff=2)
    try:
        for rep in range(4):
            order = [(5 * k + rep) % len(files) for k in range(len(files))]        # (a permutation)
            for out in outs:
                out[:] = 0
            jobs = gpu.Pipeline.make_jobs([views[i] for i in order], host_outs=[outs[i] for i in order])
            assert pl.run_jobs(jobs) == 0
            for i in range(len(files)):
                assert np.array_equal(outs[i], want[i]), (rep, i)
        c = pl.counters()
        assert c["registered"] >= 1 and c["stale"] == 0, c
    finally:
        pl.close()


# ---- PACK wire format expanded on the GPU ---------------------------

@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("ri", [0, 5])
def test_gpu_unpack_equals_quant_stage(gpu, orc, synth, sampling, ri):
    """Words + index -> planes on the GPU == the ORACLE's QUANT planes of the same file."""
    datas = [synth.synthetic_jpeg(333, 211, sampling, quality=q, restart_interval=ri, seed=q)
             for q in (95, 60, 20)]
    _, g = gpu.geom_of(datas[0])
    packs, indexes = zip(*[gpu.entropy_decode_pack(d, g)[:2] for d in datas])
    got = gpu.gpu_unpack(g, packs, indexes)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], oracle_quant(orc, d)), (sampling, ri, i)


def test_gpu_unpack_golden_words(gpu, golden_jpegs):
    """Words + index produced by the COMPILED REFERENCE -> the reference's QUANT planes."""
    for name in golden_jpegs.names:
        _, g = gpu.geom_of(golden_jpegs.jpeg(name))
        got = gpu.gpu_unpack(g, [golden_jpegs[name + ".pack"]], [golden_jpegs[name + ".index"]])
        assert np.array_equal(got[0], golden_jpegs[name + ".quant"]), name


def test_gpu_unpack_crafted_words_match_oracle(gpu, orc, synth):
    """Arbitrary words: 12-bit sign wrap, ZRL chains, runs past coefficient 63, blocks with and
    without an end word, a stream that stops mid-block — kernel == oracle restatement of
    res/horz_pack_yuv.fs.glsl:94-127, word for word."""
    _, g = gpu.geom_of(synth.synthetic_jpeg(200, 120, "420"))
    rng = np.random.default_rng(11)
    slots = gpu.block_slots(g)
    nblk = sum(len(ipos) for ipos, _ in slots)
    words, starts = [], []
    for b in range(nblk):
        starts.append(len(words))
        kind = b % 5
        n = int(rng.integers(0, 70))
        w = rng.integers(0, 65536, size=1 + n).astype(np.uint16)          # anything goes
        if kind == 1:
            w[1:] = (rng.integers(0, 3, size=n) << 12) | rng.integers(1, 4096, size=n)
        elif kind == 2:
            w[1:] = 0xF000                                                  # ZRL chain
        elif kind == 3:
            w = np.concatenate([w[:1], ((np.zeros(63, int) << 12) | 0x801).astype(np.uint16)])
        words.extend(w.tolist())
        if kind != 3:
            words.append(0)
    words = np.array(words, np.uint16)
    index = np.zeros(int(gpu.L.jga_index_count(C.byref(g))), np.int32)
    order = rng.permutation(nblk)                # blocks need not be stored in scan order
    k = 0
    for ipos, _ in slots:
        index[ipos] = np.array(starts)[order[k:k + len(ipos)]]
        k += len(ipos)
    for limit in (len(words), len(words) - 37):  # whole stream, then cut short
        got = gpu.gpu_unpack(g, [words], [index], pack_words=limit)[0]
        want = np.zeros_like(got)
        for ipos, off in slots:
            want[off[:, None] + np.arange(64)] = orc.unpack_blocks(words[:limit], index[ipos])
        assert np.array_equal(got, want), limit


def test_gpu_unpack_4k_then_rgb(gpu, orc, synth):
    """BASELINE headline geometry: words -> planes -> RGB equals the oracle's whole path."""
    data = synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234)
    _, g = gpu.geom_of(data)
    pack, index, _ = gpu.entropy_decode_pack(data, g)
    assert np.array_equal(gpu.gpu_unpack(g, [pack], [index])[0], oracle_quant(orc, data))


# ---- GPU entropy stage ---------------------------------------------

@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("ri", [0, -1, 3])
def test_gpu_huffman_equals_oracle_quant_stage(gpu, orc, synth, sampling, ri):
    """Scan decoded on the GPU == the ORACLE's QUANT planes (and, with it, the host stage's)."""
    datas = [synth.synthetic_jpeg(333, 211, sampling, quality=q, restart_interval=ri, seed=q)
             for q in (90, 60, 30)]
    g, coefs, rounds = gpu.gpu_entropy_decode(datas)
    for d, c in zip(datas, coefs):
        assert np.array_equal(c, oracle_quant(orc, d)), (sampling, ri)
        assert np.array_equal(c, gpu.entropy_decode(d, g)), (sampling, ri)


@pytest.mark.parametrize("sampling", ["420", "422", "444", "grey", "411"])
def test_gpu_huffman_into_a_dirty_buffer(gpu, synth, sampling):
    """The decode clears no plane (round 3): into a buffer full of garbage it must still leave the
    host stage's buffer, byte for byte — real blocks written whole or zeroed before their pieces
    land, the slots at the end of decimated planes that hold no block cleared — whether or not it
    was told that other decodes share the device (which changes the kernels of the later rounds)."""
    datas = [synth.synthetic_jpeg(349, 227, sampling, quality=q, restart_interval=ri, seed=q)
             for q, ri in ((90, 0), (40, 0))]
    for shared in (0, 1):
        hb = gpu.HuffBatch(len(datas), sum(map(len, datas)) + 8192)
        try:
            g = hb.prepare(datas)
            gpu.L.jga_huff_set_device_shared(hb.ptr, shared)
            stride = (g.coef_shorts + 127) // 128 * 128
            d = gpu.DeviceBuffer(stride * 2 * len(datas))
            try:
                for fill in (0x5A5A, -1):
                    d.upload(np.full(stride * len(datas), fill, np.int16))
                    hb.decode(d.ptr, stride)
                    got = d.download(dtype=np.int16).reshape(len(datas), stride)[:, :g.coef_shorts]
                    for i, data in enumerate(datas):
                        assert np.array_equal(got[i], gpu.entropy_decode(data, g)), (sampling, shared, fill, i)
            finally:
                d.free()
        finally:
            hb.close()


@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("ri", [0, -1, 1, 3])
def test_gpu_unstuffing_then_huffman_equals_oracle_quant_stage(gpu, orc, synth, sampling, ri):
    """The scan cleaned up ON THE DEVICE (csrc/unstuff_kernels.hip: stuffed zeros, fill bytes,
    RSTn markers -> restart segments) and decoded there == the ORACLE's QUANT planes; sizes that
    span several 4 KB chunks, with restart intervals from one MCU to a row to none."""
    datas = [synth.synthetic_jpeg(333, 211, sampling, quality=q, restart_interval=ri, seed=q)
             for q in (95, 60, 30)]
    g, coefs, rounds = gpu.gpu_entropy_decode(datas, device_unstuff=True)
    for d, c in zip(datas, coefs):
        assert np.array_equal(c, oracle_quant(orc, d)), (sampling, ri)


def test_gpu_unstuffing_full_size_golden_and_flat(gpu, orc, synth, golden_jpegs):
    """8K with a restart interval per MCU row (BASELINE config 5), the golden files (Pillow's
    optimised tables and DRI), and flat frames whose streams need the host's walk — which then
    has to fetch the clean streams and segment tables from the device."""
    data = synth.synthetic_jpeg(7680, 4320, "420", quality=90, restart_interval=-1, seed=5)
    g, coefs, _ = gpu.gpu_entropy_decode([data], device_unstuff=True)
    real = gpu.real_coef_mask(g)
    assert np.array_equal(coefs[0][real], oracle_quant(orc, data)[real])
    for name in golden_jpegs.names:
        g, coefs, _ = gpu.gpu_entropy_decode([golden_jpegs.jpeg(name)], device_unstuff=True)
        assert np.array_equal(coefs[0], golden_jpegs[name + ".quant"]), name
    assisted = 0
    for pattern in range(3):
        lv = np.zeros(synth.coef_shorts(1920, 1080, "420"), np.int16)
        if pattern == 1:
            lv.reshape(-1, 64)[:, 0] = 5
            lv.reshape(-1, 64)[::2, 0] = -5
        elif pattern == 2:
            lv.reshape(-1, 64)[:, 63] = 1
        flat = synth.encode_levels(lv, 1920, 1080, "420")
        g, coefs, _ = gpu.gpu_entropy_decode([flat, flat], device_unstuff=True)
        assisted += gpu.gpu_entropy_decode.assisted
        assert np.array_equal(coefs[1], oracle_quant(orc, flat)), pattern
    assert assisted > 0


def _letterboxed(gpu, synth, w, h, sampling, seed):
    """A synthetic photograph whose top and bottom quarters are flat (all levels zero)."""
    data = synth.synthetic_jpeg(w, h, sampling, quality=90, seed=seed)
    hdr, g = gpu.geom_of(data)
    lv = gpu.entropy_decode(data, g).copy()
    for p in range(g.nplanes):
        pl = g.plane[p]
        plane = lv[pl.coef_off:pl.coef_off + pl.hblocks * pl.vblocks * 64].reshape(-1, pl.hblocks * 64)
        rows = plane.shape[0]
        plane[:rows // 4] = 0
        plane[rows - rows // 4:] = 0
    return synth.encode_levels(lv, w, h, sampling, qtab=gpu.qtab_of(hdr))


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
def test_gpu_huffman_periodic_streams_that_never_fall_into_step(gpu, orc, synth, sampling):
    """Flat data parses out of step for ever (no self-synchronisation): the rounds alone would
    need one run per subsequence; the host walks those stretches (huff_api.cpp assist_chains)
    and the result is still the host entropy stage's."""
    w, h = 1920, 1080
    n = synth.coef_shorts(w, h, sampling)
    cases = {}
    lv = np.zeros(n, np.int16)
    cases["zero"] = lv.copy()
    lv.reshape(-1, 64)[:, 0] = 5
    lv.reshape(-1, 64)[::2, 0] = -5
    cases["alternating dc"] = lv.copy()
    lv = np.zeros(n, np.int16)
    lv.reshape(-1, 64)[:, 63] = 1
    cases["zrl zrl zrl + last ac"] = lv
    assisted = 0
    for name, lv in cases.items():
        data = synth.encode_levels(lv, w, h, sampling)
        g, coefs, rounds = gpu.gpu_entropy_decode([data])
        assert np.array_equal(coefs[0], oracle_quant(orc, data)), (sampling, name)
        assisted += gpu.gpu_entropy_decode.assisted
    assert assisted > 0                      # at least one of them needed the host's walk
    # an ordinary photograph never does
    data = synth.synthetic_jpeg(w, h, sampling, quality=90, seed=3)
    g, coefs, rounds = gpu.gpu_entropy_decode([data])
    assert np.array_equal(coefs[0], oracle_quant(orc, data))
    assert gpu.gpu_entropy_decode.assisted == 0 and rounds <= 8


def test_gpu_huffman_letterboxed_frames_in_a_batch_and_through_the_pipeline(gpu, orc, synth):
    """Photographs with flat bars (video frames): mixed in one batch with ordinary ones."""
    from jpeg_gpu_amd import abi
    datas = [_letterboxed(gpu, synth, 1280, 720, "420", seed=s) for s in (1, 2)]
    datas.insert(1, synth.synthetic_jpeg(1280, 720, "420", quality=90, seed=9))
    g, coefs, rounds = gpu.gpu_entropy_decode(datas)
    for d, c in zip(datas, coefs):
        assert np.array_equal(c, oracle_quant(orc, d))
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2,
                      batch=4, depth=2)
    try:
        outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in datas]
        rc, _ = pl.run(datas, host_outs=outs)
        assert rc == 0
        for d, o in zip(datas, outs):
            assert np.array_equal(o, orc.decode_rgb(d)[1].ravel())
    finally:
        pl.close()


def test_gpu_huffman_block_finished_with_the_next_intervals_bits_is_an_early_end(gpu, synth):
    """A restart interval cut short by a few bits: the symbol that would complete its last block
    reaches into the next interval.  The host stage calls that "entropy data ended early"; so
    must the GPU stage (found by tools/fuzz_gpu_huff.py 41)."""
    d = bytearray(synth.synthetic_jpeg(205, 126, "grey", quality=70, restart_interval=3, seed=190))
    d[5526] ^= 1 << 5
    d[5023] = 30
    del d[1876]
    del d[2596]
    d = bytes(d)
    _, g = gpu.geom_of(d)
    with pytest.raises(gpu.JgaError, match="ended early"):
        gpu.entropy_decode(d, g)
    with pytest.raises(gpu.JgaError, match="ended early"):
        gpu.gpu_entropy_decode([d])


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
@pytest.mark.parametrize("ri", [0, 5])
def test_gpu_huffman_several_symbols_per_lookup_edges(gpu, orc, synth, sampling, ri):
    """Both entropy stages take several short AC symbols per table look-up (GPU: the packs of the
    synchronisation runs, csrc/huff_common.h; host: csrc/entropy.c build_pairs).  Levels made for the edges —
    blocks that run to coefficient 63 with no EOB (what follows is the next block's DC code), EOBs first and
    second in a step, ZRLs, long magnitudes (tests/test_entropy.py: pair_edge_levels): GPU planes = host planes
    = the oracle's, and through the pipeline the pixels are the oracle's."""
    import oracle
    from test_entropy import pair_edge_levels
    from jpeg_gpu_amd import abi
    w, h = 416, 240                                        # enough scan for several subsequences per segment
    lv = pair_edge_levels(synth, w, h, sampling, seed=19)
    data = synth.encode_levels(lv.reshape(-1), w, h, sampling, restart_interval=ri)
    _, g = gpu.geom_of(data)
    want = orc.decode(data, oracle.QUANT)[1]
    assert np.array_equal(gpu.entropy_decode(data, g, False), want)
    _, coefs, _ = gpu.gpu_entropy_decode([data, data])
    assert np.array_equal(coefs[0], want) and np.array_equal(coefs[1], want)
    rgb = orc.decode_rgb(data)[1]
    for transport in (2, 0, 1):
        out = np.zeros(rgb.size, np.uint8)
        pl = gpu.Pipeline(device=0, nthreads=2, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=transport, batch=2, depth=1)
        try:
            rc, jobs = pl.run([data], host_outs=[out])
            assert rc == 0 and jobs[0].status == 0
        finally:
            pl.close()
        assert np.array_equal(out, rgb.ravel()), transport


@pytest.mark.parametrize("sampling", ["420", "444"])
def test_gpu_huffman_small_and_large_batches_with_mixed_tables(gpu, orc, synth, sampling):
    """Batches of at most four images decode their synchronisation rounds with 12-bit AC packs (hj_wide_ac, round 5),
    larger ones with the 9-bit tables; every image brings its own tables.  The same files — the Annex-K tables, and
    the same two AC tables the other way round (luma coded with the chroma table: another set of packs) — mixed in
    one batch, as batches of 1, 2, 4 (wide) and 5, 7 (9-bit): the oracle's QUANT planes every time."""
    files = [synth.synthetic_jpeg(640, 360, sampling, quality=q, restart_interval=ri, seed=q, flags=fl)
             for q, ri, fl in ((90, 0, 0), (60, 0, synth.SWAP_AC), (35, 7, 0), (90, 0, synth.SWAP_AC), (75, 0, 0))]
    want = [oracle_quant(orc, f) for f in files]
    for n in (1, 2, 4, 5, 7):
        order = [(3 * k + n) % len(files) for k in range(n)]
        _, coefs, _ = gpu.gpu_entropy_decode([files[i] for i in order])
        for k, i in enumerate(order):
            assert np.array_equal(coefs[k], want[i]), (sampling, n, k, i)


def test_gpu_huffman_golden_jpegs(gpu, golden_jpegs):
    """Pillow-made files (optimised tables, DRI) + ours, against the reference's QUANT planes."""
    for name in golden_jpegs.names:
        g, coefs, _ = gpu.gpu_entropy_decode([golden_jpegs.jpeg(name)])
        assert np.array_equal(coefs[0], golden_jpegs[name + ".quant"]), name


def test_gpu_huffman_full_size_and_end_to_end(gpu, orc, synth):
    """4K 4:2:0 (no DRI) and 8K 4:2:0 with DRI per MCU row (BASELINE config 5): scan
    decoded on the GPU, then the fused RGB kernel; equal to the oracle's whole decode."""
    import ctypes as C
    for w, h, ri in ((3840, 2160, 0), (7680, 4320, -1)):
        data = synth.synthetic_jpeg(w, h, "420", quality=90, restart_interval=ri, seed=5)
        hb = gpu.HuffBatch(1, len(data) + 4096)
        g = hb.prepare([data])
        stride = (g.coef_shorts * 2 + 255) // 256 * 128
        d_coef = gpu.DeviceBuffer(stride * 2)
        d_q = gpu.DeviceBuffer(3 * 64 * 2)
        d_rgb = gpu.DeviceBuffer(g.rgb_bytes)
        d_coef.upload(np.full(stride, 0x5A5A, np.int16))       # every real block must be written
        rounds = hb.decode(d_coef.ptr, stride)
        real = gpu.real_coef_mask(g)
        assert np.array_equal(d_coef.download(dtype=np.int16)[:g.coef_shorts][real],
                              oracle_quant(orc, data)[real])
        d_q.upload(hb.qtabs())
        gpu.check(gpu.L.jga_idct_rgb_batch(C.byref(g), 1, d_coef.ptr, stride, d_q.ptr, 1,
                                           d_rgb.ptr, g.rgb_bytes, None))
        gpu.check(gpu.L.jga_stream_sync(None))
        assert np.array_equal(d_rgb.download(), orc.decode_rgb(data)[1].reshape(-1))
        assert 1 <= rounds < 64
        hb.close()
        for b in (d_coef, d_q, d_rgb):
            b.free()


def test_dense_frames_settle_on_the_device_without_the_hosts_walk(gpu, orc, synth):
    """Round 6's policy sweep: a q97 8K frame (1.2 bytes per pixel: 310 k subsequences, chains of thirty steps and
    more) did not settle within twelve rounds of which each list round is ONE step of the chain, and the host's walk
    — meant for streams that never fall into step — took over: 6.5 ms where six in-group steps needed 0.9.  The walk
    is now taken only when the work lists stop shrinking: the frame settles on the device (no subsequence walked by
    the host), in more than twelve rounds, to the host stage's planes; and a stream that really never settles still
    gets the walk."""
    data = synth.synthetic_jpeg(7680, 4320, "420", quality=97, seed=11)
    assert len(data) > 20 << 20
    hb = gpu.HuffBatch(1, len(data) + 4096)
    g = hb.prepare([data])
    stride = (g.coef_shorts * 2 + 255) // 256 * 128
    d_coef = gpu.DeviceBuffer(stride * 2)
    rounds = hb.decode(d_coef.ptr, stride)
    real = gpu.real_coef_mask(g)
    assert np.array_equal(d_coef.download(dtype=np.int16)[:g.coef_shorts][real], gpu.entropy_decode(data, g)[real])
    assert hb.assisted() == 0 and 12 < rounds < 96, (hb.assisted(), rounds)
    hb.close()
    d_coef.free()
    # (the never-settling kind: a DC value alternating between two levels over a flat 4K frame)
    _, g4 = gpu.geom_of(synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1))
    levels = np.zeros(g4.coef_shorts, np.int16)
    levels.reshape(-1, 64)[:, 0] = 5
    levels.reshape(-1, 64)[::2, 0] = -5                                         # (tools/periodic_streams.py: "dc alternating")
    flat = synth.encode_levels(levels, 3840, 2160, "420")
    hb = gpu.HuffBatch(1, len(flat) + 4096)
    gpu.L.jga_huff_set_device_shared(hb.ptr, 2)                                 # (list rounds: the progress test applies)
    g = hb.prepare([flat])
    stride4 = (g.coef_shorts * 2 + 255) // 256 * 128
    d_coef = gpu.DeviceBuffer(stride4 * 2)
    hb.decode(d_coef.ptr, stride4)
    real4 = gpu.real_coef_mask(g)
    assert np.array_equal(d_coef.download(dtype=np.int16)[:g.coef_shorts][real4], gpu.entropy_decode(flat, g)[real4])
    assert hb.assisted() > 0
    hb.close()
    d_coef.free()


def test_gpu_huffman_rejects_truncated_scan(gpu, synth):
    data = synth.synthetic_jpeg(320, 200, "420", seed=3)
    with pytest.raises(gpu.JgaError):
        gpu.gpu_entropy_decode([data[:len(data) // 2]])


def test_irregular_huffman_tables_take_the_host_entropy_stage(gpu, orc, synth):
    """Tables outside the device lookup format: the GPU entropy stage refuses them by name;
    the plugin and the transport-2 pipeline decode such a file with the host entropy stage
    (same coefficients) and the GPU kernels — results equal the oracle."""
    from jpeg_gpu_amd import abi
    from conftest import three_table_variant
    odd = synth.synthetic_jpeg(320, 200, "420", quality=85, seed=3, flags=synth.FLAT_AC)
    odd3 = three_table_variant(synth.synthetic_jpeg(320, 200, "420", quality=85, seed=4))   # Cr on tables of its own
    for f in (odd, odd3):
        with pytest.raises(gpu.JgaError, match="too irregular"):
            gpu.gpu_entropy_decode([f])
        want = orc.decode_rgb(f)[1]
        with gpu.Decoder(f) as d:
            d.read_header()
            d.init_image()
            d.decode(abi.JPEG_DECODE_RGB)
            assert np.array_equal(d.pixels(), want)
    normal = [synth.synthetic_jpeg(320, 200, "420", quality=85, seed=10 + i) for i in range(5)]
    datas = normal[:2] + [odd] + normal[2:4] + [odd3] + normal[4:]
    outs = [np.zeros(320 * 200 * 3, np.uint8) for _ in datas]
    pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2,
                      batch=7, depth=1)
    try:
        rc, jobs = pl.run(datas, host_outs=outs)
        assert rc == 0 and all(j.status == 0 for j in jobs)
        for o, dd in zip(outs, datas):
            assert np.array_equal(o, orc.decode_rgb(dd)[1].reshape(-1))
    finally:
        pl.close()


@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("ri", [0, -1, 3])
def test_split_decode_dc_values_beside_the_planes(gpu, orc, synth, sampling, ri):
    """jga_huff_decode_split: the planes keep the DC DIFFERENCES, the DC values (src/xjpeg.c:480)
    come in an array of their own, one per 128-byte slot of the coefficient buffer, and
    jga_idct_rgb_batch_dc / jga_idct_yuv_batch_dc take them from there: same pixels and planes
    as the oracle, the DC array equal to the oracle's QUANT planes' DC positions, for every
    sampling, with and without restart intervals (every interval starts its predictors at 0),
    several images per batch."""
    import oracle
    files = [synth.synthetic_jpeg(200 + 16 * i, 120, sampling, quality=60 + 10 * i, restart_interval=ri, seed=70 + i)
             for i in range(1)] * 1 + [synth.synthetic_jpeg(200, 120, sampling, quality=q, restart_interval=ri, seed=80 + q)
                                       for q in (35, 95)]
    files = files[1:]                                       # (one geometry per batch)
    n = len(files)
    hb = gpu.HuffBatch(n, sum(map(len, files)) + 4096 * n)
    g = hb.prepare(files)
    cs = gpu._align(g.coef_shorts * 2) // 2
    dcs = (g.coef_shorts // 64 + 127) // 128 * 128
    d_coef, d_dc, d_q = gpu.DeviceBuffer(cs * 2 * n), gpu.DeviceBuffer(dcs * 2 * n), gpu.DeviceBuffer(384 * n)
    ob, yb = gpu._align(g.rgb_bytes), gpu._align(g.yuv_bytes, 256)
    d_rgb, d_yuv = gpu.DeviceBuffer(ob * n), gpu.DeviceBuffer(yb * n)
    try:
        d_coef.upload(np.full(cs * n, 0x5A5A, np.int16))
        d_dc.upload(np.full(dcs * n, 0x7B7B, np.int16))
        hb.decode_split(d_coef.ptr, cs, d_dc.ptr, dcs)
        d_q.upload(hb.qtabs())
        gpu.check(gpu.L.jga_idct_rgb_batch_dc(C.byref(g), n, d_coef.ptr, cs, d_dc.ptr, dcs, d_q.ptr, 1, d_rgb.ptr, ob, None))
        gpu.check(gpu.L.jga_idct_yuv_batch_dc(C.byref(g), n, d_coef.ptr, cs, d_dc.ptr, dcs, d_q.ptr, 1, d_yuv.ptr, yb, None))
        gpu.check(gpu.L.jga_stream_sync(None))
        dc = d_dc.download(dtype=np.int16).reshape(n, dcs)
        planes_dev = d_coef.download(dtype=np.int16).reshape(n, cs)
        m = gpu.real_coef_mask(g)
        slots = np.flatnonzero(m[::64])                      # buffer slots that hold a block
        for i, f in enumerate(files):
            quant = orc.decode(f, oracle.QUANT)[1]
            assert np.array_equal(dc[i][slots], quant[::64][slots]), (sampling, ri, i)
            ac = np.ones(g.coef_shorts, bool)
            ac[::64] = False                                 # everything but the DC positions is final
            assert np.array_equal(planes_dev[i][:g.coef_shorts][m & ac], quant[m & ac])
            assert np.array_equal(d_rgb.download(g.rgb_bytes, offset=i * ob), orc.decode_rgb(f)[1].reshape(-1))
            _, planes = orc.decode(f, oracle.YUV)
            want = np.concatenate([p.reshape(-1) for p in planes])
            assert np.array_equal(d_yuv.download(g.yuv_bytes, offset=i * yb), want)
        # ... and the plain call puts them in place
        hb.decode(d_coef.ptr, cs)
        full = d_coef.download(dtype=np.int16).reshape(n, cs)
        for i, f in enumerate(files):
            assert np.array_equal(full[i][:g.coef_shorts][m], orc.decode(f, oracle.QUANT)[1][m])
    finally:
        hb.close()
        for b in (d_coef, d_dc, d_q, d_rgb, d_yuv):
            b.free()


@pytest.mark.parametrize("nstreams", [0, 1])
@pytest.mark.parametrize("sampling,ri", [("420", 0), ("420", 9), ("444", 0), ("422", -1), ("grey", 3)])
def test_files_read_where_they_lie_among_copied_ones(gpu, orc, synth, sampling, ri, nstreams):
    """Clean-up on the device: files the caller flags are read where they lie, one copy call each — at whatever
    alignment their scans start — on the decode's own stream or on the copy stream the caller names
    (jga_huff_set_copy_stream); the others are copied into the blob by the host and go up from there.  Same QUANT
    planes as the oracle's for every mix, on a second decode of the same prepared batch, and for a later ordinary
    prepare on the same batch object."""
    import ctypes as C
    datas = [synth.synthetic_jpeg(333, 211, sampling, quality=40 + 5 * i, restart_interval=ri, seed=70 + i)
             for i in range(11)]
    want = [oracle_quant(orc, d) for d in datas]
    # (every pinned copy starts at another offset inside its buffer)
    pins = [gpu.PinnedBytes(b"\0" * (i % 7) + d) for i, d in enumerate(datas)]
    addr = lambda i: pins[i].array.ctypes.data + i % 7
    hb = gpu.HuffBatch(len(datas), sum(map(len, datas)) + 4096 * len(datas), device_unstuff=True)
    streams = [gpu.L.jga_stream_create() for _ in range(nstreams)]
    d = None
    try:
        gpu.L.jga_huff_set_copy_stream.argtypes = [C.c_void_p, C.c_void_p]
        gpu.L.jga_huff_set_copy_stream(hb.ptr, streams[0] if streams else None)
        for flags in ([1] * 11, [i % 2 for i in range(11)], [int(i % 3 == 0) for i in range(11)], [0] * 11):
            gpu.L.jga_huff_set_input_flags(hb.ptr, bytes(flags), len(datas))
            g = hb.prepare_at([addr(i) if flags[i] else np.frombuffer(datas[i], np.uint8).ctypes.data
                               for i in range(len(datas))], [len(x) for x in datas])
            cs = gpu._align(g.coef_shorts * 2) // 2
            if d is None:
                d = gpu.DeviceBuffer(cs * 2 * len(datas))
            m = gpu.real_coef_mask(g)
            for rep in range(2):
                d.upload(np.full(cs * len(datas), 0x5A5A, np.int16))
                hb.decode(d.ptr, cs)
                got = d.download(dtype=np.int16).reshape(len(datas), cs)
                for i in range(len(datas)):
                    assert np.array_equal(got[i][:g.coef_shorts][m], want[i][m]), (flags, rep, i)
        gpu.L.jga_huff_set_input_flags(hb.ptr, None, 0)
        gpu.L.jga_huff_set_copy_stream(hb.ptr, None)
        g = hb.prepare(datas)
        hb.decode(d.ptr, cs)
        got = d.download(dtype=np.int16).reshape(len(datas), cs)
        assert all(np.array_equal(got[i][:g.coef_shorts][m], want[i][m]) for i in range(len(datas)))
    finally:
        hb.close()
        for st in streams:
            gpu.L.jga_stream_destroy(st)
        if d is not None:
            d.free()
        for p in pins:
            p.free()


def test_device_cleanup_with_a_damaged_member(gpu, orc, synth):
    """A file with restart intervals that breaks off in mid-scan (an EOI where data should be) inside a batch whose
    files are read where they lie: the decode reports that member alone and the others' planes are the oracle's."""
    datas = [synth.synthetic_jpeg(320, 200, "420", quality=80, seed=i, restart_interval=(i % 2) * 11) for i in range(9)]
    bad = bytearray(datas[7])
    bad[len(bad) // 2:len(bad) // 2 + 2] = b"\xff\xd9"
    datas[7] = bytes(bad[:len(bad) // 2 + 2])
    verdicts = [0] * 7 + [1, 0]
    pins = [gpu.PinnedBytes(d) for d in datas]
    hb = gpu.HuffBatch(len(datas), sum(map(len, datas)) + 4096 * len(datas), device_unstuff=True)
    try:
        gpu.L.jga_huff_set_input_flags(hb.ptr, bytes([1] * 9), 9)
        g = hb.prepare_at([p.array.ctypes.data for p in pins], [len(x) for x in datas])
        cs = gpu._align(g.coef_shorts * 2) // 2
        d = gpu.DeviceBuffer(cs * 2 * len(datas))
        m = gpu.real_coef_mask(g)
        with pytest.raises(gpu.JgaError):
            hb.decode(d.ptr, cs)
        assert [int(gpu.L.jga_huff_image_error(hb.ptr, i) != 0) for i in range(9)] == verdicts
        got = d.download(dtype=np.int16).reshape(len(datas), cs)
        for i, f in enumerate(datas):
            if i != 7:
                assert np.array_equal(got[i][:g.coef_shorts][m], oracle_quant(orc, f)[m]), i
        d.free()
    finally:
        hb.close()
        for p in pins:
            p.free()


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("more", [{}, {"unstuff": 1}, {"unstuff": 2}, {"input_cache_mb": -1}, {"input_cache_mb": -2}])
def test_pipeline_short_and_long_runs(gpu, orc, synth, pinned, more):
    """A run that gives every lane a group or two takes the route with the fewest host steps (clean-up on the
    device, the files read where they lie, one wait per group); a long
    one the steady-state route.  Same pixels as the oracle's either way, for pinned files and ordinary ones (held by
    the input cache, named in copy calls, or copied), one run after the other on the same pipeline."""
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(1280, 720, "420", quality=92, seed=200 + i, restart_interval=(i % 3 == 0) * 40)
             for i in range(12)]
    _, g = gpu.geom_of(datas[0])
    want = [orc.decode_rgb(d)[1].reshape(-1) for d in datas]
    pins = [gpu.PinnedBytes(d) for d in datas] if pinned else []
    src = [p.array for p in pins] if pinned else datas
    pl = gpu.Pipeline(device=0, nthreads=6, out=abi.JPEG_DECODE_RGB, copy_back=True, transport=2, batch=16, depth=3,
                      **more)
    try:
        for n in (72, 5, 1, 160, 9):                     # 8 / 0.6 / 0.1 / 18 / 1 frame equivalents
            outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in range(n)]
            jobs = gpu.Pipeline.make_jobs([src[i % 12] for i in range(n)], host_outs=outs, pinned=pinned)
            assert pl.run_jobs(jobs) == 0
            for i in range(n):
                assert np.array_equal(outs[i], want[i % 12]), (n, i)
    finally:
        pl.close()
        for p in pins:
            p.free()


def test_decode_in_two_halves_with_the_block_decode_in_between(gpu, orc, synth):
    """jga_huff_decode_split_begin / _end: the block-decode kernel is queued between the two halves, the host waits
    once, and _end says that what was queued in between saw the final planes (a photograph-like stream settles
    within the rounds queued up front).  Pixels = the oracle's, on a repeat decode too; _end without _begin fails."""
    import ctypes as C
    datas = [synth.synthetic_jpeg(640, 360, "420", quality=88, seed=300 + i, restart_interval=(i % 2) * 13) for i in range(5)]
    n = len(datas)
    hb = gpu.HuffBatch(n, sum(map(len, datas)) + 4096 * n)
    g = hb.prepare(datas)
    cs = gpu._align(g.coef_shorts * 2) // 2
    dcs = (g.coef_shorts // 64 + 127) // 128 * 128
    os_ = gpu._align(g.rgb_bytes)
    d_coef, d_dc, d_rgb = gpu.DeviceBuffer(cs * 2 * n), gpu.DeviceBuffer(dcs * 2 * n), gpu.DeviceBuffer(os_ * n)
    try:
        with pytest.raises(gpu.JgaError):
            hb.decode_split_end()
        for rep in range(2):
            d_rgb.fill(0x11)
            hb.decode_split_begin(d_coef.ptr, cs, d_dc.ptr, dcs)
            q = hb.qtabs_device()
            assert q
            gpu.check(gpu.L.jga_idct_rgb_batch_dc(C.byref(g), n, d_coef.ptr, cs, d_dc.ptr, dcs, q, 1, d_rgb.ptr, os_, None))
            rounds, valid = hb.decode_split_end()
            assert valid and rounds >= 1
            got = d_rgb.download().reshape(n, os_)
            for i, f in enumerate(datas):
                assert np.array_equal(got[i][:g.rgb_bytes], orc.decode_rgb(f)[1].reshape(-1)), (rep, i)
    finally:
        hb.close()
        for b in (d_coef, d_dc, d_rgb):
            b.free()


def test_speculative_tail_that_ran_too_early_is_undone(gpu):
    """The tail of a decode (prefix sums, write pass, DC pass) is queued behind the first group of
    rounds without waiting for them to settle.  With ONE round per group and one in-group iteration
    (JGA_HUFF_ITERS=1,1,1: a knob of the tuning build, read once per process — hence the child on
    libjpeg_gpu_amd_tuning.so) the first tail always runs on
    unsettled states: outputs and verdicts must be reset and the decode must still end equal to
    the oracle — planes, DC array, pixels — on the second decode of the same batch object too
    (which queues more rounds first), for clean and for damaged members."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, %r)
import oracle
from jpeg_gpu_amd import lib, synth
orc = oracle.Oracle()
files = [synth.synthetic_jpeg(640, 360, "420", quality=q, restart_interval=ri, seed=s) for q, ri, s in ((90, 0, 1), (50, 7, 2), (95, -1, 3))]
bad = bytearray(files[0]); sos = bad.find(b"\xff\xda"); del bad[sos + 200:sos + 1200]; bad = bytes(bad)   # a kilobyte of entropy-coded data missing
for batch, damaged in ((files, None), (files[:2] + [bad], 2)):
    hb = lib.HuffBatch(len(batch), sum(map(len, batch)) + 4096 * len(batch))
    g = hb.prepare(batch)
    cs = lib._align(g.coef_shorts * 2) // 2
    d = lib.DeviceBuffer(cs * 2 * len(batch))
    m = lib.real_coef_mask(g)
    for rep in range(3):
        d.upload(np.full(cs * len(batch), 0x5A5A, np.int16))
        try:
            rounds = hb.decode(d.ptr, cs)
            assert damaged is None, "a damaged member must be reported"
        except lib.JgaError:
            assert damaged is not None and lib.L.jga_huff_image_error(hb.ptr, damaged) != 0
            assert all(lib.L.jga_huff_image_error(hb.ptr, i) == 0 for i in range(len(batch)) if i != damaged)
        got = d.download(dtype=np.int16).reshape(len(batch), cs)
        for i, f in enumerate(batch):
            if i == damaged: continue
            want = orc.decode(f, oracle.QUANT)[1]
            assert np.array_equal(got[i][:g.coef_shorts][m], want[m]), (rep, i)
    if damaged is None:
        # the same through the two halves: a block decode queued in between ran on unsettled planes at least once
        # (the first tail of a fresh batch object always does here) — _end must say so, and the redo must be right
        hb2 = lib.HuffBatch(len(batch), sum(map(len, batch)) + 4096 * len(batch))
        hb2.prepare(batch)
        dcs = (g.coef_shorts // 64 + 127) // 128 * 128
        osz = lib._align(g.rgb_bytes)
        ddc, drgb = lib.DeviceBuffer(dcs * 2 * len(batch)), lib.DeviceBuffer(osz * len(batch))
        seen_invalid = False
        for rep in range(3):
            hb2.decode_split_begin(d.ptr, cs, ddc.ptr, dcs)
            q = hb2.qtabs_device()
            lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), len(batch), d.ptr, cs, ddc.ptr, dcs, q, 1, drgb.ptr, osz, None))
            rounds, valid = hb2.decode_split_end()
            if not valid:
                seen_invalid = True
                lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), len(batch), d.ptr, cs, ddc.ptr, dcs, q, 1, drgb.ptr, osz, None))
                lib.check(lib.L.jga_stream_sync(None))
            px = drgb.download().reshape(len(batch), osz)
            for i, f in enumerate(batch):
                assert np.array_equal(px[i][:g.rgb_bytes], orc.decode_rgb(f)[1].reshape(-1)), ("halves", rep, i)
        assert seen_invalid, "one round per burst: the first tail cannot have been final"
        hb2.close(); ddc.free(); drgb.free()
    hb.close(); d.free()
print("OK", lib.L.jga_huff_last_rounds.__name__)
""" % ROOT
    env = dict(os.environ, JGA_HUFF_ITERS="1,1,1",
               JGA_LIB_PATH=os.path.join(ROOT, "jpeg_gpu_amd", "libjpeg_gpu_amd_tuning.so"))
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_tables_repeated_under_a_third_id_stay_on_the_gpu_entropy_stage(gpu, orc, synth):
    """An encoder that writes one DHT per component gives Cr tables byte-identical to Cb's under
    an id of their own: the device format tells tables apart by content, so the frame keeps
    the GPU entropy stage (same planes as the oracle), alone and next to ordinary files."""
    import oracle
    from conftest import three_table_variant
    base = synth.synthetic_jpeg(320, 200, "420", quality=85, seed=4)
    same = three_table_variant(base, distinct=False)
    assert same != base
    g, got, _ = gpu.gpu_entropy_decode([same, base])
    m = gpu.real_coef_mask(g)
    want = orc.decode(same, oracle.QUANT)[1]
    assert np.array_equal(got[0][m], want[m]) and np.array_equal(got[1][m], want[m])


def test_pipeline_writes_evenly_spaced_destinations_in_one_launch(gpu, orc, synth):
    """Jobs whose dev_out pointers are slices of one buffer at a fixed pitch (a caller that keeps
    every output) are written by the group's single launch, like the lane's own buffer; unevenly
    placed ones one by one.  Same pixels either way, nothing outside the slices is touched."""
    from jpeg_gpu_amd import abi
    files = [synth.synthetic_jpeg(200, 120, "420", quality=70 + i, seed=50 + i) for i in range(6)]
    nb = 200 * 120 * 3
    pitch = gpu._align(nb) + 256
    big = gpu.DeviceBuffer(pitch * 6)
    odd = gpu.DeviceBuffer(pitch * 7)
    pl = gpu.Pipeline(device=0, nthreads=2, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2,
                      batch=6, depth=1)
    try:
        for buf, ptrs in ((big, [big.ptr + i * pitch for i in range(6)]),
                          (odd, [odd.ptr + (i + (i > 2)) * pitch for i in range(6)])):   # a gap after the third
            buf.fill(0xA5)
            rc, jobs = pl.run(files, dev_outs=ptrs)
            assert rc == 0 and all(j.status == 0 for j in jobs)
            raw = buf.download()
            for i, f in enumerate(files):
                o = ptrs[i] - buf.ptr
                assert np.array_equal(raw[o:o + nb], orc.decode_rgb(f)[1].reshape(-1)), i
                assert np.all(raw[o + nb:o + pitch] == 0xA5)
    finally:
        pl.close()
        big.free()
        odd.free()


@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("size", [(17, 9), (100, 75), (642, 363)])
def test_pass3_alone_matches_oracle(gpu, orc, synth, sampling, size):
    """jga_yuv_rgb_batch (res/yuv.fs.glsl:16-24 / unyuv.fs.glsl on u8 planes): planes in HBM
    -> pixels, for every sampling and ragged sizes, two images per launch."""
    import oracle
    datas = [synth.synthetic_jpeg(size[0], size[1], sampling, quality=q, seed=q) for q in (90, 40)]
    _, g = gpu.geom_of(datas[0])
    ys, os_ = gpu._align(g.yuv_bytes), gpu._align(g.rgb_bytes)
    d_yuv, d_rgb = gpu.DeviceBuffer(ys * 2), gpu.DeviceBuffer(os_ * 2)
    try:
        for i, d in enumerate(datas):
            _, planes = orc.decode(d, oracle.YUV)
            d_yuv.upload(np.concatenate([p.reshape(-1) for p in planes]), offset=i * ys)
        gpu.check(gpu.L.jga_yuv_rgb_batch(C.byref(g), 2, d_yuv.ptr, ys, d_rgb.ptr, os_, None))
        gpu.check(gpu.L.jga_stream_sync(None))
        for i, d in enumerate(datas):
            got = d_rgb.download(g.rgb_bytes, offset=i * os_)
            assert np.array_equal(got, orc.decode_rgb(d)[1].reshape(-1)), (sampling, size, i)
    finally:
        d_yuv.free()
        d_rgb.free()


@pytest.mark.parametrize("sampling", ["420", "grey", "411"])
@pytest.mark.parametrize("size", [(1, 1), ("max", 9), (9, "max")])
def test_extreme_dimensions_every_path(gpu, orc, synth, sampling, size):
    """The smallest frame and the largest ones whose MCU-padded planes still fit image.h's
    unsigned-short plane dimensions (src/image.h:31-32): plugin with the GPU entropy stage,
    PACK expansion + fused kernel, and the host-entropy pipeline all give the oracle's pixels."""
    from jpeg_gpu_amd import abi
    mcu = {"420": (16, 16), "grey": (8, 8), "411": (32, 8)}[sampling]
    w, h = [(65535 // mcu[i]) * mcu[i] if v == "max" else v for i, v in enumerate(size)]
    data = synth.synthetic_jpeg(w, h, sampling, quality=75, seed=7, restart_interval=-1)
    want = orc.decode_rgb(data)[1].reshape(-1)
    with gpu.Decoder(data) as d:                      # GPU entropy stage + fused kernel
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)
        assert np.array_equal(d.pixels().reshape(-1), want)
    _, g = gpu.geom_of(data)
    pack, index, _ = gpu.entropy_decode_pack(data, g)  # PACK words expanded on the GPU
    assert np.array_equal(gpu.gpu_unpack(g, [pack], [index])[0], oracle_quant(orc, data))
    out = np.zeros(want.size, np.uint8)
    pl = gpu.Pipeline(device=0, nthreads=1, out=abi.JPEG_DECODE_RGB, copy_back=True)
    try:
        rc, _ = pl.run([data], host_outs=[out])
        assert rc == 0 and np.array_equal(out, want)
    finally:
        pl.close()


def test_gpu_huffman_on_corrupted_scans_agrees_with_the_host_stage(gpu, orc, synth):
    """Random byte edits, bit flips and deletions inside the entropy-coded data: the GPU
    entropy stage always returns (no hang, no crash), rejects exactly the files the host stage
    rejects, and produces the same coefficients for the ones both accept.  What to do with
    damaged data is the build's own definition (the reference reads such files unvalidated,
    SURVEY.md Appendix E), so acceptance is compared with the host stage; for every file the
    ORACLE decodes as well the coefficients must be the oracle's."""
    vs_oracle = 0
    rng = np.random.default_rng(5)
    accepted = rejected = 0
    for it in range(160):
        samp = ["420", "444", "grey", "422"][it % 4]
        d = bytearray(synth.synthetic_jpeg(200 + it % 37, 120 + it % 23, samp, quality=70,
                                           restart_interval=[0, 3, -1][it % 3], seed=it))
        lo = d.find(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(lo, len(d) - 2))
            mode = int(rng.integers(0, 3))
            if mode == 0:
                d[pos] = int(rng.integers(0, 256))
            elif mode == 1:
                d[pos] ^= 1 << int(rng.integers(0, 8))
            else:
                del d[pos]
        d = bytes(d)
        try:
            _, g = gpu.geom_of(d)
        except gpu.JgaError:
            continue
        try:
            want = gpu.entropy_decode(d, g)
        except gpu.JgaError:
            want = None
        try:
            got = gpu.gpu_entropy_decode([d], device_unstuff=bool(it & 1))[1][0]    # both clean-ups
        except gpu.JgaError:
            got = None
        assert (got is None) == (want is None), it
        if got is not None:
            assert np.array_equal(got, want), it
            accepted += 1
            try:
                oq = oracle_quant(orc, d)
            except ValueError:
                oq = None
            if oq is not None:
                real = gpu.real_coef_mask(g)
                assert np.array_equal(got[real], oq[real]), it
                vs_oracle += 1
        else:
            rejected += 1
    assert accepted > 20 and rejected > 20 and vs_oracle > 10
