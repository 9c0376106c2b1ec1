"""BASELINE.json's configs as tests (configs 1 and 4 were run by no test).

config 1  single 512x512 greyscale baseline JPEG through the CPU path (xjpeg.c + dct.c):
          plumbing, no GPU — the compiled reference, the oracle restatement and the
          plugin's host stages must agree byte for byte; on the GPU box the plugin's
          YUV / RGB stages are then held against the same planes.
config 4  a batch of 1024 x 1080p 4:2:0 files, image-sharded over 8 GPUs with no
          collective: here the whole batch and one rank's 128-image shard on ONE GPU,
          every image checked against the oracle's pixels by Adler-32.
(configs 2, 3, 5 and the 4K headline: tests/test_gpu_parity.py test_full_size_configs_match_oracle,
 test_gpu_huffman_full_size_and_end_to_end, test_4k_420_properties.)"""
import ctypes as C
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest


# ---- config 1 ---------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def grey512(synth):
    return synth.synthetic_jpeg(512, 512, "grey", quality=90, seed=1234)


def test_config1_grey_512_cpu_plumbing(lib, orc, ref, grey512):
    """SURVEY.md §8a sizes, and reference == oracle == plugin host stages for every CPU stage."""
    import oracle
    from jpeg_gpu_amd import abi
    hdr, g = lib.geom_of(grey512)
    assert (g.width, g.height, g.nplanes, g.subsamp) == (512, 512, 1, abi.JPEG_SUBSAMP_MONO)
    assert g.coef_blocks == 4096 and g.coef_shorts * 2 == 524288          # §8 a3, a6
    assert g.yuv_bytes == 262144 and g.rgb_bytes == 262144                # 1 B/px grey (§8 a9)
    for stage, ours in ((oracle.QUANT, abi.JPEG_DECODE_QUANT), (oracle.DCT, abi.JPEG_DECODE_DCT)):
        _, want = ref.decode(grey512, stage)
        assert np.array_equal(orc.decode(grey512, stage)[1], want)
        with lib.Decoder(grey512) as d:                                   # the vtable, host stage
            d.read_header()
            d.init_image()
            d.decode(ours)
            assert np.array_equal(d.coef(), want)
    rinfo, rplanes = ref.decode(grey512, oracle.YUV)                      # xjpeg.c + src/dct.c
    oinfo, oplanes = orc.decode(grey512, oracle.YUV)
    assert len(rplanes) == 1 and rplanes[0].shape == (512, 512)
    assert np.array_equal(oplanes[0], rplanes[0])
    words, index = ref.decode(grey512, oracle.PACK)[1]
    pack, idx, _ = lib.entropy_decode_pack(grey512, g)
    assert np.array_equal(pack, words) and np.array_equal(idx, index)
    # greyscale "RGB" is the Y plane at the true size (res/ungrey.fs.glsl:6-19)
    assert np.array_equal(orc.decode_rgb(grey512)[1], oplanes[0][:512, :512])


@pytest.mark.gpu
def test_config1_grey_512_through_the_plugin(gpu, orc, grey512):
    """alloc -> header -> image_init -> image(QUANT | YUV | RGB), reset between frames
    (src/jpeg_gpu.c:612-613, 1231-1237): device stages equal the CPU path's planes."""
    import oracle
    from jpeg_gpu_amd import abi
    _, planes = orc.decode(grey512, oracle.YUV)
    with gpu.Decoder(grey512) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_QUANT)
        assert np.array_equal(d.coef(), orc.decode(grey512, oracle.QUANT)[1])
        d.reset(); d.read_header()
        d.decode(abi.JPEG_DECODE_YUV)
        assert np.array_equal(d.planes()[0], planes[0])
        d.reset(); d.read_header()
        d.decode(abi.JPEG_DECODE_RGB)
        assert np.array_equal(d.pixels(), planes[0])


# ---- config 4 ---------------------------------------------------------------------------------

N_BATCH, W4, H4 = 1024, 1920, 1080


@pytest.fixture(scope="module")
def batch_1080p(synth, orc):
    """1024 distinct 1080p 4:2:0 q90 files (seeds 0..1023) and the Adler-32 of
    the oracle's RGB for each."""
    workers = max(1, min(64, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        datas = list(ex.map(lambda s: synth.synthetic_jpeg(W4, H4, "420", quality=90, seed=s),
                            range(N_BATCH)))
        sums = list(ex.map(lambda d: zlib.adler32(orc.decode_rgb(d)[1].tobytes()), datas))
    return datas, sums


@pytest.mark.gpu
def test_config4_1024_x_1080p_kernel_batch_on_one_gpu(gpu, batch_1080p):
    """All 1024 coefficient buffers resident in HBM, ONE launch of the fused kernel
    (SURVEY.md §8a sizes: 6 266 880 B of planes + 6 220 800 B of pixels per image)."""
    datas, sums = batch_1080p
    _, g = gpu.geom_of(datas[0])
    assert g.coef_shorts * 2 == 6266880 and g.rgb_bytes == 6220800 and g.coef_blocks == 48960
    cstride = gpu._align(g.coef_shorts * 2) // 2
    ostride = gpu._align(g.rgb_bytes)
    d_coef, d_q = gpu.DeviceBuffer(cstride * 2 * N_BATCH), gpu.DeviceBuffer(384 * N_BATCH)
    d_out = gpu.DeviceBuffer(ostride * N_BATCH)
    try:
        with ThreadPoolExecutor(max_workers=max(1, min(32, len(os.sched_getaffinity(0))))) as ex:
            coefs = list(ex.map(lambda d: gpu.entropy_decode(d, g), datas))
        q = np.zeros((N_BATCH, 3, 64), np.uint16)
        for i, d in enumerate(datas):
            d_coef.upload(coefs[i], offset=i * cstride * 2)
            q[i] = gpu.qtab_of(gpu.parse_header(d))
        d_q.upload(q)
        d_out.fill(0xA5)
        gpu.check(gpu.L.jga_idct_rgb_batch(C.byref(g), N_BATCH, d_coef.ptr, cstride, d_q.ptr, 1,
                                           d_out.ptr, ostride, None))
        gpu.check(gpu.L.jga_stream_sync(None))
        for lo in range(0, N_BATCH, 128):                      # download one shard's worth at a time
            raw = d_out.download(ostride * 128, offset=lo * ostride).reshape(128, ostride)
            for i in range(128):
                assert zlib.adler32(raw[i, :g.rgb_bytes].tobytes()) == sums[lo + i], lo + i
    finally:
        d_coef.free(); d_q.free(); d_out.free()


@pytest.mark.gpu
@pytest.mark.parametrize("transport", [2, 0])
def test_config4_batch_and_shard_through_the_pipeline(gpu, batch_1080p, transport):
    """JPEG bytes in host RAM -> pixels: the whole batch with results left in HBM, then rank 3's
    shard of an 8-way split (shard_range: 128 contiguous images, no exchange with anyone)
    with the pixels copied back — GPU entropy stage and the north-star host-Huffman transport."""
    from jpeg_gpu_amd import abi
    from jpeg_gpu_amd.shard import shard_range
    datas, sums = batch_1080p
    _, g = gpu.geom_of(datas[0])
    ostride = gpu._align(g.rgb_bytes)
    n_all = N_BATCH if transport == 2 else 256                 # (host Huffman: a quarter is plenty)
    pl = gpu.Pipeline(device=0, nthreads=min(48, len(os.sched_getaffinity(0))),
                      out=abi.JPEG_DECODE_RGB, copy_back=False, transport=transport, batch=32, depth=4)
    d_out = gpu.DeviceBuffer(ostride * n_all)
    try:
        rc, jobs = pl.run(datas[:n_all], dev_outs=[d_out.ptr + i * ostride for i in range(n_all)])
        assert rc == 0 and all(j.status == 0 for j in jobs)
        assert all((j.width, j.height, j.nplanes) == (W4, H4, 3) for j in jobs)
        for lo in range(0, n_all, 128):
            raw = d_out.download(ostride * 128, offset=lo * ostride).reshape(128, ostride)
            for i in range(128):
                assert zlib.adler32(raw[i, :g.rgb_bytes].tobytes()) == sums[lo + i], lo + i
    finally:
        pl.close()
        d_out.free()
    mine = shard_range(N_BATCH, 3, 8)
    assert len(mine) == 128 and mine[0] == 384
    outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in mine]
    pl = gpu.Pipeline(device=0, nthreads=min(32, len(os.sched_getaffinity(0))),
                      out=abi.JPEG_DECODE_RGB, copy_back=True, transport=transport, batch=32, depth=4)
    try:
        rc, jobs = pl.run([datas[i] for i in mine], host_outs=outs)
        assert rc == 0
        for k, i in enumerate(mine):
            assert zlib.adler32(outs[k].tobytes()) == sums[i], i
    finally:
        pl.close()


# ---- photograph-like content ---------------------------------------------------

def test_photo_like_content_lands_where_photographs_do(synth, orc):
    """The f^-1.5-spectrum recipe at q90 4:2:0: 0.10-0.20 bytes per pixel at 1080p (the SURVEY recipe's sin + N(0,12)
    noise costs 0.37), deterministic by seed, and a valid baseline file the oracle decodes."""
    a = synth.photo_like_jpeg(1920, 1080, "420", 90, seed=1)
    assert 0.10 < len(a) / (1920 * 1080) < 0.20
    assert a == synth.photo_like_jpeg(1920, 1080, "420", 90, seed=1)
    assert a != synth.photo_like_jpeg(1920, 1080, "420", 90, seed=2)
    assert len(synth.synthetic_jpeg(1920, 1080, "420", 90, seed=1)) / (1920 * 1080) > 0.3
    info, rgb = orc.decode_rgb(a)
    assert rgb.shape == (1080, 1920, 3) and 20 < rgb.std() < 90          # a picture, not a flat field


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,samp,ri", [(1920, 1080, "420", 0), (3840, 2160, "420", 0), (1920, 1080, "422", 0),
                                         (1000, 600, "444", -1), (640, 480, "grey", 7)])
def test_photo_like_content_through_every_gpu_route(gpu, orc, synth, w, h, samp, ri):
    """Lighter content exercises what the recipe's noise never does: long zero runs and EOBs early in the block,
    short codes, many blocks per subsequence of the GPU entropy stage.  The plugin (RGB), the batch decoder
    (transport 2: a group of identical-geometry files, twice, clean-up on host and on device) and the host
    entropy stage (transport 0) all equal the oracle bit for bit."""
    from jpeg_gpu_amd import abi
    files = [synth.photo_like_jpeg(w, h, samp, 90, ri, seed=40 + i) for i in range(3)]
    want = [orc.decode_rgb(f)[1].reshape(-1) for f in files]
    with gpu.Decoder(files[0]) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)
        assert np.array_equal(d.pixels().reshape(-1), want[0])
    for more in ({"transport": 2, "unstuff": 1}, {"transport": 2, "unstuff": 2}, {"transport": 0}):
        outs = [np.zeros(want[0].size, np.uint8) for _ in range(6)]
        pl = gpu.Pipeline(device=0, nthreads=3, out=abi.JPEG_DECODE_RGB, copy_back=True, batch=4, **more)
        try:
            rc, jobs = pl.run([files[i % 3] for i in range(6)], host_outs=outs)
            assert rc == 0 and all(j.status == 0 for j in jobs)
            for i in range(6):
                assert np.array_equal(outs[i], want[i % 3]), (more, i)
        finally:
            pl.close()
