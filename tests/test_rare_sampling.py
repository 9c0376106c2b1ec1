"""Sampling factors beyond the usual six.  The reference takes any of 1, 2, 4 per axis and
component (src/xjpeg.c:384-391, decimation from the largest: src/image.c:49-55) and its pass 3 is
generic in xdec / ydec (res/unyuv.fs.glsl:30-31): luma 4x2, 2x4, 1x4 are ordinary files, luma 4x4
(18 blocks per MCU) is more than T.81 B.2.3 allows but is decoded all the same.  The files of
tests/golden/jpegs_rare.npz also run through every golden-driven test (conftest.GoldenJpegs);
here: the 18-block files, and the device stages at sizes with tile tails."""
import numpy as np
import pytest

RARE = [(4, 2), (2, 4), (1, 4), (4, 4)]
# Cb and Cr with their own factors (luma still the finest plane): res/unyuv.fs.glsl:6-9 takes
# u_xdec/u_ydec and v_xdec/v_ydec separately, src/jpeg_gpu.c:868-877
MIXED = [((2, 2), (2, 1), (1, 1)), ((2, 1), (1, 1), (2, 1)), ((2, 2), (1, 2), (2, 1)),
         ((4, 1), (1, 1), (2, 1)), ((2, 2), (2, 2), (1, 1)), ((4, 2), (2, 1), (1, 2))]


def rgb_of(orc, data):
    return orc.decode_rgb(data)[1].reshape(-1)


# ---- without a GPU ---------------------------------------------------------------------------

def test_mcu18_goldens_oracle_and_host_stages(orc, lib, golden_mcu18):
    import oracle
    G = golden_mcu18
    assert len(G.names) == 2
    for name in G.names:
        data = G.jpeg(name)
        info, quant = orc.decode(data, oracle.QUANT)
        assert info.as_dict() == G.info(name), name
        assert (quant == G[name + ".quant"]).all(), name
        assert (orc.decode(data, oracle.DCT)[1] == G[name + ".dct"]).all(), name
        for a, b in zip(orc.decode(data, oracle.YUV)[1], G.planes(name)):
            assert (a == b).all(), name
        h, g = lib.geom_of(data)
        assert [(g.plane[p].xdec, g.plane[p].ydec) for p in range(3)] == [(0, 0), (2, 2), (2, 2)]
        assert (lib.entropy_decode(data, g, False) == G[name + ".quant"]).all(), name
        assert (lib.entropy_decode(data, g, True) == G[name + ".dct"]).all(), name
        pack, index, _ = lib.entropy_decode_pack(data, g)
        assert (pack == G[name + ".pack"]).all() and (index == G[name + ".index"]).all(), name


@pytest.mark.parametrize("samp", RARE + MIXED)
@pytest.mark.parametrize("size", [(8, 8), (150, 90), (33, 65)])
def test_oracle_equals_reference(orc, ref, synth, samp, size):
    import oracle
    for ri in (0, -1, 3):
        data = synth.synthetic_jpeg(size[0], size[1], samp, quality=80, restart_interval=ri, seed=3)
        assert orc.parse(data).as_dict() == ref.parse(data).as_dict()
        for mode in (oracle.QUANT, oracle.DCT):
            assert (orc.decode(data, mode)[1] == ref.decode(data, mode)[1]).all()
        for a, b in zip(orc.decode(data, oracle.YUV)[1], ref.decode(data, oracle.YUV)[1]):
            assert (a == b).all()


def test_gpu_entropy_stage_verdicts(synth):
    """Ten blocks per MCU fit the device format, eighteen do not: that file is the host entropy
    stage's (verdict 2, like tables outside the lookup format), not an error."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["bash", os.path.join(root, "tools", "build_emul.sh")], check=True, capture_output=True)
    emul = C.CDLL(os.path.join(root, "tools", "bin", "libhuff_emul.so"))
    for samp, want in (((4, 2), 0), ((2, 4), 0), ((1, 4), 0), ((4, 4), 2)):
        d = synth.synthetic_jpeg(100, 60, samp, quality=80, seed=1)
        assert emul.huff_emul_prepare_head(d, len(d)) == want, samp


@pytest.mark.parametrize("samp", [(4, 2), MIXED[0], MIXED[3]])
def test_plugin_host_stages(lib, orc, synth, samp):
    """The plugin instance's CPU-side stages (PACK / QUANT / DCT) for a rare sampling."""
    import oracle
    from jpeg_gpu_amd import abi
    data = synth.synthetic_jpeg(90, 70, samp, quality=75, restart_interval=2, seed=9)
    with lib.Decoder(data) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_QUANT)
        assert (d.coef() == orc.decode(data, oracle.QUANT)[1]).all()
        d.reset()
        d.read_header()
        d.decode(abi.JPEG_DECODE_DCT)
        assert (d.coef() == orc.decode(data, oracle.DCT)[1]).all()


# ---- on the GPU --------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("samp", RARE + MIXED)
@pytest.mark.parametrize("size", [(8, 8), (17, 9), (150, 90), (640, 360), (1031, 517)])
def test_kernels_match_oracle(gpu, orc, synth, samp, size):
    from test_gpu_parity import check_image
    data = synth.synthetic_jpeg(size[0], size[1], samp, quality=90, seed=size[0])
    check_image(gpu, orc, data)
    if size == (150, 90):
        check_image(gpu, orc, data, dequant=False)


@pytest.mark.gpu
def test_mcu18_goldens_device_stages(gpu, golden_mcu18):
    """The compiled reference's QUANT planes -> its Y/Cb/Cr planes, on the device."""
    from test_gpu_parity import run_device
    G = golden_mcu18
    for name in G.names:
        data = G.jpeg(name)
        h, g = gpu.geom_of(data)
        yuv = run_device(gpu, g, G[name + ".quant"], gpu.qtab_of(h), rgb=False)
        for a, b in zip(gpu.split_planes(g, yuv), G.planes(name)):
            assert np.array_equal(a, b), name
        # and its PACK words + index -> its QUANT planes (18 slots: two descriptor words)
        got = gpu.gpu_unpack(g, [G[name + ".pack"]], [G[name + ".index"]])
        assert np.array_equal(got[0], G[name + ".quant"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("samp", RARE + MIXED)
def test_plugin_and_pipeline(gpu, orc, synth, samp):
    """decode_image(YUV / RGB) through the plugin instance, and the three pipeline transports
    (the GPU entropy stage for the samplings it can hold, the host stage for luma 4x4)."""
    import oracle
    from jpeg_gpu_amd import abi
    datas = [synth.synthetic_jpeg(200 + 16 * i, 120, samp, quality=60 + 10 * i,
                                  restart_interval=[0, -1, 5][i], seed=40 + i) for i in range(3)]
    datas += datas[:2]                                   # same-geometry groups for transport 2
    want = [rgb_of(orc, d) for d in datas]
    with gpu.Decoder(datas[0]) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_YUV)
        for a, b in zip(d.planes(), orc.decode(datas[0], oracle.YUV)[1]):
            assert np.array_equal(a, b)
        d.reset()
        d.read_header()
        d.decode(abi.JPEG_DECODE_RGB)
        assert np.array_equal(d.pixels().reshape(-1), want[0])
    for transport in (0, 1, 2):
        outs = [np.zeros(w.size, np.uint8) for w in want]
        pl = gpu.Pipeline(device=0, nthreads=4, out=abi.JPEG_DECODE_RGB, copy_back=True,
                          transport=transport, batch=2, depth=2)
        try:
            rc = pl.run_jobs(gpu.Pipeline.make_jobs(datas, host_outs=outs))
        finally:
            pl.close()
        assert rc == 0, (transport, gpu.L.jga_last_error())
        for o, w in zip(outs, want):
            assert np.array_equal(o, w), transport


@pytest.mark.gpu
def test_gpu_entropy_stage(gpu, orc, synth):
    """Ten-block MCUs through the GPU Huffman decoder (both clean-ups) against the oracle's QUANT
    planes; the eighteen-block one is turned away with the host-stage verdict."""
    import oracle
    def blocks_per_mcu(samp):
        return sum(h * v for h, v in samp) if isinstance(samp[0], tuple) else samp[0] * samp[1] + 2
    assert [blocks_per_mcu(s) for s in RARE] == [10, 10, 6, 18]
    for samp in [s for s in RARE + MIXED if blocks_per_mcu(s) <= 10]:
        for ri in (0, -1, 3):
            datas = [synth.synthetic_jpeg(333, 222, samp, quality=q, restart_interval=ri, seed=q)
                     for q in (35, 92)]
            for dev in (False, True):
                g, coefs, _ = gpu.gpu_entropy_decode(datas, device_unstuff=dev)
                real = gpu.real_coef_mask(g)
                for d, c in zip(datas, coefs):
                    assert np.array_equal(c[real], orc.decode(d, oracle.QUANT)[1][real]), (samp, ri, dev)
    for samp in [s for s in RARE + MIXED if blocks_per_mcu(s) > 10]:      # luma 4x4; 4x2 + 2x1 + 1x2
        with pytest.raises(gpu.JgaError, match="GPU entropy stage"):
            gpu.gpu_entropy_decode([synth.synthetic_jpeg(64, 64, samp, seed=1)])
