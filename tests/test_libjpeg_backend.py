"""LIBJPEG_DECODE_CTX_VTBL (csrc/libjpeg_vtbl.c): the comparison backend, the counterpart of
the reference's libjpeg wrapper (src/jpeg_wrap.c:56-252).  CPU only.

What can be pinned: its QUANT stage is Huffman decoding and nothing else, so it must give
the oracle's (= the compiled reference's) coefficient planes bit for bit.  Its YUV / RGB
stages run libjpeg-turbo's integer IDCT and colour conversion: *parity unpinned*
 — compared with a tolerance only, which is written here."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "jpeg_gpu_amd", "jpeg_gpu_hip")
YUV_TOL = 1          # libjpeg-turbo ISLOW vs src/dct.c: +-1 on a few percent of samples
RGB_TOL = 4          # +-1 per plane through the colour matrix (max gain 1.772) + rounding


@pytest.fixture(scope="module")
def ljp(lib):
    if not lib.L.jga_libjpeg_available():
        pytest.skip("libjpeg.so.8 is not installed on this box")
    return lib


def true_region(info, i, plane):
    h = (info.height + (1 << info.ydec[i]) - 1) >> info.ydec[i]
    w = (info.width + (1 << info.xdec[i]) - 1) >> info.xdec[i]
    return plane[:h, :w].astype(int)


@pytest.mark.parametrize("restart", [0, 3])
@pytest.mark.parametrize("sampling,size", [("grey", (65, 33)), ("444", (100, 75)), ("422", (91, 50)),
                                           ("420", (200, 120)), ("420", (123, 77)),
                                           ("440", (50, 91)), ("411", (130, 40))])
def test_every_stage_against_the_oracle(ljp, orc, synth, sampling, size, restart):
    import oracle
    from jpeg_gpu_amd import abi
    data = synth.synthetic_jpeg(size[0], size[1], sampling, quality=88, seed=9,
                                restart_interval=restart)
    with ljp.Decoder(data, ljp.LIBJPEG_VTBL) as d:
        hdr = d.read_header()
        info, want = orc.decode(data, oracle.QUANT)
        assert (hdr.width, hdr.height, hdr.ncomps) == (info.width, info.height, info.ncomps)
        assert hdr.restart_interval == restart
        for i in range(info.ncomps):
            c = hdr.comp[i]
            assert (c.hblocks, c.vblocks, c.hsamp, c.vsamp) == (
                info.hblocks[i], info.vblocks[i], info.hsamp[i], info.vsamp[i])
            assert np.array_equal(np.ctypeslib.as_array(c.quant.contents.tbl), info.qtab()[i])
        assert hdr.subsamp == ljp.parse_header(data).subsamp
        d.init_image()
        # QUANT: bit-exact (pinned through the oracle)
        d.decode(abi.JPEG_DECODE_QUANT)
        real = ljp.real_coef_mask(ljp.geom_from_header(hdr))
        assert np.array_equal(d.coef()[real], want[real])
        # YUV / RGB: the reference's own call order per frame (reset -> header -> image)
        d.reset(); d.read_header()
        d.decode(abi.JPEG_DECODE_YUV)
        _, planes = orc.decode(data, oracle.YUV)
        for i, (a, b) in enumerate(zip(d.planes(), planes)):
            assert np.abs(true_region(info, i, a) - true_region(info, i, b)).max() <= YUV_TOL
        d.reset(); d.read_header()
        d.decode(abi.JPEG_DECODE_RGB)
        _, rgb = orc.decode_rgb(data)
        px = d.pixels()
        assert np.abs(px.astype(int) - rgb.reshape(px.shape)).max() <= RGB_TOL


def test_golden_quant_planes(ljp, golden_jpegs):
    """The compiled reference's own QUANT planes (tests/golden) from the libjpeg backend."""
    from jpeg_gpu_amd import abi
    for name in golden_jpegs.names:
        data = golden_jpegs.jpeg(name)
        with ljp.Decoder(data, ljp.LIBJPEG_VTBL) as d:
            hdr = d.read_header()
            d.init_image()
            d.decode(abi.JPEG_DECODE_QUANT)
            real = ljp.real_coef_mask(ljp.geom_from_header(hdr))
            assert np.array_equal(d.coef()[real], golden_jpegs[name + ".quant"][real]), name


def test_unsupported_stages_and_damaged_files(ljp, synth):
    """PACK/DCT are refused with the reference's message (src/jpeg_wrap.c:224-228); library
    errors come back as EXIT_FAILURE instead of ending the process."""
    from jpeg_gpu_amd import abi
    data = synth.synthetic_jpeg(64, 48, "420", seed=1)
    with ljp.Decoder(data, ljp.LIBJPEG_VTBL) as d:
        d.read_header(); d.init_image()
        for stage, name in ((abi.JPEG_DECODE_PACK, "pack"), (abi.JPEG_DECODE_DCT, "dct")):
            with pytest.raises(ljp.JgaError, match="Unsupported output '%s' for libjpeg wrapper" % name):
                d.decode(stage)
    with ljp.Decoder(b"\xff\xd8\xff\xe0 not a jpeg at all" * 4, ljp.LIBJPEG_VTBL) as d:
        with pytest.raises(ljp.JgaError):
            d.read_header()
    big = synth.synthetic_jpeg(256, 192, "420", seed=1)
    cut = big[:len(big) // 2]                      # header intact, scan cut short: still returns
    with ljp.Decoder(cut, ljp.LIBJPEG_VTBL) as d:
        d.read_header(); d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)              # libjpeg pads a short scan (warning, not error)


def test_harness_offers_the_backend(ljp, synth, tmp_path, golden_jpegs):
    """`-i libjpeg` (src/jpeg_gpu.c:545-557): header text and the quant dump equal the default
    backend's; pack is refused."""
    name = golden_jpegs.names[0]
    p = tmp_path / "g.jpg"
    p.write_bytes(golden_jpegs.jpeg(name))
    env = dict(os.environ, JGA_QUIET="0")

    def run(*a):
        return subprocess.run([EXE] + list(a) + [str(p)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, timeout=120, env=env)
    a, b = run("-H"), run("-H", "-i", "libjpeg")
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout
    a, b = run("-d", "-o", "quant"), run("-d", "-o", "quant", "-i", "libjpeg")
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout
    r = run("-d", "-o", "pack", "-i", "libjpeg")
    assert r.returncode == 1 and "Unsupported output 'pack' for libjpeg wrapper." in r.stderr
    r = run("-i", "libjpeg", "-o", "rgb", "--no-gpu", "--frames", "5")
    assert r.returncode == 0 and r.stdout.startswith("5 FPS (cpu ")
