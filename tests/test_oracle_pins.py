"""Pin the oracle (oracle/oracle.c): constants, known-answer vectors, committed
golden vectors produced by the compiled reference, the reference's own IEEE-1180
unit test procedure, and — when oracle/_ref is present — the reference itself."""
import struct

import numpy as np
import pytest


def f32hex(x):
    return struct.pack(">f", float(x)).hex()


def test_constants_bit_patterns(orc):
    """SURVEY.md Appendix A.2/A.3 float32 patterns of src/dct.c:51,62-65,89-98."""
    c = orc.constants()
    want = ["3eb504f3", "3efb14be", "3eec835e", "3ed4db31", "3eb504f3", "3e8e39da",
            "3e43ef15", "3dc7c5c2", "3fb504f3", "3fec835e", "3f8a8bd4", "40273d75"]
    assert [f32hex(v) for v in c] == want


def test_dezigzag_is_t81_figure_a6(orc):
    dz = orc.dezigzag()
    assert sorted(dz) == list(range(64))
    assert list(dz[:12]) == [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25]
    assert list(dz[-4:]) == [47, 55, 62, 63]


def _blk(**kv):
    b = np.zeros(64, np.int16)
    for k, v in kv.items():
        b[int(k[1:])] = v
    return b


def test_known_answer_vectors(orc):
    """SURVEY.md Appendix C (captured from the compiled reference)."""
    out = orc.idct_blocks([_blk(i0=-1), _blk(i0=-4), _blk(i0=8)])
    assert (out[0] == 0).all() and (out[1] == 0).all() and (out[2] == 1).all()
    row = [17, 15, 10, 3, -3, -10, -15, -17]
    o = orc.idct_blocks([_blk(i1=100)])[0].reshape(8, 8)
    assert all(list(r) == row for r in o)
    o = orc.idct_blocks([_blk(i8=100)])[0].reshape(8, 8)
    assert all(list(o[:, c]) == row for c in range(8))
    o = orc.idct_blocks([_blk(i0=1016, i9=55, i63=-37)])[0].reshape(8, 8)
    assert list(o[0]) == [140, 139, 133, 131, 123, 121, 115, 114]
    assert list(o[7]) == [114, 115, 121, 123, 131, 133, 139, 140]
    dense_in = np.array(
        [-58, 126, 385, -405, -181, -5, -30, -261, 20, 54, -3, 31, 60, -36, 33, -57,
         -63, -1, -15, 30, 22, 50, -49, 7, -1, -2, -1, 1, 2, 0, 4, 1,
         2, -2, -5, 3, -6, -3, -7, 0, 4, -4, -3, 7, -1, 6, -3, 3,
         2, -6, 1, 0, 4, -5, 2, -2, 6, 7, 1, 1, 7, 7, 7, -2], np.int16)
    dense_out = np.array(
        [9, 97, -12, 20, -172, -62, -42, 56, 5, 109, 9, 11, -180, -60, -47, 71,
         7, 117, 33, 8, -175, -43, -42, 96, -15, 130, 49, -6, -171, -36, -22, 100,
         -16, 122, 60, -8, -179, -43, -17, 103, -35, 104, 44, -17, -181, -59, 4, 92,
         -34, 83, 31, -11, -188, -80, 28, 60, -50, 80, 14, -14, -189, -94, 37, 47], np.int16)
    assert (orc.idct_blocks([dense_in])[0] == dense_out).all()


def test_golden_blocks(orc, golden_blocks):
    inp, out = golden_blocks
    assert (orc.idct_blocks(inp) == out).all()


def test_golden_jpegs(orc, golden_jpegs):
    import oracle
    for name in golden_jpegs.names:
        data = golden_jpegs.jpeg(name)
        info, quant = orc.decode(data, oracle.QUANT)
        want = golden_jpegs.info(name)
        got = info.as_dict()
        assert got == want, name
        assert (quant == golden_jpegs[name + ".quant"]).all(), name
        _, dct = orc.decode(data, oracle.DCT)
        assert (dct == golden_jpegs[name + ".dct"]).all(), name
        _, planes = orc.decode(data, oracle.YUV)
        for a, b in zip(planes, golden_jpegs.planes(name)):
            assert (a == b).all(), name
        # the device-stage restatement (dequant + idct + clamp on QUANT planes)
        for a, b in zip(orc.coef_to_planes(info, quant, True), golden_jpegs.planes(name)):
            assert (a == b).all(), name
        for a, b in zip(orc.coef_to_planes(info, dct, False), golden_jpegs.planes(name)):
            assert (a == b).all(), name


# ---- the reference's own unit test: IEEE-1180-1990 (test/dct.c:229-261) --------

def ieee1180_blocks(n, lo, hi, sign, seed=1):
    """LCG of test/dct.c:70-81 driving the spec's random block generator."""
    x = seed
    out = np.empty(n * 64, np.int64)
    span = hi - lo + 1
    for i in range(n * 64):
        x = (x * 1103515245 + 12345) & 0xFFFFFFFF
        v = (x & 0x7FFFFFFE) / float(0x7FFFFFFF)
        out[i] = (int(v * span) + lo) * sign
    return out.reshape(n, 8, 8)


def dct_matrix():
    k = np.arange(8)
    m = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * 0.5
    m[0] *= 1 / np.sqrt(2)
    return m          # X = m @ x @ m.T ; x = m.T @ X @ m


def ieee1180_check(idct_fn, n=2000):
    m = dct_matrix()
    stats = []
    for lo, hi in ((-256, 255), (-5, 5), (-300, 300)):
        for sign in (1, -1):
            px = ieee1180_blocks(n, lo, hi, sign).astype(np.float64)
            coef = np.clip(np.rint(np.einsum("ij,njk,lk->nil", m, px, m)), -2048, 2047)
            ref = np.clip(np.rint(np.einsum("ji,njk,kl->nil", m, coef, m)), -256, 255)
            got = idct_fn(coef.astype(np.int16).reshape(-1, 64)).reshape(-1, 8, 8)
            got = np.clip(got, -256, 255)
            err = got - ref
            stats.append(dict(peak=np.abs(err).max(), pmse=(err ** 2).mean(0).max(),
                              omse=(err ** 2).mean(), pme=np.abs(err.mean(0)).max(),
                              ome=abs(err.mean())))
    return stats


def assert_ieee1180(stats):
    for s in stats:     # bounds asserted by test/dct.c:182, 200, 204, 222, 226
        assert s["peak"] <= 1
        assert s["pmse"] <= 0.015
        assert s["omse"] <= 0.02
        assert s["pme"] <= 0.015
        assert s["ome"] <= 0.0015


def test_ieee1180_accuracy(orc):
    assert_ieee1180(ieee1180_check(orc.idct_blocks))
    assert (orc.idct_blocks(np.zeros((1, 64), np.int16)) == 0).all()   # dct.c test 257-260


# ---- against the compiled reference itself (present in the build container) -----

def test_oracle_equals_reference_on_random_blocks(orc, ref):
    rng = np.random.default_rng(1)
    for lo, hi in ((-32768, 32768), (-2048, 2048), (-16, 17)):
        b = rng.integers(lo, hi, (50000, 64)).astype(np.int16)
        assert (orc.idct_blocks(b) == ref.idct_blocks(b)).all()


@pytest.mark.parametrize("sampling,size", [("grey", (57, 31)), ("444", (40, 24)),
                                           ("422", (81, 50)), ("420", (321, 203)),
                                           ("440", (24, 48)), ("411", (97, 33))])
def test_oracle_equals_reference_on_images(orc, ref, synth, sampling, size):
    import oracle
    for ri, flags in ((0, 0), (-1, 0), (5, synth.DQT16 | synth.SPLIT_DHT)):
        data = synth.synthetic_jpeg(size[0], size[1], sampling, quality=85,
                                    restart_interval=ri, seed=3, flags=flags)
        assert orc.parse(data).as_dict() == ref.parse(data).as_dict()
        for mode in (oracle.QUANT, oracle.DCT):
            assert (orc.decode(data, mode)[1] == ref.decode(data, mode)[1]).all()
        for a, b in zip(orc.decode(data, oracle.YUV)[1], ref.decode(data, oracle.YUV)[1]):
            assert (a == b).all()


def test_rgb_stage_definition(orc):
    """The RGB stage is defined by the build: spot values."""
    import oracle
    info = oracle.Info()
    info.width, info.height, info.ncomps = 2, 2, 3
    for i in range(3):
        info.hblocks[i] = info.vblocks[i] = 1
    y = np.zeros((8, 8), np.uint8)
    u = np.full((8, 8), 128, np.uint8)
    v = np.full((8, 8), 128, np.uint8)
    y[0, 0], y[0, 1], y[1, 0], y[1, 1] = 0, 255, 100, 100
    u[1, 0], v[1, 0] = 255, 0          # strong blue, no red
    u[1, 1], v[1, 1] = 0, 255
    rgb = orc.planes_to_rgb(info, [y, u, v])
    assert list(rgb[0, 0]) == [0, 0, 0] and list(rgb[0, 1]) == [255, 255, 255]
    # R = 100 + 1.402*(-128) <0 -> 0 ; G = (100 - .34414*127) - .71414*(-128) ; B = 100+1.772*127 -> 255
    assert list(rgb[1, 0]) == [0, int(np.float32(np.float32(100 + np.float32(-0.34414) * 127) +
                                                 np.float32(-0.71414) * np.float32(-128)) + 0.5), 255]
    assert rgb[1, 1, 0] == 255 and rgb[1, 1, 2] == 0


def test_pack_consumer_restatement_against_reference_goldens(orc, lib, golden_jpegs):
    """oracle.c:orc_unpack_blocks (res/horz_pack_yuv.fs.glsl:94-127) applied to the PACK words
    and block index the COMPILED REFERENCE produced gives the reference's own QUANT planes."""
    for name in golden_jpegs.names:
        _, g = lib.geom_of(golden_jpegs.jpeg(name))
        pack, index, quant = (golden_jpegs[name + k] for k in (".pack", ".index", ".quant"))
        got = np.zeros_like(quant)
        for ipos, off in lib.block_slots(g):
            blocks = orc.unpack_blocks(pack, index[ipos])
            got[off[:, None] + np.arange(64)] = blocks
        assert np.array_equal(got, quant), name


@pytest.mark.parametrize("sampling", ["444", "422", "420"])
def test_rgb_stage_definition_is_sane_against_libjpeg_turbo(orc, synth, sampling):
    """The reference has no CPU code for upsample + YCbCr->RGB, so oracle.c's
    restatement of res/unyuv.fs.glsl IS the definition of that stage.  Cross-check, with a
    tolerance and not for equality: libjpeg-turbo (through Pillow, when it is installed) decodes
    the same files with a different IDCT (ISLOW integer), fancy upsampling off, to pixels that
    differ by rounding only."""
    PIL = pytest.importorskip("PIL.Image")
    import io
    data = synth.synthetic_jpeg(320, 200, sampling, quality=92, seed=9)
    _, rgb = orc.decode_rgb(data)
    im = PIL.open(io.BytesIO(data))
    im.draft("RGB", im.size)
    ref = np.asarray(im.convert("RGB"), dtype=np.int16)
    diff = np.abs(rgb.reshape(200, 320, 3).astype(np.int16) - ref)
    # 4:4:4: rounding only.  Subsampled: nearest-neighbour replication here vs libjpeg's
    # triangle filter, on chroma that carries per-pixel noise — a few levels, no pixel far off.
    mean_max, p99_max = (1.0, 4) if sampling == "444" else (3.0, 16)
    assert diff.mean() < mean_max and np.percentile(diff, 99) <= p99_max, (diff.mean(), diff.max())
