"""Host entropy stage (csrc/entropy.c) against the oracle, the committed golden
vectors, a levels round trip, and malformed input."""
import numpy as np
import pytest


def test_golden_jpegs_all_stages(lib, golden_jpegs):
    for name in golden_jpegs.names:
        data = golden_jpegs.jpeg(name)
        h, g = lib.geom_of(data)
        want = golden_jpegs.info(name)
        assert (h.width, h.height, h.ncomps, h.restart_interval) == \
            (want["width"], want["height"], want["ncomps"], want["restart_interval"])
        assert g.coef_shorts == want["coef_shorts"]
        assert (lib.entropy_decode(data, g, False) == golden_jpegs[name + ".quant"]).all(), name
        assert (lib.entropy_decode(data, g, True) == golden_jpegs[name + ".dct"]).all(), name
        pack, index, per = lib.entropy_decode_pack(data, g)
        assert (pack == golden_jpegs[name + ".pack"]).all(), name
        assert (index == golden_jpegs[name + ".index"]).all(), name
        assert sum(per) == len(pack)


@pytest.mark.parametrize("sampling", ["grey", "444", "422", "420", "440", "411"])
@pytest.mark.parametrize("ri", [0, -1, 1, 7])
def test_matches_oracle(lib, orc, synth, sampling, ri):
    import oracle
    data = synth.synthetic_jpeg(123, 77, sampling, quality=80, restart_interval=ri, seed=5)
    h, g = lib.geom_of(data)
    info = orc.parse(data)
    assert (lib.qtab_of(h)[:info.ncomps] == info.qtab()[:info.ncomps]).all()
    assert (lib.entropy_decode(data, g, False) == orc.decode(data, oracle.QUANT)[1]).all()
    assert (lib.entropy_decode(data, g, True) == orc.decode(data, oracle.DCT)[1]).all()


def test_header_fields(lib, synth):
    from jpeg_gpu_amd import abi
    data = synth.synthetic_jpeg(1920, 1080, "420", restart_interval=-1)
    h = lib.parse_header(data)
    assert (h.bits, h.width, h.height, h.ncomps) == (8, 1920, 1080, 3)
    assert h.subsamp == abi.JPEG_SUBSAMP_420 and h.restart_interval == 120
    assert [(h.comp[i].hblocks, h.comp[i].vblocks) for i in range(3)] == \
        [(240, 136), (120, 68), (120, 68)]
    assert h.quant[0].valid and h.quant[1].valid and not h.quant[2].valid
    assert h.quant[0].bits == 8
    h16 = lib.parse_header(synth.synthetic_jpeg(16, 16, "444", flags=synth.DQT16))
    assert h16.quant[0].bits == 16


def test_levels_round_trip(lib, synth):
    """decode(encode(levels)) == levels for random baseline-codable levels,
    incl. long zero runs (ZRL), full-magnitude ACs and 16-bit-wrapping DC sums."""
    rng = np.random.default_rng(11)
    for sampling, (w, h) in (("420", (64, 48)), ("444", (24, 24)), ("grey", (40, 8))):
        n = synth.coef_shorts(w, h, sampling)
        lv = np.zeros(n, np.int16).reshape(-1, 64)
        for b in lv:
            k = rng.integers(0, 20)
            b[rng.integers(1, 64, k)] = rng.integers(-1023, 1024, k)
            b[0] = rng.integers(-1000, 1001)
        lv[3, 1:] = 0
        lv[3, 63] = -1023            # 62 zeros then a value: three ZRLs
        lv[4, :] = 1023
        data = synth.encode_levels(lv.reshape(-1), w, h, sampling, restart_interval=2)
        _, g = lib.geom_of(data)
        got = lib.entropy_decode(data, g, False).reshape(-1, 64)
        # only real blocks are coded; unused packed slots stay zero on both sides
        assert (got == lv).all() or _only_unused_slots_differ(lib, g, got, lv)


def _only_unused_slots_differ(lib, g, got, lv):
    import ctypes as C
    used = np.zeros(len(lv), bool)
    for p in range(g.nplanes):
        for by in range(g.plane[p].vblocks):
            for bx in range(g.plane[p].hblocks):
                used[lib.L.jga_block_offset(C.byref(g), p, bx, by) // 64] = True
    return (got[used] == lv[used]).all() and (got[~used] == 0).all()


def test_dequant_wraps_to_int16(lib, synth):
    """DCT stage = (short)(level*q) (xjpeg.c:501-503, 524-527): 16-bit tables wrap."""
    n = synth.coef_shorts(8, 8, "grey")
    lv = np.zeros(n, np.int16)
    lv[0], lv[1], lv[9] = 1000, -1023, 77
    q = np.full((3, 64), 1, np.uint16)
    q[0, 0], q[0, 1], q[0, 9] = 40000, 65535, 1234
    data = synth.encode_levels(lv, 8, 8, "grey", qtab=q, flags=synth.DQT16)
    _, g = lib.geom_of(data)
    got = lib.entropy_decode(data, g, True)
    want = (lv.astype(np.int64) * q[0].astype(np.int64)).astype(np.int16)
    assert (got[:64] == want).all()


def test_malformed_inputs_fail_cleanly(lib, synth):
    good = synth.synthetic_jpeg(64, 64, "420", restart_interval=2)
    _, g = lib.geom_of(good)
    for bad in (b"", b"\xff\xd8", good[:100], good[:len(good) // 2], b"GIF89a" + good[6:],
                good.replace(b"\xff\xc0", b"\xff\xc2", 1)):
        with pytest.raises(lib.JgaError):
            h = lib.parse_header(bad)
            lib.entropy_decode(bad, lib.geom_from_header(h), False)
    # RST markers out of order
    swapped = good.replace(b"\xff\xd0", b"\xff\xd3", 1)
    with pytest.raises(lib.JgaError):
        lib.entropy_decode(swapped, g, False)
    # random corruption inside the entropy data never crashes
    rng = np.random.default_rng(3)
    sos = good.index(b"\xff\xda")
    for _ in range(200):
        b = bytearray(good)
        for k in rng.integers(sos + 14, len(b) - 2, 4):
            b[k] = rng.integers(0, 256)
        try:
            lib.entropy_decode(bytes(b), g, False)
        except lib.JgaError:
            pass
    # geometry mismatch is rejected
    _, g2 = lib.geom_of(synth.synthetic_jpeg(32, 32, "420"))
    with pytest.raises(lib.JgaError):
        lib.entropy_decode(good, g2, False)


def test_plugin_host_stages(lib, orc, synth):
    """HIPJPEG vtable: alloc -> header -> image_init -> image(QUANT|DCT|PACK) ->
    reset -> header -> image, the reference's call order (jpeg_gpu.c:612-613,
    1231-1237); decode_image before decode_header is refused."""
    import oracle
    from jpeg_gpu_amd import abi
    data = synth.synthetic_jpeg(90, 60, "422", seed=21)
    with lib.Decoder(data) as d:
        hdr = d.read_header()
        assert hdr.comp[0].quant.contents.tbl[0] == hdr.quant[0].tbl[0]
        d.init_image()
        d.decode(abi.JPEG_DECODE_QUANT)
        assert (d.coef() == orc.decode(data, oracle.QUANT)[1]).all()
        d.reset()
        with pytest.raises(lib.JgaError):
            d.decode(abi.JPEG_DECODE_DCT)
        d.read_header()
        d.decode(abi.JPEG_DECODE_DCT)
        assert (d.coef() == orc.decode(data, oracle.DCT)[1]).all()
        d.reset()
        d.read_header()
        d.decode(abi.JPEG_DECODE_PACK)
        assert d.img.packed == sum(d.img.plane[i].packed for i in range(3)) > 0
        with pytest.raises(lib.JgaError):
            d.decode(7)


def test_irregular_huffman_tables(lib, orc, synth):
    """A valid file whose AC tables put all 162 symbols on 10-bit codes (81 long-code
    prefixes; the GPU entropy stage's lookup format holds 16): the host stage decodes it
    like any other."""
    import oracle
    data = synth.synthetic_jpeg(200, 120, "420", quality=85, restart_interval=5,
                                flags=synth.FLAT_AC)
    _, g = lib.geom_of(data)
    assert np.array_equal(lib.entropy_decode(data, g), orc.decode(data, oracle.QUANT)[1])
    if oracle.Reference.available():
        assert np.array_equal(oracle.Reference().decode(data, oracle.QUANT)[1],
                              lib.entropy_decode(data, g))


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
def test_two_symbols_per_lookup_edges(lib, orc, synth, sampling):
    """The host stage takes two short AC symbols per table look-up (csrc/entropy.c: build_pairs).  Levels made
    for that path's edges: blocks that run to coefficient 63 with no EOB (the bits behind them are the next
    block's DC code — an "EOB" found there is not one), runs of +-1 (three-bit symbols: every step a pair),
    an EOB as first and as second symbol of a step, a ZRL inside a pair, magnitudes too long for the table
    beside short ones; QUANT, DCT and PACK against the oracle."""
    import oracle
    w, h = 96, 80
    lv = pair_edge_levels(synth, w, h, sampling)
    data = synth.encode_levels(lv.reshape(-1), w, h, sampling, restart_interval=5)
    _, g = lib.geom_of(data)
    got = lib.entropy_decode(data, g, False).reshape(-1, 64)
    assert (got == lv).all() or _only_unused_slots_differ(lib, g, got, lv)
    assert (lib.entropy_decode(data, g, False) == orc.decode(data, oracle.QUANT)[1]).all()
    assert (lib.entropy_decode(data, g, True) == orc.decode(data, oracle.DCT)[1]).all()
    pack, idx, per = lib.entropy_decode_pack(data, g)
    assert sum(per) == len(pack)
    if oracle.Reference.available():                                # the compiled reference's own words and index
        words, index = oracle.Reference().decode(data, oracle.PACK)[1]
        assert np.array_equal(pack, words) and np.array_equal(idx, index)
    # every block's words expanded (res/horz_pack_yuv.fs.glsl:94-127, oracle.c) = its QUANT block
    quant = lib.entropy_decode(data, g, False)
    got = np.zeros_like(quant)
    for ipos, off in lib.block_slots(g):
        got[off[:, None] + np.arange(64)] = orc.unpack_blocks(pack, idx[ipos])
    assert np.array_equal(got, quant)


def pair_edge_levels(synth, w, h, sampling, seed=7):
    """Levels for the edges of "several symbols per table look-up" (host stage: build_pairs; GPU stage: packs)."""
    n = synth.coef_shorts(w, h, sampling)
    rng = np.random.default_rng(seed)
    lv = np.zeros(n, np.int16).reshape(-1, 64)
    for i, b in enumerate(lv):
        kind = i % 8
        b[0] = rng.integers(-60, 61)
        if kind == 0:
            b[1:] = rng.choice([-1, 1], 63)                         # full block of +-1: ends at 63, no EOB
        elif kind == 1:
            b[1:] = rng.choice([-1, 1], 63); b[40:] = 0             # pairs, then an EOB
        elif kind == 2:
            pass                                                    # DC only: the EOB is a step's first symbol
        elif kind == 3:
            b[1] = 1                                                # value, EOB: the EOB is the second
        elif kind == 4:
            b[1:] = rng.choice([-2, -1, 1, 2, 3], 63); b[63] = 1    # ends at 63 on a short symbol
        elif kind == 5:
            b[1] = 1; b[18] = -1; b[35] = 1; b[63] = -1             # ZRLs between short symbols
        elif kind == 6:
            b[1:] = rng.integers(-1023, 1024, 63)                   # long magnitudes: the one-symbol path
        else:
            b[1:32] = rng.choice([-1, 0, 0, 1, 300], 31)            # a mixture
    return lv
