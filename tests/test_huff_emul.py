"""The GPU entropy stage's algorithm, validated without a GPU: tools/huff_emul.cpp
runs the same host+device core (csrc/huff_common.h) lane by lane with the kernels'
pass structure, in worst-case "every lane sees last round's states" order."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, three_table_variant


@pytest.fixture(scope="module")
def emul(lib):
    subprocess.run([os.path.join(ROOT, "tools", "build_emul.sh")], check=True)
    E = C.CDLL(os.path.join(ROOT, "tools", "bin", "libhuff_emul.so"))
    E.huff_emul_decode.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_longlong)]
    E.huff_emul_set_sub.argtypes = [C.c_int]
    E.huff_emul_set_assist.argtypes = [C.c_int]
    E.huff_emul_set_assist.restype = None
    E.huff_emul_walked.restype = C.c_longlong
    return E


def run(E, lib, data, jacobi=1, sub=128):
    assert E.huff_emul_set_sub(sub) == 0
    _, g = lib.geom_of(data)
    got = np.zeros(g.coef_shorts, np.int16)
    r, n, runs = C.c_int(), C.c_int(), C.c_longlong()
    rc = E.huff_emul_decode(data, len(data), got.ctypes.data, got.size, jacobi, C.byref(r),
                            C.byref(n), C.byref(runs))
    return rc, got, r.value, n.value, runs.value


@pytest.mark.parametrize("sampling", ["grey", "444", "422", "420", "440", "411"])
@pytest.mark.parametrize("ri", [0, -1, 1, 5])
@pytest.mark.parametrize("sub", [32, 64, 128])
def test_emulated_gpu_decode_equals_host_decode(emul, lib, synth, sampling, ri, sub):
    for q, size in ((90, (333, 211)), (35, (97, 64))):
        data = synth.synthetic_jpeg(size[0], size[1], sampling, quality=q, restart_interval=ri,
                                    seed=q)
        _, g = lib.geom_of(data)
        rc, got, rounds, nsub, runs = run(emul, lib, data, sub=sub)
        assert rc == 0
        assert np.array_equal(got, lib.entropy_decode(data, g))
        assert rounds <= nsub + 1


def test_golden_and_levels(emul, lib, synth, golden_jpegs):
    for name in golden_jpegs.names:
        for sub in (32, 64, 128):
            rc, got, *_ = run(emul, lib, golden_jpegs.jpeg(name), sub=sub)
            assert rc == 0 and np.array_equal(got, golden_jpegs[name + ".quant"]), (name, sub)
    # dense full-magnitude levels, long zero runs, stuffed FF bytes galore
    rng = np.random.default_rng(4)
    n = synth.coef_shorts(256, 128, "420")
    lv = rng.integers(-1023, 1024, n).astype(np.int16)
    lv.reshape(-1, 64)[::2, 5:] = 0
    data = synth.encode_levels(lv, 256, 128, "420")
    assert data.count(b"\xff\x00") > 50
    _, g = lib.geom_of(data)
    for sub in (32, 64, 128):
        rc, got, rounds, nsub, runs = run(emul, lib, data, sub=sub)
        assert rc == 0 and np.array_equal(got, lib.entropy_decode(data, g)), sub


def test_rounds_are_few(emul, lib, synth):
    """Self-synchronisation: lane-runs stay a small multiple of the lane count."""
    data = synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=1)
    rc, got, rounds, nsub, runs = run(emul, lib, data)
    assert rc == 0 and rounds < 40 and runs < 4 * nsub
    # shorter subsequences need more hand-overs to fall into step, not more bytes decoded
    rc, got, rounds32, nsub32, runs32 = run(emul, lib, data, sub=32)
    assert rc == 0 and nsub32 > 3 * nsub and rounds32 < 80 and runs32 < 8 * nsub32


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
def test_streams_that_never_fall_into_step_and_the_host_walk(emul, lib, synth, sampling):
    """Periodic data (flat blocks) can be parsed out of step for ever: the rounds alone then
    need about one round per subsequence.  With the host walk of the unsettled stretches
    (huff_prepare.cpp hj_walk_unsettled, what huff_api.cpp's assist_chains applies on the GPU
    path) the same streams settle two rounds after it, with the same coefficients."""
    w, h = 640, 360
    n = synth.coef_shorts(w, h, sampling)
    cases = []
    lv = np.zeros(n, np.int16)
    cases.append(lv.copy())
    lv.reshape(-1, 64)[:, 0] = 5
    lv.reshape(-1, 64)[::2, 0] = -5
    cases.append(lv.copy())
    lv = np.zeros(n, np.int16)
    lv.reshape(-1, 64)[:, 63] = 1
    cases.append(lv)
    slow = 0
    try:
        for lv in cases:
            data = synth.encode_levels(lv, w, h, sampling)
            _, g = lib.geom_of(data)
            want = lib.entropy_decode(data, g)
            emul.huff_emul_set_assist(0)
            rc, got, rounds_alone, nsub, _ = run(emul, lib, data)
            assert rc == 0 and np.array_equal(got, want)
            emul.huff_emul_set_assist(4)
            rc, got, rounds, nsub, _ = run(emul, lib, data)
            assert rc == 0 and np.array_equal(got, want)
            assert rounds <= 4 + 3
            if rounds_alone > 12:
                slow += 1
                assert emul.huff_emul_walked() > 0 and rounds < rounds_alone
        assert slow > 0                  # at least one of the cases really does not self-synchronise
        # an ordinary photograph never needs the walk (36 single-run rounds here = the GPU
        # path's 12 launches of up to 3 in-group iterations)
        emul.huff_emul_set_assist(36)
        data = synth.synthetic_jpeg(w, h, sampling, quality=90, seed=3)
        rc, got, rounds, nsub, _ = run(emul, lib, data)
        assert rc == 0 and emul.huff_emul_walked() == 0
    finally:
        emul.huff_emul_set_assist(0)


def test_frames_beyond_32_bit_offsets_are_handed_to_the_host_stage(emul, synth):
    """The kernels keep plane byte offsets and scan bit positions in 32 bits; prepare() must
    turn away (verdict 2 = "host entropy stage") a frame whose planes reach 4 GiB instead of
    letting the offsets wrap.  Only the header is looked at, so a small file with
    its SOF0 dimensions patched will do."""
    emul.huff_emul_prepare_head.argtypes = [C.c_char_p, C.c_int]
    data = bytearray(synth.synthetic_jpeg(64, 48, "444", seed=3))
    assert emul.huff_emul_prepare_head(bytes(data), len(data)) == 0
    sof = data.index(b"\xff\xc0")
    for dims, want in (((30000, 23000), 0),          # 30000 x 23000 4:4:4: 4.14e9 B < 2^32
                       ((30000, 24000), 2),          # 4.32e9 B >= 2^32
                       ((65500, 65500), 2)):
        data[sof + 5:sof + 9] = bytes([dims[1] >> 8, dims[1] & 255, dims[0] >> 8, dims[0] & 255])
        assert emul.huff_emul_prepare_head(bytes(data), len(data)) == want, dims


def test_ac_packs_change_nothing_on_any_bit_string(emul, synth, golden_jpegs):
    """hj_tables' AC entries carry PACKS (several whole symbols of the next 9 bits taken in one
    step; small batches: of the next 12, hj_wide_ac).  A run with them must compute what the symbols one by one
    would — for real scans and for arbitrary bytes (runs that start out of step decode garbage): end state and
    block count of 20 000 runs per table set, with the 9-bit packs, with the 12-bit ones and with them stripped."""
    emul.huff_emul_pack_mismatches.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                               C.POINTER(C.c_longlong)]
    rng = np.random.default_rng(12)
    files = [synth.synthetic_jpeg(64, 48, s, quality=q, seed=q) for s, q in (("420", 90), ("444", 35), ("grey", 75))]
    files.append(synth.synthetic_jpeg(64, 48, "420", quality=80, seed=3, flags=synth.SWAP_AC))   # luma on the chroma AC table
    files += [golden_jpegs.jpeg(n) for n in golden_jpegs.names[:4]]          # Pillow's optimised tables too
    for f in files:
        real = synth.synthetic_jpeg(640, 480, "420", quality=88, seed=5)
        scan = real[real.find(b"\xff\xda") + 14:]
        for data in (rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(), bytes(scan[:60000]),
                     bytes(60000), b"\xff" * 60000):
            packs = C.c_longlong()
            bad = emul.huff_emul_pack_mismatches(f, len(f), data, len(data), 5000, C.byref(packs))
            assert bad == 0
            assert packs.value > 200                               # a good part of the 9-bit patterns hold >= 2 symbols


def test_three_tables_of_a_class_go_to_the_host_stage(emul, lib, orc, synth):
    """Two DC + two AC tables is what the device format holds (what every encoder emits: luma /
    chroma).  A frame whose Cr has tables of its own is valid JPEG: prepare() gives it the
    verdict "host entropy stage", and that stage (and the oracle) decode it to the same planes."""
    import oracle
    emul.huff_emul_prepare_head.argtypes = [C.c_char_p, C.c_int]
    base = synth.synthetic_jpeg(97, 64, "420", quality=80, seed=4)
    odd = three_table_variant(base)
    assert emul.huff_emul_prepare_head(base, len(base)) == 0
    assert emul.huff_emul_prepare_head(odd, len(odd)) == 2
    # ... while a third id whose tables merely REPEAT the chroma ones (an encoder that writes one
    # DHT per component) shares their slots: tables are told apart by content, not by id
    same = three_table_variant(base, distinct=False)
    assert emul.huff_emul_prepare_head(same, len(same)) == 0
    assert np.array_equal(lib.entropy_decode(same, g if False else lib.geom_of(same)[1]), lib.entropy_decode(base, lib.geom_of(base)[1]))
    _, g = lib.geom_of(odd)
    want = lib.entropy_decode(base, g)
    assert np.array_equal(lib.entropy_decode(odd, g), want)
    assert np.array_equal(orc.decode(odd, oracle.QUANT)[1], want)


def test_subsequence_length_by_batch(emul):
    """hj_choose_sub_log2: 128 bytes, except for small batches of frames without subsampling
    (they fall into step within a few dozen bytes) and of frames cut into restart intervals (the
    next interval bounds the distance a run must walk): 64."""
    E = emul
    E.huff_emul_choose_sub.argtypes = [C.c_ulonglong, C.c_int, C.c_int]
    MB = 1 << 20
    assert E.huff_emul_choose_sub(3 * MB, 6, 0) == 128          # one 4K 4:2:0 frame
    assert E.huff_emul_choose_sub(6 * MB, 3, 0) == 64           # one 4K 4:4:4 frame
    assert E.huff_emul_choose_sub(9 * MB, 3, 0) == 128          # ... two of them
    assert E.huff_emul_choose_sub(4 * MB, 4, 0) == 64           # one 4K 4:2:2 frame (round 6's sweep)
    assert E.huff_emul_choose_sub(16 * MB, 4, 0) == 128
    assert E.huff_emul_choose_sub(12 * MB, 6, 480) == 64        # BASELINE config 5: 8K, an interval per MCU row
    assert E.huff_emul_choose_sub(3 * MB, 6, 8) == 64
    assert E.huff_emul_choose_sub(148 * MB, 6, 480) == 128      # a batch that fills the device
    assert E.huff_emul_choose_sub(1, 1, 0) == 64
