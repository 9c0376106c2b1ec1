"""The GPU entropy stage's LIST rounds (csrc/huff_kernels.hip: hj_list_build + hj_sync_list): from the second launch
on only the subsequences whose start state moved run, from per-image work lists.  Batches that fill the device take
them by default (tests/test_gpu_parity.py::test_gpu_huffman_full_size_and_end_to_end, tests/test_baseline_configs.py);
here small inputs are put on the same path (jga_huff_set_device_shared) so that every sampling, restart pattern,
stream that never falls into step and damaged scan goes through them, against the ORACLE's QUANT planes
(oracle.c, pinned to the compiled reference: src/xjpeg.c:449-632)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SAMPLINGS = ["grey", "444", "422", "420", "440", "411"]


def oracle_quant(orc, data):
    import oracle
    return orc.decode(data, oracle.QUANT)[1]


@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("ri", [0, -1, 3])
def test_list_rounds_equal_oracle_quant_stage(gpu, orc, synth, sampling, ri):
    """Frames large enough that their chains outlast the first launch (three in-group steps), several per batch:
    lists shared by several workgroups, then images iterated inside one workgroup."""
    datas = [synth.synthetic_jpeg(1280, 720, sampling, quality=q, restart_interval=ri, seed=q) for q in (90, 60, 35)]
    for unstuff in (False, True):
        g, coefs, rounds = gpu.gpu_entropy_decode(datas, device_unstuff=unstuff, shared=1)
        for d, c in zip(datas, coefs):
            assert np.array_equal(c, oracle_quant(orc, d)), (sampling, ri, unstuff)
        assert rounds >= 1


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
def test_list_rounds_tiny_and_odd_frames(gpu, orc, synth, sampling):
    """One MCU, one subsequence, sizes that are no multiple of anything: lists of zero or one entry."""
    for w, h, q in ((8, 8, 90), (17, 9, 50), (333, 211, 90), (641, 47, 20)):
        data = synth.synthetic_jpeg(w, h, sampling, quality=q, seed=w + h)
        _, coefs, _ = gpu.gpu_entropy_decode([data, data], shared=1)
        want = oracle_quant(orc, data)
        assert np.array_equal(coefs[0], want) and np.array_equal(coefs[1], want), (sampling, w, h)


@pytest.mark.parametrize("sampling", ["420", "444"])
def test_list_rounds_streams_that_never_fall_into_step(gpu, orc, synth, sampling):
    """Flat data parses out of step for ever: the host walks those stretches (assist_chains) and the lists are made
    afresh from the corrected states (the rebuild that first resets the counters)."""
    w, h = 1920, 1080
    n = synth.coef_shorts(w, h, sampling)
    cases = {}
    lv = np.zeros(n, np.int16)
    cases["zero"] = lv.copy()
    lv.reshape(-1, 64)[:, 0] = 5
    lv.reshape(-1, 64)[::2, 0] = -5
    cases["alternating dc"] = lv.copy()
    assisted = 0
    for name, lv in cases.items():
        data = synth.encode_levels(lv, w, h, sampling)
        _, coefs, _ = gpu.gpu_entropy_decode([data], shared=1)
        assert np.array_equal(coefs[0], oracle_quant(orc, data)), (sampling, name)
        assisted += gpu.gpu_entropy_decode.assisted
    assert assisted > 0
    data = synth.synthetic_jpeg(w, h, sampling, quality=90, seed=3)
    _, coefs, rounds = gpu.gpu_entropy_decode([data], shared=1)
    assert np.array_equal(coefs[0], oracle_quant(orc, data))
    assert gpu.gpu_entropy_decode.assisted == 0 and rounds <= 8


def test_list_rounds_on_corrupted_scans_agree_with_the_host_stage(gpu, synth):
    """Damaged entropy-coded data: the list rounds return (no hang), the decode rejects exactly what the host stage
    rejects and gives its coefficients otherwise."""
    rng = np.random.default_rng(11)
    accepted = rejected = 0
    for it in range(120):
        samp = ["420", "444", "grey", "422"][it % 4]
        d = bytearray(synth.synthetic_jpeg(400 + it % 37, 300 + it % 23, samp, quality=75,
                                           restart_interval=[0, 5, -1][it % 3], seed=it))
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
            got = gpu.gpu_entropy_decode([d], device_unstuff=bool(it & 1), shared=1)[1][0]
        except gpu.JgaError:
            got = None
        assert (got is None) == (want is None), it
        if got is not None:
            assert np.array_equal(got, want), it
            accepted += 1
        else:
            rejected += 1
    assert accepted > 10 and rejected > 10


def test_list_rounds_and_dense_rounds_give_the_same_planes_on_a_full_size_batch(gpu, orc, synth):
    """Six 4K 4:2:0 frames: alone on the device (dense kernel for every round) and as a batch that shares it (list
    rounds) — the same planes, the oracle's."""
    datas = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=40 + i) for i in range(3)] * 2
    _, alone, _ = gpu.gpu_entropy_decode(datas)
    _, shared, _ = gpu.gpu_entropy_decode(datas, shared=1)
    assert np.array_equal(alone, shared)
    for i in range(3):
        assert np.array_equal(shared[i], oracle_quant(orc, datas[i])), i


@pytest.mark.parametrize("sampling", ["420", "444"])
def test_list_rounds_with_shared_and_with_mixed_tables(gpu, orc, synth, sampling):
    """Batches of five and seven frames (beyond the per-image 12-bit tables of small batches): all with the Annex-K
    tables — ONE set of 12-bit AC tables goes up and the later list rounds take it — and mixed with files that code
    luma with the chroma AC table (no shared set: 9-bit tables throughout).  The oracle's QUANT planes every time,
    with the scan cleaned up on the host and on the device."""
    files = [synth.synthetic_jpeg(1280, 720, sampling, quality=q, restart_interval=ri, seed=q, flags=fl)
             for q, ri, fl in ((90, 0, 0), (60, 0, synth.SWAP_AC), (35, 7, 0), (90, 0, synth.SWAP_AC), (75, 0, 0),
                               (50, 0, 0), (82, 0, 0))]
    want = [oracle_quant(orc, f) for f in files]
    same = [0, 4, 5, 6, 0, 4, 5]                 # the Annex-K tables only, no restart markers
    mixed = [0, 1, 4, 3, 5, 6, 1]
    for order in (same[:5], same, mixed[:5], mixed):
        for unstuff in (False, True):
            _, coefs, _ = gpu.gpu_entropy_decode([files[i] for i in order], device_unstuff=unstuff, shared=1)
            for k, i in enumerate(order):
                assert np.array_equal(coefs[k], want[i]), (sampling, order, unstuff, k)
