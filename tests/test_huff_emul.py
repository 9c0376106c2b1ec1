"""The GPU entropy stage's algorithm, validated without a GPU: tools/huff_emul.cpp
runs the same host+device core (csrc/huff_common.h) lane by lane with the kernels'
pass structure, in worst-case "every lane sees last round's states" order."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def emul(lib):
    subprocess.run([os.path.join(ROOT, "tools", "build_emul.sh")], check=True)
    E = C.CDLL(os.path.join(ROOT, "tools", "bin", "libhuff_emul.so"))
    E.huff_emul_decode.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_longlong, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_longlong)]
    E.huff_emul_set_sub.argtypes = [C.c_int]
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
