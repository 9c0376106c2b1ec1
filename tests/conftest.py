"""Shared fixtures.  `-m "not gpu"` = oracle pins, host logic, C-ABI surface (CPU
only); `-m gpu` = parity tests proper, through the C-ABI, on a real MI355X."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the native libraries exist (cross-compiles without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def orc(_built):
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def ref(_built):
    """The compiled reference (oracle/_ref); absent on a clone without /root/reference."""
    import oracle
    if not oracle.Reference.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return oracle.Reference()


@pytest.fixture(scope="session")
def lib(_built):
    from jpeg_gpu_amd import lib as L
    return L


@pytest.fixture(scope="session")
def synth(_built):
    from jpeg_gpu_amd import synth as S
    return S


@pytest.fixture(scope="session")
def golden_blocks():
    z = np.load(os.path.join(GOLDEN, "idct_blocks.npz"))
    return z["inp"], z["out"]


class GoldenJpegs:
    """Files + the compiled reference's outputs for them (tests/golden/make_golden.py)."""
    def __init__(self, files=("jpegs.npz", "jpegs_rare.npz")):
        self.z = {}
        for f in files:
            with np.load(os.path.join(GOLDEN, f)) as z:
                self.z.update({k: z[k] for k in z.files})
        self.names = sorted(k[:-4] for k in self.z if k.endswith(".jpg"))

    def jpeg(self, name):
        return self.z[name + ".jpg"].tobytes()

    def info(self, name):
        return json.loads(self.z[name + ".info"].tobytes().decode())

    def planes(self, name):
        return [self.z["%s.plane%d" % (name, i)] for i in range(self.info(name)["ncomps"])]

    def __getitem__(self, key):
        return self.z[key]


@pytest.fixture(scope="session")
def golden_jpegs():
    return GoldenJpegs()


@pytest.fixture(scope="session")
def golden_mcu18():
    """Luma 4x4 (18 blocks per MCU): beyond T.81 B.2.3 and libjpeg, decoded by the reference."""
    return GoldenJpegs(("jpegs_mcu18.npz",))


@pytest.fixture(scope="session")
def golden_layout():
    with open(os.path.join(GOLDEN, "layout.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu(lib):
    if lib.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible "
                    "(the product has no CPU fallback)")
    return lib


def three_table_variant(data, distinct=True):
    """The same image with its Cr component on a THIRD pair of Huffman tables: the chroma DHTs are
    repeated under table id 2 and the SOS points Cr there.  Decodes to the same coefficients.
    distinct: the DC copy also gets one more (never used) 16-bit code, so that the frame really
    has three different DC tables — more than the GPU entropy stage's table format (two DC + two
    AC tables per frame) holds; without it the copies are byte-identical to the chroma tables
    and the device stage shares their slots."""
    d = bytearray(data)
    i, dhts, sos = 2, [], None
    while i < len(d):
        assert d[i] == 0xFF
        m, n = d[i + 1], (d[i + 2] << 8) | d[i + 3]
        if m == 0xC4:
            dhts.append((i, n))
        if m == 0xDA:
            sos = i
            break
        i += 2 + n
    extra = bytearray()
    for off, n in dhts:
        p = off + 4
        while p < off + 2 + n:
            cnt = sum(d[p + 1:p + 17])
            if d[p] & 15 == 1:                          # a chroma table (id 1): repeat it as id 2
                seg = bytearray(d[p:p + 17 + cnt])
                seg[0] = (seg[0] & 0xF0) | 2
                if distinct and seg[0] >> 4 == 0:       # DC class: one more code at length 16
                    seg[16] += 1
                    seg.append(0x0F)
                extra += b"\xff\xc4" + bytes([(len(seg) + 2) >> 8, (len(seg) + 2) & 255]) + seg
            p += 17 + cnt
    assert extra and d[sos + 4] == 3                    # three components in the scan
    d[sos + 5 + 2 * 2 + 1] = 0x22                       # third component: Td = Ta = 2
    return bytes(d[:sos]) + bytes(extra) + bytes(d[sos:])
