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
    def __init__(self):
        self.z = np.load(os.path.join(GOLDEN, "jpegs.npz"))
        self.names = sorted(k[:-4] for k in self.z.files if k.endswith(".jpg"))

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
def golden_layout():
    with open(os.path.join(GOLDEN, "layout.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu(lib):
    if lib.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible "
                    "(the product has no CPU fallback)")
    return lib
