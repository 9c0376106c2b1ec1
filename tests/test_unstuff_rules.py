"""The rules csrc/unstuff_kernels.hip decides everything by are LOCAL (a byte's fate depends on its
two neighbours and on prefix counts).  Here they are restated in numpy, byte for byte as the
kernel header states them, and held against the sequential host clean-up
(huff_prepare.cpp hj_prepare_scan, through tools/bin/libhuff_emul.so) on intact files and on
files with random damage inside the entropy-coded bytes: same clean stream, same restart
segments, same accept / reject.  Runs without a GPU; the kernels themselves are compared with
the oracle in tests/test_gpu_parity.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def emul(lib):
    subprocess.run([os.path.join(ROOT, "tools", "build_emul.sh")], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    E = C.CDLL(os.path.join(ROOT, "tools", "bin", "libhuff_emul.so"))
    E.huff_emul_clean_scan.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_longlong,
                                       C.POINTER(C.c_uint), C.c_void_p, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return E


def host_clean(E, data):
    clean = np.zeros(len(data) + 64, np.uint8)
    segs = np.zeros(2 * 70000, np.uint32)
    n, nseg, off = C.c_uint(), C.c_int(), C.c_int()
    rc = E.huff_emul_clean_scan(data, len(data), clean.ctypes.data, clean.size, C.byref(n),
                                segs.ctypes.data, segs.size // 2, C.byref(nseg), C.byref(off))
    if rc:
        return rc, None, None, off.value
    return 0, clean[:n.value].copy(), segs[:2 * nseg.value].reshape(-1, 2).copy(), off.value


def device_rules(raw, nseg, ri):
    """The kernel's rules on the whole scan at once.  -> (ok, clean, segments)"""
    avail = len(raw)
    b = np.frombuffer(raw, np.uint8).astype(np.int32)
    nxt = np.concatenate([b[1:], [0xD9]])                       # EOI past the end
    prv = np.concatenate([[0], b[:-1]])
    lead = b == 0xFF
    rst = lead & ((nxt & 0xF8) == 0xD0)
    drop = (lead & (nxt != 0)) | ((prv == 0xFF) & ~lead)
    term = lead & (nxt != 0) & (nxt != 0xFF) & ~rst
    end = int(np.argmax(term)) if term.any() else avail
    rpos = np.flatnonzero(rst)
    limit = nseg - 1
    if len(rpos) > limit and rpos[limit] < end:
        end = int(rpos[limit])                                  # the RSTn one too many
    keep = ~drop
    keep[end:] = False
    clean = b[keep].astype(np.uint8)
    kept_before = np.concatenate([[0], np.cumsum(keep)])        # clean position of every raw position
    accepted = [p for p in rpos[:limit] if p < end]
    ok = len(accepted) == limit
    for k, p in enumerate(accepted):
        ok = ok and nxt[p] == 0xD0 + (k & 7)                    # RSTn counters
    bnd = [int(kept_before[p]) for p in accepted]
    starts = [0] + bnd
    ends = bnd + [len(clean)]
    segs = np.array(list(zip(starts, ends)), np.uint32).reshape(-1, 2)
    return ok, clean, segs


def check(E, lib, data):
    hdr = lib.parse_header(data)
    g = lib.geom_from_header(hdr)
    mcus = g.nhmb * g.nvmb
    ri = hdr.restart_interval
    nseg = (mcus + ri - 1) // ri if ri else 1
    rc, hclean, hsegs, off = host_clean(E, data)
    ok, dclean, dsegs = device_rules(bytes(data[off:]), nseg, ri)
    assert ok == (rc == 0)
    if ok:
        assert np.array_equal(dclean, hclean)
        assert np.array_equal(dsegs, hsegs)
    return ok


@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
@pytest.mark.parametrize("ri", [0, -1, 1, 3])
def test_intact_files(emul, lib, synth, sampling, ri):
    for q, size in ((90, (333, 211)), (30, (97, 64))):
        assert check(emul, lib, synth.synthetic_jpeg(size[0], size[1], sampling, quality=q,
                                                     restart_interval=ri, seed=q))


def test_damaged_scans(emul, lib, synth):
    """Byte edits, bit flips, deletions, and planted FF xx pairs (markers, fill bytes, stuffed
    zeros at the very end) inside the scan."""
    rng = np.random.default_rng(7)
    accepted = rejected = 0
    for it in range(400):
        samp = ["420", "444", "grey", "422"][it % 4]
        d = bytearray(synth.synthetic_jpeg(120 + it % 37, 80 + it % 23, samp, quality=70,
                                           restart_interval=[0, 3, -1, 1][it % 4], seed=it))
        lo = d.find(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(lo, len(d) - 2))
            mode = int(rng.integers(0, 5))
            if mode == 0:
                d[pos] = int(rng.integers(0, 256))
            elif mode == 1:
                d[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2:
                del d[pos]
            elif mode == 3:
                d[pos:pos + 2] = bytes([0xFF, int(rng.choice([0x00, 0xFF, 0xD0, 0xD3, 0xD7, 0xD9, 0xC4, 0x01]))])
            else:
                d[pos:pos + 3] = b"\xff\xff\xff"
        if it % 9 == 0:
            d = d[:-2] + b"\xff"                                 # the file ends on a lone FF
        try:
            lib.geom_of(bytes(d))
        except lib.JgaError:
            continue
        if check(emul, lib, bytes(d)):
            accepted += 1
        else:
            rejected += 1
    assert accepted > 50 and rejected > 50
