#!/bin/bash
# Differential fuzz of the host entropy stage across a rewrite: csrc/entropy.c of commit $1 against the tree's, on
# randomly damaged files — the same accept / reject decision, and for accepted files the same QUANT planes, DCT
# planes and PACK words + index.  NEWFLAGS=-DJGA_ENTROPY_NO_BMI2 runs the variants for CPUs without BMI2.  Usage: tools/archive/r4_entropy_diff_fuzz.sh <old commit> [nfiles] [seed]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); C=$ROOT/jpeg_gpu_amd/csrc; W=$(mktemp -d)
git -C $ROOT show $1:jpeg_gpu_amd/csrc/entropy.c > $W/entropy_old.c
gcc -std=gnu11 -O2 -fPIC -shared -I$C $W/entropy_old.c $C/layout.c -o $W/old.so 2>/dev/null
gcc -std=gnu11 -O2 -fPIC -shared $NEWFLAGS -I$C $C/entropy.c $C/layout.c -o $W/new.so 2>/dev/null
JGA_QUIET=1 PYTHONPATH=$ROOT python3 - $W ${2:-3000} ${3:-1} <<'PY'
import ctypes as C, sys, numpy as np
from jpeg_gpu_amd import lib, synth, abi
W, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
old, new = C.CDLL(W + "/old.so"), C.CDLL(W + "/new.so")
rng = np.random.default_rng(seed)
acc = rej = 0
for it in range(n):
    samp = ["420", "444", "grey", "422", "440", "411"][it % 6]
    ri = [0, 1, 3, -1, 7][it % 5]
    flags = [0, 0, synth.DQT16, synth.SPLIT_DHT, synth.NO_JFIF, synth.FLAT_AC][it % 6]
    d = bytearray(synth.synthetic_jpeg(1 + (it * 37) % 160, 1 + (it * 23) % 120, samp, quality=5 + (it * 13) % 95,
                                       restart_interval=ri, seed=it, flags=flags))
    sos = d.find(b"\xff\xda")
    if it % 4:                                                   # one file in four stays whole
        for _ in range(int(rng.integers(1, 6))):
            mode = int(rng.integers(0, 7))
            pos = int(rng.integers(min(sos + 14, len(d) - 2) if it % 3 and sos > 0 else 2, len(d) - 1))      # mostly the scan
            if mode == 0: d[pos] = int(rng.integers(0, 256))
            elif mode == 1: d[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2: del d[pos]
            elif mode == 3: d.insert(pos, int(rng.integers(0, 256)))
            elif mode == 4: d[pos] = 0xFF
            elif mode == 5: d[pos:pos + 2] = bytes([0xFF, 0xD0 + int(rng.integers(0, 10))])
            else: d = d[:pos]
            if len(d) < 8: break
            sos = d.find(b"\xff\xda")
    d = bytes(d)
    h = abi.jpeg_header()
    g = abi.jga_geom()
    if old.jga_parse_header(d, len(d), C.byref(h)) or old.jga_geom_from_header(C.byref(g), C.byref(h)):
        assert new.jga_parse_header(d, len(d), C.byref(h)) != 0 or new.jga_geom_from_header(C.byref(g), C.byref(h)) != 0, it
        rej += 1
        continue
    res = []
    for L in (old, new):
        r = []
        for dq in (0, 1):
            o = np.full(g.coef_shorts, 0x5a5a, np.int16)
            rc = L.jga_entropy_decode(d, len(d), C.byref(g), o.ctypes.data_as(C.c_void_p), dq)
            r.append((rc, o if rc == 0 else None))
        nidx = sum((g.plane[p].hblocks << g.plane[p].xdec) * g.plane[p].cstride for p in range(g.nplanes))
        cap = g.coef_shorts + 64 * nidx + 1024
        pk, ix = np.zeros(cap, np.int16), np.zeros(nidx, np.int32)
        nw, pw = C.c_longlong(0), (C.c_longlong * 3)()
        rc = L.jga_entropy_decode_pack(d, len(d), C.byref(g), pk.ctypes.data_as(C.c_void_p), C.c_longlong(cap),
                                       ix.ctypes.data_as(C.c_void_p), C.byref(nw), pw)
        r.append((rc, (pk[:nw.value].copy(), ix, list(pw)) if rc == 0 else None))
        res.append(r)
    for (rc0, a), (rc1, b) in zip(*res):
        assert (rc0 == 0) == (rc1 == 0), ("decision differs", it, rc0, rc1)
        if rc0 == 0:
            if isinstance(a, tuple):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], ("PACK differs", it)
            else:
                assert np.array_equal(a, b), ("planes differ", it)
    if res[0][0][0] == 0: acc += 1
    else: rej += 1
print("%d files: accepted %d, rejected %d; the two versions agree on every decision and every output" % (n, acc, rej))
PY
rm -rf $W
