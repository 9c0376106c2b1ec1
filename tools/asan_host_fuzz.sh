#!/bin/bash
# Host stages (marker parse, geometry, entropy decode to planes and to PACK, GPU-stage prepare)
# under AddressSanitizer + UBSan on randomly damaged files (headers and scans).  No GPU needed.
# Usage: tools/asan_host_fuzz.sh [nfiles] [seed]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/jpeg_gpu_amd/csrc; W=$(mktemp -d)
N=${1:-3000}; SEED=${2:-1}
F="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc -std=gnu11 $F -c $C/entropy.c -o $W/entropy.o
gcc -std=gnu11 $F -c $C/layout.c -o $W/layout.o
gcc -std=gnu11 $F -c $C/band.c -o $W/band.o
g++ -std=c++17 -mavx2 $F -I$C -c $C/huff_prepare.cpp -o $W/huff_prepare.o
g++ -std=c++17 $F -I$C -c $ROOT/tools/asan_driver.cpp -o $W/drv.o
g++ -fsanitize=address,undefined $W/drv.o $W/entropy.o $W/layout.o $W/band.o $W/huff_prepare.o -o $W/drv
mkdir -p $W/corpus
python3 - "$ROOT" "$W/corpus" "$N" "$SEED" <<'PY'
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from jpeg_gpu_amd import synth
out, n, seed = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(seed)
for it in range(n):
    samp = ["420", "444", "grey", "422", "440", "411"][it % 6]
    ri = [0, 1, 3, -1, 7][it % 5]
    flags = [0, 0, synth.DQT16, synth.SPLIT_DHT, synth.NO_JFIF, synth.FLAT_AC][it % 6]
    d = bytearray(synth.synthetic_jpeg(1 + (it * 37) % 160, 1 + (it * 23) % 120, samp, quality=5 + (it * 13) % 95,
                                       restart_interval=ri, seed=it, flags=flags))
    sos = d.find(b"\xff\xda")
    for _ in range(int(rng.integers(1, 8))):
        mode = int(rng.integers(0, 6))
        hi = sos + 14 if it % 3 == 0 else len(d) - 1            # a third of the files: headers only
        pos = int(rng.integers(2, max(3, min(hi, len(d) - 1))))
        if mode == 0: d[pos] = int(rng.integers(0, 256))
        elif mode == 1: d[pos] ^= 1 << int(rng.integers(0, 8))
        elif mode == 2: del d[pos]
        elif mode == 3: d.insert(pos, int(rng.integers(0, 256)))
        elif mode == 4: d[pos] = 0xFF
        else: d = d[:pos]
        if len(d) < 8: break
    open("%s/%05d.jpg" % (out, it), "wb").write(bytes(d))
PY
ASAN_OPTIONS=detect_leaks=1 JGA_QUIET=1 $W/drv $W/corpus/*.jpg
rm -rf $W
