"""Study (CPU emulation, tools/huff_emul.cpp huff_emul_hypotheses): a first guess of every subsequence's start state from
runs in every MCU-slot phase, linked by their states at a checkpoint `margin` bytes into the next subsequence — how
many states it gets wrong, how long the wrong stretches are, and how many Jacobi rounds the synchronisation then needs
against the plain guess.  Usage: python tools/archive/r5_hypotheses_study.py"""
import os, sys, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import synth
E = C.CDLL(os.path.join(ROOT, "tools", "bin", "libhuff_emul.so"))
out = (C.c_int * 9)()
E.huff_emul_set_sub(128)
for name, w, h, s in (("1080p 4:2:0", 1920, 1080, "420"), ("4K 4:2:0", 3840, 2160, "420"), ("4K 4:4:4", 3840, 2160, "444"), ("1080p 4:2:2", 1920, 1080, "422")):
    f = synth.synthetic_jpeg(w, h, s, quality=90, seed=1234)
    for m in (48, 64, 96):
        rc = E.huff_emul_hypotheses(f, len(f), m, out)
        print("%-12s margin %3d: %6d subsequences, %5d start states wrong (%4d although a hypothesis had the right one), chain broken %5d times, "
              "longest wrong stretch %2d | Jacobi rounds from this guess %2d (%.2f runs per subsequence), from the plain guess %2d"
              % (name, m, out[0], out[1], out[8], out[2], out[4], out[5], out[7] / 100, out[6]), flush=True)
