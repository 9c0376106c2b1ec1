"""How often do real-world Huffman tables fall outside the GPU entropy stage's lookup format
(hj_tables: 16 level-2 blocks)?  Pillow/libjpeg-turbo files with optimised and standard tables
over content kinds, qualities, samplings and sizes; only the headers are examined
(hj_prepare_head through tools/bin/libhuff_emul.so), so this runs without a GPU."""
import ctypes as C, io, os, sys
import numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 24
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = C.CDLL(os.path.join(ROOT, "tools", "bin", "libhuff_emul.so"))
E.huff_emul_prepare_head.argtypes = [C.c_char_p, C.c_int]
rng = np.random.default_rng(0)


def content(kind, w, h):
    x = np.linspace(0, 1, w)[None, :, None]; y = np.linspace(0, 1, h)[:, None, None]; c = np.arange(3)[None, None, :]
    if kind == 0: img = 128 + 80*np.sin(20*x*(c + 1))*np.cos(14*y) + rng.normal(0, 12, (h, w, 3))   # the bench recipe
    elif kind == 1: img = 128 + 60*np.sin(6*x*(c + 1))*np.cos(4*y) + rng.normal(0, 2, (h, w, 3))    # smooth
    elif kind == 2: img = rng.integers(0, 256, (h, w, 3)).astype(float)                              # white noise
    elif kind == 3: img = np.full((h, w, 3), 90.) + 40*(x > 0.5)                                    # flat + one edge
    else:                                                                                           # discs + grain
        img = np.zeros((h, w, 3))
        for _ in range(40):
            cx, cy, r = rng.random(), rng.random(), rng.random()*0.2
            img += (((x - cx)**2 + (y - cy)**2) < r*r)*rng.integers(-80, 80, 3)
        img = 128 + img + rng.normal(0, 4, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


by = {}
for kind in range(5):
    for q in (10, 30, 50, 75, 85, 90, 95, 100):
        for sub in (0, 1, 2):
            for size in ((640, 480), (333, 211)):
                for opt in (True, False):
                    buf = io.BytesIO()
                    Image.fromarray(content(kind, *size)).save(buf, "JPEG", quality=q, subsampling=sub, optimize=opt)
                    d = buf.getvalue()
                    rc = E.huff_emul_prepare_head(d, len(d))
                    by[(opt, rc)] = by.get((opt, rc), 0) + 1
                    if rc == 2:
                        print("outside the device format: content %d q %d subsampling %d %s optimize=%s" % (kind, q, sub, size, opt))
for opt in (True, False):
    n = sum(v for (o, _), v in by.items() if o == opt)
    print("optimize=%-5s: %d files, %d usable on the GPU, %d need the host entropy stage, %d rejected" % (
        opt, n, by.get((opt, 0), 0), by.get((opt, 2), 0), by.get((opt, 1), 0)))
