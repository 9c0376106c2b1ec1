"""The drop-in boundary at the LINK level.

A maintainer who follows INTEGRATION.md §1 keeps linking the reference's jpeg_wrap.o, which DEFINES the
data symbols XJPEG_DECODE_CTX_VTBL and LIBJPEG_DECODE_CTX_VTBL (src/jpeg_wrap.c:246-252, 352-358), and
adds -ljpeg_gpu_amd.  So the library must not export either name (its comparison backend is
JGA_LIBJPEG_DECODE_CTX_VTBL), and a program that defines them itself must link, keep ITS tables, and run
the reference's call sequence (src/jpeg_gpu.c:612-613, 1215, 1231-1237) through HIPJPEG_DECODE_CTX_VTBL.
The QUANT stage is host work, so this runs without a GPU."""
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "jpeg_gpu_amd")
INC = os.path.join(ROOT, "include")

STUB_WRAP = r"""
/* stands where the reference's jpeg_wrap.o stands: defines the reference's two plugin tables */
#include "jpeg_gpu_amd.h"
#undef LIBJPEG_DECODE_CTX_VTBL
static jpeg_decode_ctx *stub_alloc(jpeg_info *info) { (void)info; return (jpeg_decode_ctx *)0; }
static int stub_header(jpeg_decode_ctx *d, jpeg_header *h) { (void)d; (void)h; return 77; }
static int stub_image(jpeg_decode_ctx *d, image *i, jpeg_decode_out o) { (void)d; (void)i; (void)o; return 78; }
static void stub_reset(jpeg_decode_ctx *d, jpeg_info *i) { (void)d; (void)i; }
static void stub_free(jpeg_decode_ctx *d) { (void)d; }
const jpeg_decode_ctx_vtbl LIBJPEG_DECODE_CTX_VTBL = {stub_alloc, stub_header, stub_image, stub_reset, stub_free};
const jpeg_decode_ctx_vtbl XJPEG_DECODE_CTX_VTBL = {stub_alloc, stub_header, stub_image, stub_reset, stub_free};
"""

MAIN = r"""
#include <stdio.h>
#include <stdlib.h>
#include "jpeg_gpu_amd.h"
#undef LIBJPEG_DECODE_CTX_VTBL                       /* as a file that included the reference's jpeg_wrap.h sees it */
extern const jpeg_decode_ctx_vtbl LIBJPEG_DECODE_CTX_VTBL, XJPEG_DECODE_CTX_VTBL;
int main(int argc, char **argv) {
  jpeg_decode_ctx_vtbl vtbl = HIPJPEG_DECODE_CTX_VTBL;
  jpeg_info info;
  jpeg_header header;
  image img;
  jpeg_decode_ctx *dec;
  FILE *f;
  long n, k;
  unsigned long sum = 0;
  int frame;
  if (argc != 2 || !(f = fopen(argv[1], "rb"))) return 2;
  fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
  info.size = (int)n; info.buf = (unsigned char *)malloc((size_t)n);
  if (fread(info.buf, 1, (size_t)n, f) != (size_t)n) return 2;
  fclose(f);
  /* the program's own tables are the ones it sees: the library interposes nothing */
  if (LIBJPEG_DECODE_CTX_VTBL.decode_header(NULL, NULL) != 77 || XJPEG_DECODE_CTX_VTBL.decode_image(NULL, NULL, JPEG_DECODE_YUV) != 78) return 3;
  if (&JGA_LIBJPEG_DECODE_CTX_VTBL == &LIBJPEG_DECODE_CTX_VTBL) return 4;
  if (JGA_LIBJPEG_DECODE_CTX_VTBL.decode_alloc == LIBJPEG_DECODE_CTX_VTBL.decode_alloc) return 5;
  /* src/jpeg_gpu.c:612-613, 1215: alloc -> header -> image_init; 1231-1237: reset -> header -> image per frame */
  dec = vtbl.decode_alloc(&info);
  if (!dec || vtbl.decode_header(dec, &header) != EXIT_SUCCESS) return 6;
  if (jga_image_init(&img, &header) != EXIT_SUCCESS) return 7;
  for (frame = 0; frame < 3; frame++) {
    vtbl.decode_reset(dec, &info);
    if (vtbl.decode_header(dec, &header) != EXIT_SUCCESS) return 8;
    if (vtbl.decode_image(dec, &img, JPEG_DECODE_QUANT) != EXIT_SUCCESS) return 9;
  }
  {
    jga_geom g;
    if (jga_geom_from_header(&g, &header) != EXIT_SUCCESS) return 10;
    for (k = 0; k < g.coef_shorts; k++) sum = sum*31u + (unsigned short)img.coef[k];
    printf("%dx%d %d %lu\n", header.width, header.height, header.ncomps, sum & 0xfffffffful);
  }
  vtbl.decode_free(dec);
  jga_image_clear(&img);
  free(info.buf);
  return 0;
}
"""


def test_library_exports_neither_of_the_references_table_names():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, "libjpeg_gpu_amd.so")],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    names = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert "HIPJPEG_DECODE_CTX_VTBL" in names and "JGA_LIBJPEG_DECODE_CTX_VTBL" in names
    assert "LIBJPEG_DECODE_CTX_VTBL" not in names and "XJPEG_DECODE_CTX_VTBL" not in names


def test_a_program_that_defines_the_references_tables_links_and_runs(tmp_path, lib, synth, orc):
    import oracle
    (tmp_path / "stub_wrap.c").write_text(STUB_WRAP)
    (tmp_path / "main.c").write_text(MAIN)
    exe = str(tmp_path / "dropin")
    r = subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + INC, "-o", exe,
                        str(tmp_path / "main.c"), str(tmp_path / "stub_wrap.c"), "-L" + PKG, "-ljpeg_gpu_amd",
                        "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    data = synth.synthetic_jpeg(203, 117, "420", quality=85, restart_interval=5, seed=9)
    (tmp_path / "t.jpg").write_bytes(data)
    r = subprocess.run([exe, str(tmp_path / "t.jpg")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr)
    want = orc.decode(data, oracle.QUANT)[1].astype(np.uint16)
    s = 0
    for v in want.tolist():
        s = (s * 31 + v) & 0xFFFFFFFFFFFFFFFF
    assert r.stdout.split() == ["203x117", "3", str(s & 0xFFFFFFFF)]


def test_header_coexists_with_the_references_own_headers(tmp_path):
    """With the reference's headers included FIRST (their include guards skip this header's
    restatement of the data model) the reference's name stays the reference's: no alias."""
    ref = "/root/reference/src"
    if not os.path.exists(os.path.join(ref, "jpeg_wrap.h")):
        import pytest
        pytest.skip("the reference's headers are not on this box")
    (tmp_path / "both.c").write_text(
        '#include "jpeg_wrap.h"\n#include "jpeg_gpu_amd.h"\n'
        "#ifdef LIBJPEG_DECODE_CTX_VTBL\n#error the alias must not exist beside jpeg_wrap.h\n#endif\n"
        "const void *a(void) { return &LIBJPEG_DECODE_CTX_VTBL; }\n"
        "const void *b(void) { return &JGA_LIBJPEG_DECODE_CTX_VTBL; }\n"
        "const void *c(void) { return &HIPJPEG_DECODE_CTX_VTBL; }\n")
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I" + ref, "-I" + INC, "-c",
                        str(tmp_path / "both.c"), "-o", str(tmp_path / "both.o")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    syms = subprocess.run(["nm", str(tmp_path / "both.o")], stdout=subprocess.PIPE, text=True).stdout
    assert re.search(r"\bU LIBJPEG_DECODE_CTX_VTBL\b", syms) and re.search(r"\bU JGA_LIBJPEG_DECODE_CTX_VTBL\b", syms)


def test_product_reads_no_tuning_variable_from_the_environment():
    """The A/B knobs live behind jga_tune() (NULL in the product, getenv only in layout.c of the tuning
    build); what the default build reads from the environment is three documented variables, all in C files."""
    csrc = os.path.join(PKG, "csrc")
    hits = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".h", ".c")) and f not in ("harness.c", "synth_encode.c"):
            text = open(os.path.join(csrc, f)).read()
            code = "\n".join(l for l in text.splitlines() if not l.strip().startswith(("//", "*", "/*")))
            n = len(re.findall(r"\bgetenv\s*\(", code))
            if n:
                hits[f] = n
    assert hits == {"layout.c": 3, "libjpeg_vtbl.c": 1}, hits       # JGA_QUIET, JGA_CPU_BUDGET, jga_tune; JGA_LIBJPEG


def test_tuning_variables_move_the_tuning_build_only(tmp_path):
    """JGA_PIPE_MIN_GROUP (one of rounds 2-3's A/B variables) changes the plan jga_pipeline_plan_cfg makes in
    libjpeg_gpu_amd_tuning.so and nothing at all in the product library (host logic: no GPU needed)."""
    import sys
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np
from jpeg_gpu_amd import lib, synth
d = bytearray(synth.synthetic_jpeg(16, 16, "420", quality=50, seed=1))
i = d.index(b"\xff\xc0"); d[i + 5:i + 9] = bytes([1080 >> 8, 1080 & 255, 1920 >> 8, 1920 & 255])
jobs = lib.Pipeline.make_jobs([bytes(d)] * 128)
g = (C.c_int * 128)()
c = lib.Pipeline.config(transport=2, depth=8, batch=32)
print(lib.L.jga_pipeline_plan_cfg(C.byref(c), jobs, 128, g))
""" % ROOT
    out = {}
    for name in ("libjpeg_gpu_amd.so", "libjpeg_gpu_amd_tuning.so"):
        env = dict(os.environ, JGA_LIB_PATH=os.path.join(PKG, name), JGA_PIPE_MIN_GROUP="8")
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        out[name] = int(r.stdout.strip().split()[-1])
    assert out == {"libjpeg_gpu_amd.so": 8, "libjpeg_gpu_amd_tuning.so": 4}
