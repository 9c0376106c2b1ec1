"""Layout + C-ABI surface: image_init semantics (reference test/image.c:21-55 and
SURVEY.md Appendix B), struct sizes, exported symbols."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from conftest import ROOT


def make_header(abi, width, height, samp):
    h = abi.jpeg_header()
    h.bits, h.width, h.height, h.ncomps = 8, width, height, len(samp)
    hmax = max(s[0] for s in samp)
    vmax = max(s[1] for s in samp)
    nhmb = (width + 8 * hmax - 1) // (8 * hmax)
    nvmb = (height + 8 * vmax - 1) // (8 * vmax)
    for i, (hs, vs) in enumerate(samp):
        h.comp[i].hsamp, h.comp[i].vsamp = hs, vs
        h.comp[i].hblocks, h.comp[i].vblocks = nhmb * hs, nvmb * vs
    return h


def test_image_init_8bit_420(lib):
    """The reference's own layout test, check for check (test/image.c:21-55)."""
    from jpeg_gpu_amd import abi
    h = make_header(abi, 32, 24, [(2, 2), (1, 1), (1, 1)])
    img = abi.image()
    assert lib.L.jga_image_init(C.byref(img), C.byref(h)) == 0
    p = img.plane
    assert (p[0].xdec, p[0].ydec, p[0].xstride, p[0].ystride) == (0, 0, 1, 32)
    assert (p[1].xdec, p[1].ydec, p[1].xstride, p[1].ystride) == (1, 1, 1, 16)
    assert (p[2].xdec, p[2].ydec, p[2].xstride, p[2].ystride) == (1, 1, 1, 16)
    lib.L.jga_image_clear(C.byref(img))
    assert img.nplanes == 0 and not img.pixels


def test_layout_matches_reference_image_init(lib, golden_layout):
    """cstride / coef offsets / sizes for the BASELINE geometries (Appendix B),
    golden values captured from the compiled reference image_init."""
    from jpeg_gpu_amd import abi
    for name, want in golden_layout.items():
        samp = [tuple(s) for s in want["samp"]]
        h = make_header(abi, want["width"], want["height"], samp)
        g = lib.geom_from_header(h)
        n = want["ncomps"]
        assert g.coef_shorts == want["coef_shorts"], name
        assert [g.plane[i].coef_off for i in range(n)] == want["coef_off"], name
        assert [g.plane[i].cstride for i in range(n)] == want["cstride"], name
        assert [g.plane[i].xdec for i in range(n)] == want["xdec"], name
        assert [g.plane[i].ydec for i in range(n)] == want["ydec"], name
        assert (g.nhmb, g.nvmb) == (want["nhmb"], want["nvmb"]), name
        img = abi.image()
        assert lib.L.jga_image_init(C.byref(img), C.byref(h)) == 0
        for i in range(n):
            assert img.plane[i].cstride == want["cstride"][i]
            assert (img.plane[i].coef - img.coef) // 2 == want["coef_off"][i]
            assert img.plane[i].data % 16 == 0
        assert img.coef % 16 == 0 and img.pixels % 16 == 0 and img.index % 16 == 0
        lib.L.jga_image_clear(C.byref(img))


def test_appendix_b_sizes(lib):
    from jpeg_gpu_amd import abi
    cases = {(512, 512, ((1, 1),)): 524288, (1920, 1080, ((2, 2), (1, 1), (1, 1))): 6266880,
             (3840, 2160, ((1, 1),) * 3): 49766400,
             (3840, 2160, ((2, 2), (1, 1), (1, 1))): 24944640,
             (7680, 4320, ((2, 2), (1, 1), (1, 1))): 99532800}
    for (w, h, samp), nbytes in cases.items():
        g = lib.geom_from_header(make_header(abi, w, h, list(samp)))
        assert g.coef_shorts * 2 == nbytes
        assert g.rgb_bytes == w * h * len(samp)


def test_block_offset_formula(lib):
    from jpeg_gpu_amd import abi
    g = lib.geom_from_header(make_header(abi, 100, 75, [(2, 2), (1, 1), (1, 1)]))
    rs = g.w0 * 8
    assert lib.L.jga_block_offset(C.byref(g), 0, 3, 2) == rs * 2 + 3 * 64
    assert lib.L.jga_block_offset(C.byref(g), 1, 2, 3) == g.plane[1].coef_off + rs * 1 + rs // 2 + 128


def test_unsupported_headers_fail(lib):
    from jpeg_gpu_amd import abi
    h = make_header(abi, 16, 16, [(1, 1), (2, 2), (1, 1)])   # chroma wider than luma
    with pytest.raises(lib.JgaError):
        lib.geom_from_header(h)
    h = make_header(abi, 16, 16, [(1, 1)])
    h.ncomps = 2
    with pytest.raises(lib.JgaError):
        lib.geom_from_header(h)


def test_exports_every_declared_symbol(lib):
    """Every function/variable include/jpeg_gpu_amd.h declares is exported."""
    hdr = open(os.path.join(ROOT, "include", "jpeg_gpu_amd.h")).read()
    declared = set(re.findall(r"\b(jga_[a-z0-9_]+)\s*\(", hdr))
    declared |= set(re.findall(r"extern const jpeg_decode_ctx_vtbl (\w+);", hdr))
    assert {"HIPJPEG_DECODE_CTX_VTBL", "JGA_LIBJPEG_DECODE_CTX_VTBL"} <= declared
    declared -= {"jga_plane_geom", "jga_geom", "jga_pipeline_config", "jga_job"}
    declared -= set(re.findall(r"#define (jga_\w+)\(", hdr))           # macros expand in the caller
    assert declared == set(lib.EXPORTED)
    for name in sorted(declared):
        assert hasattr(lib.L, name), name
    v = lib.VTBL
    assert all(bool(f) for f in (v.decode_alloc, v.decode_header, v.decode_image,
                                 v.decode_reset, v.decode_free))
    assert lib.version().startswith("jpeg_gpu_amd")


def test_product_never_touches_oracle():
    """The product must not import/link/execute anything under oracle/."""
    pkg = os.path.join(ROOT, "jpeg_gpu_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".cpp", ".h", ".hip")):
                text = open(os.path.join(base, f)).read()
                code = "\n".join(l for l in text.splitlines()
                                 if not l.strip().startswith(("#", "//", "*", "/*")))
                assert "import oracle" not in code and "liboracle" not in code \
                    and "libjpeggpu_ref" not in code, f


def test_frames_whose_padded_planes_overflow_image_h_are_refused(lib, synth):
    """image_plane.width/height are unsigned shorts (src/image.h:31-32): a 65535-wide 4:2:0 frame
    pads to 65536 and cannot be described; the reference wraps silently, this build refuses."""
    import pytest
    with lib.Decoder(synth.synthetic_jpeg(65535, 8, "420", quality=50)) as d:
        d.read_header()
        with pytest.raises(lib.JgaError, match="too large"):
            d.init_image()
    with lib.Decoder(synth.synthetic_jpeg(65520, 8, "420", quality=50)) as d:
        d.read_header()
        d.init_image()
        assert d.img.plane[0].width == 65520 and d.img.plane[1].width == 32760


def test_pipeline_create_turns_away_a_caller_from_another_header_revision(lib):
    """jga_pipeline_config leads with the caller's sizeof(config) / sizeof(jga_job): a binary built
    against another revision of the header is refused before its job array is walked with the
    wrong stride (no device is touched before the check: runs without a GPU)."""
    import ctypes as C
    from jpeg_gpu_amd import abi
    for cs, js in ((0, 0), (C.sizeof(abi.jga_pipeline_config) - 4, C.sizeof(abi.jga_job)),
                   (C.sizeof(abi.jga_pipeline_config), C.sizeof(abi.jga_job) - 8)):
        cfg = abi.jga_pipeline_config(cs, js, 0, 1, 1, abi.JPEG_DECODE_RGB, 0, 0, 0, 2, 4, 0)
        assert not lib.L.jga_pipeline_create(C.byref(cfg))
        assert b"another revision" in lib.L.jga_last_error()


def test_cpu_quota_walks_the_process_cgroup_and_its_ancestors(tmp_path):
    """The CPU grant is the tightest limit between the process's own cgroup and the mount point,
    cgroup v2 (cpu.max) or v1 (cpu.cfs_*): what csrc/layout.c:jga_cpu_budget() does too."""
    from jpeg_gpu_amd import shard
    v2 = tmp_path / "v2"
    (v2 / "a" / "b").mkdir(parents=True)
    (v2 / "a" / "cpu.max").write_text("400000 100000\n")
    (v2 / "a" / "b" / "cpu.max").write_text("max 100000\n")
    (tmp_path / "p2").write_text("0::/a/b\n")
    assert shard.cpu_quota(str(v2), str(tmp_path / "p2")) == 4.0
    (v2 / "cpu.max").write_text("150000 100000\n")                    # the root is tighter still
    assert shard.cpu_quota(str(v2), str(tmp_path / "p2")) == 1.5
    v1 = tmp_path / "v1"
    (v1 / "cpu" / "x").mkdir(parents=True)
    (v1 / "cpu" / "x" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "x" / "cpu.cfs_period_us").write_text("100000\n")
    (tmp_path / "p1").write_text("3:cpu,cpuacct:/x\n2:memory:/y\n")
    assert shard.cpu_quota(str(v1), str(tmp_path / "p1")) == 2.5
    (v1 / "cpu" / "x" / "cpu.cfs_quota_us").write_text("-1\n")        # unlimited
    assert shard.cpu_quota(str(v1), str(tmp_path / "p1")) is None
    assert shard.rank_cpu_budget(64, 8, 16.0) == 2 and shard.rank_cpu_budget(64, 8, None) == 64
