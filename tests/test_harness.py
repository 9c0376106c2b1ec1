"""The headless harness jpeg_gpu_hip (csrc/harness.c) against the reference program's
command-line behaviour (src/jpeg_gpu.c:473-506, 508-606, 610-704): option handling,
`--header` text, `--dump` text per stage, and the steady-state loop finishing every `-o`
stage on the device."""
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "jpeg_gpu_amd", "jpeg_gpu_hip")
# the same program on libjpeg_gpu_amd_tuning.so: the only build in which the JGA_* A/B variables of
# rounds 1-3 exist (csrc/jga_tune.h) — tests of alternate code paths run this one
EXE_TUNING = os.path.join(ROOT, "jpeg_gpu_amd", "jpeg_gpu_hip_tuning")
SUBSAMP = ["Unknown", "4:4:4", "4:2:2", "4:2:0", "4:4:0", "4:1:1", "Mono"]


def run(*args, ok=True, env=None, exe=None):
    r = subprocess.run([exe or EXE] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300, env=dict(os.environ, JGA_QUIET="0", **(env or {})))
    if ok:
        assert r.returncode == 0, r.stderr
    return r


@pytest.fixture()
def jpg(tmp_path, synth):
    def make(*a, **k):
        data = synth.synthetic_jpeg(*a, **k)
        p = tmp_path / ("t%d.jpg" % len(list(tmp_path.iterdir())))
        p.write_bytes(data)
        return str(p), data
    return make


def numbers(text):
    """`Plane i` sections of a --dump -> list of int arrays (row-major)."""
    planes = []
    for sec in text.split("Plane ")[1:]:
        rows = [l for l in sec.split("\n")[1:] if l.strip()]
        planes.append(np.array([[int(v) for v in l.split()] for l in rows]))
    return planes


def test_usage_and_bad_options(jpg):
    path, _ = jpg(16, 16, "444")
    r = run(ok=False)
    assert r.returncode == 1 and r.stderr.startswith("Usage: jpeg_gpu_hip [options] jpeg_file")
    r = run("-h", ok=False)
    assert r.returncode == 1 and "--no-cpu" in r.stderr and "--header" in r.stderr
    r = run("-i", "nope", path, ok=False)
    assert r.returncode == 1 and r.stderr.startswith("Invalid decoder implementation: nope\n")
    r = run("-o", "jpeg", path, ok=False)
    assert r.returncode == 1 and r.stderr.startswith("Invalid decoder output format: jpeg\n")
    r = run("/nonexistent.jpg", ok=False)
    assert r.returncode == 1 and "Error, could not open jpeg file /nonexistent.jpg" in r.stderr


@pytest.mark.parametrize("sampling", ["grey", "444", "422", "420", "440", "411"])
def test_header_text(jpg, lib, sampling):
    """Byte-for-byte the text of src/jpeg_gpu.c:614-637."""
    path, data = jpg(123, 77, sampling, quality=60, restart_interval=5)
    h = lib.parse_header(data)
    want = ["Image Size         : %ix%i" % (h.width, h.height),
            "Bits Per Pixel     : %i" % h.bits,
            "Components         : %i" % h.ncomps,
            "Chroma Subsampling : %s" % SUBSAMP[h.subsamp],
            "Minimum Coded Unit : " + " ".join("%ix%i" % (h.comp[i].hsamp, h.comp[i].vsamp)
                                               for i in range(h.ncomps)),
            "Restart Interval   : %i" % h.restart_interval]
    for i in range(4):
        if h.quant[i].valid:
            want.append("Quant Table %i Bits : %i" % (i, h.quant[i].bits))
            t = list(h.quant[i].tbl)
            want += ["".join("%4i" % v for v in t[r * 8:r * 8 + 8]) for r in range(8)]
    for flag in ("-H", "--header"):
        assert run(flag, path).stdout == "\n".join(want) + "\n"


def test_dump_host_stages(jpg, lib, golden_jpegs, tmp_path):
    """pack / quant / dct dumps (host stages: no device needed)."""
    for name in golden_jpegs.names:
        p = tmp_path / (name + ".jpg")
        p.write_bytes(golden_jpegs.jpeg(name))
        _, g = lib.geom_of(golden_jpegs.jpeg(name))
        _, _, per = lib.entropy_decode_pack(golden_jpegs.jpeg(name), g)
        want = "".join("Plane %i Packed Data: %i\n" % (i, per[i]) for i in range(g.nplanes))
        want += "Packed Data : %i\n" % len(golden_jpegs[name + ".pack"])
        assert run("--dump", "-o", "pack", str(p)).stdout == want, name
        for stage in ("quant", "dct"):
            planes = numbers(run("-d", "-o", stage, str(p)).stdout)
            assert len(planes) == g.nplanes
            coef = golden_jpegs["%s.%s" % (name, stage)]      # from the compiled reference
            for i, got in enumerate(planes):
                pl = g.plane[i]
                w, hgt = pl.hblocks * 8, pl.vblocks * 8
                assert got.shape == (hgt, w)
                # the reference prints the plane's share of the buffer linearly
                assert np.array_equal(got.ravel(), coef[pl.coef_off:pl.coef_off + w * hgt]), name


def test_no_gpu_loop_prints_statistics(jpg):
    path, _ = jpg(64, 48, "420")
    out = run("--no-gpu", "-o", "quant", "--frames", "25", path).stdout
    assert out.startswith("25 FPS (cpu ") and "gpu 0.000 ms" in out


@pytest.mark.gpu
@pytest.mark.parametrize("sampling", ["grey", "444", "420", "411"])
def test_dump_device_stages(gpu, orc, jpg, sampling):
    import oracle
    path, data = jpg(100, 75, sampling, quality=85, restart_interval=3)
    info, planes = orc.decode(data, oracle.YUV)
    got = numbers(run("--dump", path).stdout)                 # -o yuv is the default
    assert len(got) == info.ncomps
    for a, b in zip(got, planes):
        assert np.array_equal(a, b)
    _, rgb = orc.decode_rgb(data)
    got = numbers(run("-d", "-o", "rgb", path).stdout)
    nc = info.ncomps
    want = rgb.reshape(75, 100, nc)
    for i in range(nc):
        assert np.array_equal(got[i], want[:, :, i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["pack", "quant", "dct", "yuv", "rgb"])
@pytest.mark.parametrize("sampling", ["420", "444", "grey"])
def test_main_loop_finishes_every_stage_on_the_device(gpu, orc, jpg, stage, sampling):
    """Whatever stage the host stops at, the RGB left in HBM is the oracle's."""
    path, data = jpg(333, 211, sampling, quality=90)
    _, rgb = orc.decode_rgb(data)
    out = run("-o", stage, "--frames", "3", "--check", path).stdout.strip().split("\n")
    assert out[0].startswith("3 FPS (cpu ")
    nc = 1 if sampling == "grey" else 3
    assert out[-1] == "RGB 333x211x%d adler32 %08x" % (nc, zlib.adler32(rgb.tobytes())), out
    # --no-cpu: the first decode is kept and the device still finishes it
    out = run("-o", stage, "--no-cpu", "--frames", "2", "--check", path).stdout.strip().split("\n")
    assert out[-1].endswith("%08x" % zlib.adler32(rgb.tobytes()))


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", ["gpu", "host"])
def test_plugin_entropy_stage_choice(gpu, orc, jpg, entropy):
    """decode_image(YUV|RGB) decodes the scan on the GPU by default and on the host with
    jga_plugin_config.host_entropy (the harness sets it from JPEG_GPU_HIP_ENTROPY=host): identical
    planes and pixels either way."""
    import oracle
    path, data = jpg(517, 389, "420", quality=92, restart_interval=7)
    env = {"JPEG_GPU_HIP_ENTROPY": entropy}
    _, planes = orc.decode(data, oracle.YUV)
    for a, b in zip(numbers(run("--dump", "-o", "yuv", path, env=env).stdout), planes):
        assert np.array_equal(a, b)
    _, rgb = orc.decode_rgb(data)
    out = run("-o", "rgb", "--frames", "4", "--check", path, env=env).stdout.strip().split("\n")
    assert out[-1].endswith("%08x" % zlib.adler32(rgb.tobytes()))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"JGA_STAGED": "0"}, {"JGA_HUFF_WRITE_GMEM": "0"},
                                 {"JGA_HUFF_LIST": "1"}, {"JGA_HUFF_ITERS": "1,1,3"},
                                 {"JGA_HUFF_FLUSH": "1"}, {"JPEG_GPU_HIP_REGISTER": "0"}, {"JPEG_GPU_HIP_REGISTER": "-1"},
                                 {"JGA_HUFF_SUB": "32"}, {"JGA_HUFF_SUB": "64"},
                                 {"JGA_HUFF_LITE": "0"}, {"JGA_HUFF_LITE": "1", "JGA_HUFF_ITERS": "1,1,2"},
                                 {"JGA_HUFF_LITE": "120"}, {"JGA_HUFF_PACKS": "0"},
                                 {"JGA_HUFF_LEAN": "0"}, {"JGA_HUFF_BY_BLOCK": "0"}])
def test_alternate_code_paths_give_the_same_pixels(gpu, orc, jpg, env):
    """Every tuning knob selects a different route to the same result: per-lane loads in the
    YUV kernel, the write pass with LDS-staged rows, list rounds for a lone frame, one in-group iteration
    per launch, unbatched block write-out, staged D2H, 32- and 64-byte subsequences, a counted
    first run, a lite first run that is a launch's only iteration, one that starts from the
    subsequence's first bit / from near its end, tables without multi-symbol packs, the dense
    rounds' stateless row reader, the write pass one lane per subsequence for a lone frame."""
    import oracle
    path, data = jpg(777, 431, "420", quality=88, restart_interval=0)
    _, rgb = orc.decode_rgb(data)
    want = "%08x" % zlib.adler32(rgb.tobytes())
    for stage in ("rgb", "yuv"):
        out = run("-o", stage, "--frames", "2", "--check", path, env=env, exe=EXE_TUNING).stdout.strip().split("\n")
        assert out[-1].endswith(want), (env, stage)
    _, planes = orc.decode(data, oracle.YUV)
    for a, b in zip(numbers(run("--dump", "-o", "yuv", path, env=env, exe=EXE_TUNING).stdout), planes):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("ri", [0, -1])
def test_registered_buffers_big_file(gpu, orc, jpg, ri):
    """jga_plugin_config.register_buffers (the harness's default; JPEG_GPU_HIP_REGISTER=0 turns it off) with a file of 1.5 MB or more: the file itself is registered, DMA'd
    where it lies and cleaned up on the device, the pixels come straight into the caller's
    registered buffers — same checksum as the oracle's pixels, frame after frame."""
    path, data = jpg(3840, 2160, "420", quality=90, restart_interval=ri)
    assert len(data) >= 3 << 19
    _, rgb = orc.decode_rgb(data)
    want = "%08x" % zlib.adler32(rgb.tobytes())
    # 1: buffers registered for the life of the decoder context; 0: for the length of each decode_image
    # call (the plugin's default: safe for any caller); -1: never, copies staged through pinned buffers
    for env in ({"JPEG_GPU_HIP_REGISTER": "1"}, {"JPEG_GPU_HIP_REGISTER": "0"}, {"JPEG_GPU_HIP_REGISTER": "-1"}):
        out = run("-o", "rgb", "--frames", "3", "--check", path, env=env).stdout.strip().split("\n")
        assert out[-1].endswith(want), (env, ri)
    out = run("-o", "yuv", "--frames", "2", "--check", path, env={"JPEG_GPU_HIP_REGISTER": "1"}).stdout.strip().split("\n")
    for mode in ("0", "-1"):
        assert out[-1].split()[-1] == run("-o", "yuv", "--frames", "2", "--check", path,
                                          env={"JPEG_GPU_HIP_REGISTER": mode}).stdout.strip().split("\n")[-1].split()[-1]


@pytest.mark.gpu
def test_plugin_when_the_callers_heap_hands_a_files_memory_out_again_for_pixels(gpu, orc, synth):
    """The reference's caller mallocs its file, decodes, frees (src/jpeg_info.c:31-62) and mallocs its frame
    (src/image.c:66-95).  Once glibc serves such sizes from the heap, a freed 6 MB file and the next frame's 6 MB of
    pixels are the same memory.  Round 5: the plugin's upload named the file's ordinary memory, the runtime pinned it
    READ-ONLY as a copy's source and kept that cached, and the copy back into the pixels that took its place died of
    "Memory access fault by GPU ... Write access to a read-only page" (bench.py's configs leg, one run in three).
    The file is now registered for the length of the call.  In a process of its own: a fault ends it."""
    import subprocess
    import sys
    import textwrap
    prog = textwrap.dedent("""
        import ctypes as C, gc, sys
        libc = C.CDLL("libc.so.6")
        libc.mallopt(-3, 64 << 20)            # M_MMAP_THRESHOLD: the heap serves everything under 64 MB
        sys.path.insert(0, %r)
        import numpy as np
        from jpeg_gpu_amd import abi, lib, synth
        import oracle
        orc = oracle.Oracle()
        big = synth.synthetic_jpeg(3840, 2160, "444", quality=90, seed=7)       # 6 MB: read by the device where it lies
        frame = synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=8)     # 6 MB of RGB
        want_big, want_frame = orc.decode_rgb(big)[1], orc.decode_rgb(frame)[1]
        for rep in range(6):
            with lib.Decoder(big) as d:
                d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
                assert np.array_equal(d.pixels(), want_big)
            gc.collect()
            with lib.Decoder(frame) as d:
                d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
                assert np.array_equal(d.pixels(), want_frame)
            gc.collect()
        print("fine")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fine" in r.stdout, (r.returncode, r.stderr[-600:])
