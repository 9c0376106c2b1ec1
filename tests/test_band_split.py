"""One frame in bands of MCU rows (include/jpeg_gpu_amd.h: jga_band_plan / jga_band_file; SURVEY.md §8e's
note; restart intervals: reference src/xjpeg.c:593-629).  Host side here: the bands of a frame are files
the oracle AND the compiled reference decode, and their pixels / planes stacked are the frame's."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _stack_rgb(orc, lib, data, count):
    bands = lib.band_plan(data, count)
    rows = []
    for b in bands:
        f = lib.band_file(data, b)
        h = lib.parse_header(f)
        assert (h.height, h.width) == (b.rows, lib.parse_header(data).width)
        rgb = orc.decode_rgb(f)[1]
        assert rgb.shape[0] == b.rows
        rows.append(rgb)
    return bands, np.concatenate(rows, 0)


@pytest.mark.parametrize("sampling", ["grey", "444", "422", "420", "440", "411"])
@pytest.mark.parametrize("count", [1, 2, 3, 8])
def test_bands_stack_to_the_frame(lib, orc, synth, sampling, count):
    # 13 x 9 MCUs of 16 x 16 at most; one restart interval per MCU row
    data = synth.synthetic_jpeg(203, 141, sampling, quality=85, restart_interval=-1, seed=11)
    whole = orc.decode_rgb(data)[1]
    bands, stacked = _stack_rgb(orc, lib, data, count)
    assert (stacked == whole).all()
    # the bands tile the frame, sizes within one row of MCUs of each other
    assert [b.y0 for b in bands] == list(np.cumsum([0] + [b.rows for b in bands[:-1]]))
    assert sum(b.rows for b in bands) == 141
    assert max(b.mcu_rows for b in bands) - min(b.mcu_rows for b in bands) <= 1
    assert len(bands) == min(count, sum(b.mcu_rows for b in bands))


@pytest.mark.parametrize("ri,rows_per_group", [(13, 1), (26, 2), (1, 1), (39, 3), (5, 5), (6, 6)])
def test_intervals_that_do_not_equal_a_row(lib, orc, synth, ri, rows_per_group):
    # 203 px of 4:2:0 = 13 MCUs per row, 9 rows: a band starts where row*13 is a multiple of the interval
    data = synth.synthetic_jpeg(203, 141, "420", quality=85, restart_interval=ri, seed=3)
    bands = lib.band_plan(data, 64)
    assert len(bands) == -(-9 // rows_per_group)
    assert all(b.mcu_row0 % rows_per_group == 0 for b in bands)
    whole = orc.decode_rgb(data)[1]
    _, stacked = _stack_rgb(orc, lib, data, 64)
    assert (stacked == whole).all()
    # counters renumbered: a band that starts at interval 8k + j, j > 0, begins again at RST0
    for b in bands[1:]:
        f = lib.band_file(data, b)
        scan = f[f.index(b"\xff\xda"):]
        marks = [scan[i + 1] for i in range(len(scan) - 1) if scan[i] == 0xFF and 0xD0 <= scan[i + 1] <= 0xD7]
        assert marks == [0xD0 + (i & 7) for i in range(len(marks))]


def test_no_restart_markers_is_one_band(lib, orc, synth):
    data = synth.synthetic_jpeg(96, 64, "420", seed=2)
    bands = lib.band_plan(data, 8)
    assert len(bands) == 1 and (bands[0].y0, bands[0].rows) == (0, 64)
    assert (orc.decode_rgb(lib.band_file(data, bands[0]))[1] == orc.decode_rgb(data)[1]).all()
    from jpeg_gpu_amd import shard
    assert shard.band_of_rank(data, 3, 8) == (0, 0, None)
    with pytest.raises(ValueError):
        shard.band_of_rank(data, 8, 8)


def test_coefficient_planes_of_the_bands_are_the_frames(lib, orc, synth):
    # QUANT planes: band rows of blocks = the same rows of the frame's planes (layout: Appendix B)
    import oracle
    data = synth.synthetic_jpeg(208, 144, "420", quality=90, restart_interval=-1, seed=9)   # 13 x 9 MCUs, no padding
    h, g = lib.geom_of(data)
    whole = lib.entropy_decode(data, g, False)
    for b in lib.band_plan(data, 4):
        f = lib.band_file(data, b)
        hb, gb = lib.geom_of(f)
        part = lib.entropy_decode(f, gb, False)
        assert (part == orc.decode(f, oracle.QUANT)[1]).all()
        for p in range(3):
            vs = h.comp[p].vsamp
            for by in range(b.mcu_rows*vs):
                for bx in (0, h.comp[p].hblocks - 1):
                    o_w = lib.L.jga_block_offset(g, p, bx, b.mcu_row0*vs + by)
                    o_b = lib.L.jga_block_offset(gb, p, bx, by)
                    assert (whole[o_w:o_w + 64] == part[o_b:o_b + 64]).all()


def test_the_compiled_reference_decodes_the_bands(lib, synth):
    import oracle
    if not oracle.Reference.available():
        pytest.skip("oracle/_ref not built")
    ref = oracle.Reference()
    data = synth.synthetic_jpeg(203, 141, "420", quality=85, restart_interval=13, seed=21)
    _, whole = ref.decode(data, oracle.YUV)
    parts = [ref.decode(lib.band_file(data, b), oracle.YUV)[1] for b in lib.band_plan(data, 3)]
    for p in range(3):
        rows = np.concatenate([np.asarray(x[p]) for x in parts], 0)
        w = np.asarray(whole[p])
        assert (rows[:w.shape[0]] == w[:rows.shape[0]]).all() and rows.shape[0] >= w.shape[0] - 16


def test_bad_inputs(lib, synth):
    data = synth.synthetic_jpeg(203, 141, "420", restart_interval=-1, seed=1)
    with pytest.raises(lib.JgaError):
        lib.band_plan(b"\xff\xd8\xff\xd9", 2)
    with pytest.raises(lib.JgaError):
        lib.band_plan(data[:len(data)//2], 2)            # markers missing
    bands = lib.band_plan(data, 2)
    bad = type(bands[0])()
    bad.rows, bad.scan_off, bad.scan_bytes = 16, len(data) - 4, 64
    with pytest.raises(lib.JgaError):
        lib.band_file(data, bad)
    # a truncated band FILE fails in the decoder like any truncated file
    f = lib.band_file(data, bands[1])
    h, g = lib.geom_of(f)
    with pytest.raises(lib.JgaError):
        lib.entropy_decode(f[:len(f)//2], g, False)


WORKER = r'''
import json, os, sys, zlib
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from jpeg_gpu_amd import lib, synth, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
frame = synth.synthetic_jpeg(331, 250, "420", quality=90, restart_interval=-1, seed=77)
y0, rows, f = shard.band_of_rank(frame, rank, world)
_, g = lib.geom_of(f)
coef = lib.entropy_decode(f, g, False)             # the stage that exists without a GPU
got = [None]*world
dist.all_gather_object(got, (y0, rows, zlib.crc32(coef.tobytes())))
if rank == 0:
    print("RESULT " + json.dumps(got))
dist.destroy_process_group()
'''


def test_two_ranks_take_a_band_each(tmp_path, lib, synth):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-3000:]
    import json, zlib
    res = json.loads(line[0][7:])
    frame = synth.synthetic_jpeg(331, 250, "420", quality=90, restart_interval=-1, seed=77)
    bands = lib.band_plan(frame, 2)
    assert [(y0, rows) for y0, rows, _ in res] == [(b.y0, b.rows) for b in bands] == [(0, 128), (128, 122)]
    for (_, _, crc), b in zip(res, bands):
        f = lib.band_file(frame, b)
        _, g = lib.geom_of(f)
        assert crc == zlib.crc32(lib.entropy_decode(f, g, False).tobytes())


# ---- on the GPU: every band through the product's decode paths = its rows of the frame ----------

def _bands_on_gpu(gpu, abi, files, host_outs, transport=2):
    pl = gpu.Pipeline(device=0, nthreads=min(8, len(os.sched_getaffinity(0))), out=abi.JPEG_DECODE_RGB,
                      copy_back=True, transport=transport, batch=8, depth=2)
    try:
        rc, jobs = pl.run(files, host_outs=host_outs)
        assert rc == 0 and all(j.status == 0 for j in jobs)
    finally:
        pl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sampling,count", [("420", 8), ("444", 3), ("grey", 5), ("422", 2), ("411", 4)])
def test_bands_decoded_on_the_gpu_are_the_frames_rows(gpu, orc, synth, sampling, count):
    from jpeg_gpu_amd import abi
    data = synth.synthetic_jpeg(1283, 977, sampling, quality=88, restart_interval=-1, seed=31)
    whole = orc.decode_rgb(data)[1]
    bands = gpu.band_plan(data, count)
    assert len(bands) == count
    files = [gpu.band_file(data, b) for b in bands]
    n = 1 if sampling == "grey" else 3
    outs = [np.zeros(b.rows * 1283 * n, np.uint8) for b in bands]
    # bands differ in height by a row of MCUs: one job list per geometry is not required — the
    # pipeline groups files by geometry itself
    _bands_on_gpu(gpu, abi, files, outs)
    got = np.concatenate([o.reshape((b.rows, 1283, 3) if n == 3 else (b.rows, 1283)) for o, b in zip(outs, bands)], 0)
    assert np.array_equal(got, whole)
    # ... and one band through the plugin (decode_alloc -> header -> image -> pixels)
    b = bands[len(bands) // 2]
    with gpu.Decoder(files[len(bands) // 2]) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)
        assert np.array_equal(d.pixels(), whole[b.y0:b.y0 + b.rows])


@pytest.mark.gpu
def test_config5_8k_frame_in_eight_bands(gpu, orc, synth):
    """BASELINE.json configs[4] (7680x4320 4:2:0, an interval per MCU row) across 8 GPUs, band by band on
    the one GPU here: each rank's band (shard.band_of_rank) decoded to RGB = its rows of the oracle's frame."""
    from jpeg_gpu_amd import abi, shard
    data = synth.synthetic_jpeg(7680, 4320, "420", quality=90, restart_interval=-1, seed=1234)
    whole = orc.decode_rgb(data)[1]
    parts = [shard.band_of_rank(data, r, 8) for r in range(8)]
    assert [p[0] for p in parts] == [0, 544, 1088, 1632, 2176, 2720, 3264, 3792]      # 270 rows of MCUs: 34 x 6 + 33 x 2
    assert sum(p[1] for p in parts) == 4320
    assert sum(len(p[2]) for p in parts) < len(data) + 8 * 1024
    outs = [np.zeros(rows * 7680 * 3, np.uint8) for _, rows, _ in parts]
    _bands_on_gpu(gpu, abi, [p[2] for p in parts], outs)
    for (y0, rows, _), o in zip(parts, outs):
        assert np.array_equal(o.reshape(rows, 7680, 3), whole[y0:y0 + rows]), y0
