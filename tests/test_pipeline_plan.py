"""How the transport-2 pipeline cuts a run into groups (jga_pipeline_plan: host logic, no device):
same-geometry jobs in arrival order, groups sized by pixels, short jobs cut finer, a long job's first
groups rising in size from lane to lane, unparsable files on their own."""
import ctypes as C
from collections import Counter

import numpy as np
import pytest


def plan(lib, files, order, lanes=8, batch=48):
    jobs = lib.Pipeline.make_jobs([files[i] for i in order])
    n = len(order)
    group_of = (C.c_int * n)()
    lib.L.jga_pipeline_plan.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    ng = lib.L.jga_pipeline_plan(lanes, batch, jobs, n, group_of)
    g = np.array(group_of[:n])
    assert ng == g.max() + 1
    sizes = [int((g == k).sum()) for k in range(ng)]
    return g, sizes


def tiny(synth, w, h, sampling="420", seed=1):
    """A file with the header of a w x h frame (the planner reads SOF0 only): made small, then its
    frame size patched — cheap stand-ins for 4K and 1080p files."""
    data = bytearray(synth.synthetic_jpeg(16, 16, sampling, quality=50, seed=seed))
    i = data.index(b"\xff\xc0")
    data[i + 5:i + 9] = bytes([h >> 8, h & 255, w >> 8, w & 255])
    return bytes(data)


def test_long_job_groups_of_a_batch_with_a_rising_start(lib, synth):
    f = [tiny(synth, 3840, 2160)]
    g, sizes = plan(lib, f, [0] * 2560)
    assert sizes[:8] == [6, 12, 18, 24, 30, 36, 42, 48]           # lanes' first groups: 1/8 .. 8/8 of a group
    assert set(sizes[8:-1]) == {48} and 1 <= sizes[-1] <= 48
    assert sum(sizes) == 2560 and list(g) == sorted(g)            # arrival order kept


def test_short_jobs_are_cut_finer_and_stay_equal(lib, synth):
    f = [tiny(synth, 1920, 1080)]
    _, sizes = plan(lib, f, [0] * 128, batch=32)                  # rank 3's shard of config 4: 32 frame equivalents
    assert sizes == [16] * 8                                      # four frame equivalents each, one per lane
    _, sizes = plan(lib, f, [0] * 1024, batch=32)                 # all of config 4: 256 frame equivalents
    assert sizes[:8] == [4, 8, 12, 16, 20, 24, 28, 32] and set(sizes[8:-1]) == {32}
    _, sizes = plan(lib, f, [0] * 3, batch=32)
    assert sizes == [3]


def test_geometries_are_kept_apart_and_bad_files_alone(lib, synth):
    a, b = tiny(synth, 3840, 2160), tiny(synth, 1920, 1080, "444")
    files = [a, b, b"not a jpeg at all"]
    order = [0, 1, 0, 2, 1, 0, 1, 2] * 5
    g, sizes = plan(lib, files, order, lanes=4, batch=8)
    by_kind = {}
    for k, o in zip(g, order):
        by_kind.setdefault(int(k), set()).add(o)
    assert all(len(v) == 1 for v in by_kind.values())             # one geometry (or one bad file) per group
    assert Counter(len([1 for k, o in zip(g, order) if o == 2 and k == kk]) for kk in set(g[np.array(order) == 2])) == Counter({1: 10})


def test_plan_of_a_configuration_is_the_plan_of_its_lanes_and_batch(lib, synth):
    """jga_pipeline_plan_cfg makes the plan jga_pipeline_run of a pipeline created from the same
    configuration makes.  (The scheduling fields of round 4 — groups per lane, smallest group, the ramp — are
    no longer configuration: round 5 removed them, every other value had measured slower or the same.)"""
    f = [tiny(synth, 1920, 1080)]
    jobs = lib.Pipeline.make_jobs(f * 128)
    group_of = (C.c_int * 128)()

    def sizes(**cfg):
        c = lib.Pipeline.config(transport=2, **cfg)
        ng = lib.L.jga_pipeline_plan_cfg(C.byref(c), jobs, 128, group_of)
        g = np.array(group_of[:128])
        return [int((g == k).sum()) for k in range(ng)]
    assert sizes(depth=8, batch=32) == [16] * 8                   # as jga_pipeline_plan(8, 32, ...)
    assert sizes(depth=4, batch=32) == [16] * 8                   # none under four frame equivalents
    assert sizes(depth=2, batch=32) == [8] + [16] * 7 + [8]       # (long for two lanes: the first groups rise)
    assert sizes(depth=8, batch=2) == [8] * 16                    # ... and none over a full group
    with pytest.raises(TypeError):
        lib.Pipeline.config(transport=2, min_group=8)             # (gone)
