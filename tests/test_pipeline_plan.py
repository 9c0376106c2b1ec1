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


def test_the_plan_of_a_decodes_rounds_is_one_table(lib):
    """csrc/huff_api.cpp: choose_rounds — the rows of its table (jga_huff_policy: host logic, no device).  A batch
    alone on the device and small keeps the dense kernel up to 128 k subsequences (and whenever it brought its own
    12-bit tables, or has restart intervals) and writes by block up to 64 k; everything that fills or shares the device
    runs its later rounds from work lists; in-group steps: 3, 4 with own 12-bit tables, 6 for long restart intervals
    at 64-byte subsequences, 2 for batches over 8 MB of frames with <= 3 blocks per MCU (24 MB with 4)."""
    L = lib.L
    L.jga_huff_policy.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.jga_huff_policy.restype = None

    def plan_of(total_sub, total_seg=1, sub_log2=7, nslots=6, ri=0, shared=0, own12=0):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.jga_huff_policy(total_sub, total_seg, sub_log2, nslots, ri, shared, own12, C.byref(a), C.byref(b), C.byref(c))
        return {"lists": bool(a.value), "iters": b.value, "by_block": bool(c.value)}
    k = 1024
    assert plan_of(6 * k) == {"lists": False, "iters": 3, "by_block": True}               # one 1080p 4:2:0 frame
    assert plan_of(24 * k, own12=1) == {"lists": False, "iters": 4, "by_block": True}      # one 4K frame, own 12-bit tables
    assert plan_of(96 * k) == {"lists": False, "iters": 3, "by_block": False}              # 8 x 2.7K: dense (round 6)
    assert plan_of(96 * k, shared=1) == {"lists": True, "iters": 3, "by_block": False}     # ... a pipeline's group: lists
    assert plan_of(192 * k) == {"lists": True, "iters": 3, "by_block": False}              # 8 x 4K alone
    assert plan_of(192 * k, own12=1)["lists"] is False and plan_of(192 * k, total_seg=2160, ri=240)["lists"] is False
    assert plan_of(1146 * k) == {"lists": True, "iters": 3, "by_block": False}             # 48 x 4K: the headline's batch
    # long restart intervals cut into 64-byte subsequences: six in-group steps (BASELINE config 5's 8K frame)
    assert plan_of(192 * k, total_seg=270, sub_log2=6, ri=480)["iters"] == 6
    assert plan_of(192 * k, total_seg=270 * 40, sub_log2=6, ri=12)["iters"] == 3           # ... short ones: three
    # few blocks per MCU, beyond a small batch: two steps
    assert plan_of(400 * k, nslots=3)["iters"] == 2 and plan_of(60 * k, nslots=3)["iters"] == 3      # 4:4:4: 50 MB / 7.5 MB
    assert plan_of(400 * k, nslots=1)["iters"] == 2
    assert plan_of(400 * k, nslots=4)["iters"] == 2 and plan_of(128 * k, nslots=4)["iters"] == 3     # 4:2:2: 50 MB / 16 MB
    assert plan_of(400 * k, nslots=6)["iters"] == 3
