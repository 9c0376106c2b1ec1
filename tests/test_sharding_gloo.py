"""N>1 path on CPU: world_size-2 gloo processes shard a batch of JPEGs by image,
each runs the host entropy stage on its shard (the only stage that exists without
a GPU), and the job-level aggregation (sum of units / max of time, barrier-
bracketed) is what bench.py uses with RCCL on the GPU box."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from jpeg_gpu_amd import lib, synth, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
jpegs = [synth.synthetic_jpeg(160, 96, "420", seed=s) for s in range(11)]
mine = shard.shard_range(len(jpegs), rank, world)
dist.barrier()
t0 = time.perf_counter()
sums = []
for i in mine:
    _, g = lib.geom_of(jpegs[i])
    sums.append(int(lib.entropy_decode(jpegs[i], g).astype(np.int64).sum()))
dist.barrier()
dt = time.perf_counter() - t0
rate, units, tmax = shard.aggregate_throughput(len(mine) * 160 * 96, dt, dist)
gathered = [None] * world
dist.all_gather_object(gathered, (list(mine), sums))
if rank == 0:
    print("RESULT " + json.dumps({"rate": rate, "units": units, "tmax": tmax, "parts": gathered}))
dist.destroy_process_group()
'''


def test_shard_range_partitions():
    from jpeg_gpu_amd.shard import shard_range
    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            parts = [list(shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert list(shard_range(1024, 3, 8)) == list(range(384, 512))   # config 4: 128 per GPU
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_two_rank_gloo_job(tmp_path, lib, synth):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-3000:]
    import json
    res = json.loads(line[0][7:])
    assert res["units"] == 11 * 160 * 96
    idx = sorted(sum((p[0] for p in res["parts"]), []))
    assert idx == list(range(11))                      # every image decoded exactly once
    # same checksums as a single-process decode
    want = []
    for s in range(11):
        j = synth.synthetic_jpeg(160, 96, "420", seed=s)
        _, g = lib.geom_of(j)
        want.append(int(lib.entropy_decode(j, g).astype(np.int64).sum()))
    got = {}
    for ids, sums in res["parts"]:
        got.update(dict(zip(ids, sums)))
    assert [got[i] for i in range(11)] == want
    assert res["rate"] > 0 and res["tmax"] > 0
