"""bench.py's output contract (one JSON line: metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config / roofline /
cpu_baseline), on a short run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def run_bench(*flags, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT,
                       env={k: v for k, v in os.environ.items()
                            if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # ONE line on stdout, and it is the JSON
    assert len(lines[0]) < 4096, len(lines[0])                          # ... small enough for the driver's reader
    return json.loads(lines[0])


def details(line):
    """The full report the compact line points at (bench.py writes it beside itself and under gpurun_out/)."""
    with open(os.path.join(ROOT, line["details"])) as f:
        return json.load(f)


def _full_report(n_ranks):
    """A full report of realistic size: round 5's own (profiles/r5_bench.json, the 23.7 KB line the driver could not
    read), widened to n_ranks."""
    import copy
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench.json")))
    if n_ranks > 1:
        full["n_gpus"] = n_ranks
        full["per_rank"]["ranks"] = [dict(copy.deepcopy(full["per_rank"]["ranks"][0]), rank=i, gpu=i)
                                     for i in range(n_ranks)]
        for k in ("cpu_baseline", "configs", "scale_proxy", "other_kernels", "pack_stage", "gpu_entropy"):
            full.pop(k, None)                                          # (N = 1 only)
    return full


@pytest.mark.parametrize("n_ranks", [1, 2, 8])
def test_the_line_stays_under_4_kb_and_parses_alone(n_ranks):
    """the one line had grown to 23.7 KB and the driver recorded `parsed: null`.  compact_line() of a
    full-size report — 1 rank with every leg, and the 8-rank shape — is under 4 KB, parses on its own and still
    carries the contract keys, `roofline`, `cpu_baseline` (N = 1) and a summary of every config."""
    import bench
    full = _full_report(n_ranks)
    line = bench.compact_line(full, "bench_details.json")
    assert len(line.encode()) < bench.LINE_LIMIT == 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d and d[k] == full[k] or k in ("config", "roofline"), k
    assert "dropped_for_size" not in d and d["details"] == "bench_details.json"
    assert len(d["config"]["workload"]) <= 128 and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert d["roofline"][k] == full["roofline"][k]
    assert len(d["per_rank"]["Mpixel_s"]) == n_ranks
    if n_ranks == 1:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in d["cpu_baseline"]
        assert len(d["cpu_baseline"]["sample"]) <= 128
        assert set(d["configs"]) == set(full["configs"]) - {"note"}
        c4 = d["configs"]["config4_batch_1080p_420"]["rank3_shard_of_8"]
        assert c4["ms"] == [3.3, 3.43] and c4["best"] == [2.97, 3.09]   # [pageable, pinned]: medians, best
        assert d["configs"]["config5_8k_420_dri"]["of_link_ceiling"] == 0.979
    else:
        assert "cpu_baseline" not in d


def test_an_oversized_report_sheds_objects_instead_of_exceeding_the_limit():
    import bench
    full = _full_report(1)
    full["configs"].update({"extra_%d" % i: full["configs"]["config5_8k_420_dri"] for i in range(40)})
    line = bench.compact_line(full, "x")
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT and d["dropped_for_size"]
    assert d["value"] == full["value"] and "roofline" in d and "cpu_baseline" in d


def test_gpus_n_run_plainly_starts_n_ranks():
    """`python bench.py --gpus 2` with no torchrun around it must itself become 2 ranks (VERDICT r1:
    it used to fall back to one).  --dry-launch stops after the launch path: rendezvous on
    127.0.0.1, one process per rank, each with its own share of the host CPUs."""
    d = run_bench("--gpus", "2", "--dry-launch", timeout=300)
    assert d["dry_launch"] is True and d["n_gpus"] == 2
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert ranks[0]["pid"] != ranks[1]["pid"]
    a, b = set(ranks[0]["cpus"]), set(ranks[1]["cpus"])
    assert a and b
    if len(os.sched_getaffinity(0)) >= 2:
        assert not (a & b)                      # the ranks do not share host cores
        assert a | b <= set(os.sched_getaffinity(0))
    one = run_bench("--dry-launch", timeout=120)
    assert one["n_gpus"] == 1 and len(one["ranks"]) == 1


def test_eight_ranks_dry_launch():
    """The shape of the driver's 8-GPU run, without the GPUs: eight ranks rendezvous on 127.0.0.1,
    split the host CPUs among themselves (disjoint shares when there are at least eight) and each
    reports the share of the CPU grant it will size its threads by."""
    d = run_bench("--gpus", "8", "--dry-launch", timeout=600)
    ranks = d["ranks"]
    assert d["n_gpus"] == 8 and [r["rank"] for r in ranks] == list(range(8))
    assert len({r["pid"] for r in ranks}) == 8
    allowed = set(os.sched_getaffinity(0))
    for r in ranks:
        assert r["cpus"] and set(r["cpus"]) <= allowed and r["cpu_budget"] >= 1
    if len(allowed) >= 8:
        seen = set()
        for r in ranks:
            assert not (seen & set(r["cpus"]))
            seen |= set(r["cpus"])


def _failing_job(mode, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["JGA_BENCH_FAIL_RANK"] = mode
    env["JGA_BENCH_COMM_TIMEOUT_S"] = "60"
    t0 = __import__("time").perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.strip().startswith("{")]
    return r, lines, __import__("time").perf_counter() - t0


def test_a_rank_that_raises_mid_run_ends_the_job_and_is_named():
    """world 2 over gloo, rank 1 raises in the middle of its steps.  It tells the others in the
    status exchange that precedes every barrier; rank 0 does not wait in the barrier for the backend's timeout: the
    job ends non-zero within seconds and rank 0's one line says which rank failed and why."""
    r, lines, dt = _failing_job("1")
    assert r.returncode != 0 and dt < 120, (r.returncode, dt, r.stderr[-2000:])
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["failed_ranks"] == [1] and "injected failure" in d["error"] and "rank 1" in d["error"]
    assert d["value"] is None and d["n_gpus"] == 2 and "dry_launch" not in d
    # ... and rank 0 failing reports itself
    r, lines, dt = _failing_job("0")
    assert r.returncode != 0 and dt < 120 and len(lines) == 1 and lines[0]["failed_ranks"] == [0]


def test_a_rank_that_dies_without_a_word_ends_the_job_too():
    """... and when rank 1 is simply gone (os._exit: what an abort after a GPU fault looks like from outside) the job
    still ends non-zero well inside the timeout — rank 0's exchange fails as the peer's sockets close, or the
    launcher tears the job down first — and no result line is printed."""
    r, lines, dt = _failing_job("1,die")
    assert r.returncode != 0 and dt < 150, (r.returncode, dt, r.stderr[-2000:])
    assert not [d for d in lines if "dry_launch" in d]
    for d in lines:                                                # (rank 0 got its line out before the launcher's SIGTERM)
        assert d["value"] is None and "error" in d


def test_inputs_are_synthesised_once_per_box(tmp_path, synth):
    """make_inputs(): rank r makes the files whose index is r modulo the world size into a cache
    keyed by recipe + seed, every rank reads them all in its own rotation; a second call makes
    nothing.  (One process plays both ranks here: the barrier is a no-op.)"""
    import bench

    class NoComm:
        def barrier(self):
            pass
    os.environ["JGA_BENCH_CACHE"] = str(tmp_path)
    saved = bench.W, bench.H
    bench.W, bench.H = 64, 48
    try:
        made = []
        real = synth.synthetic_jpeg

        class Spy:
            SAMPLING = synth.SAMPLING

            @staticmethod
            def synthetic_jpeg(*a, **k):
                made.append(k.get("seed"))
                return real(*a, **k)
        with pytest.raises(FileNotFoundError):                         # rank 1 of 2 makes seeds 1235, 1237 —
            bench.make_inputs(Spy, 4, 1, 2, 2, NoComm())               # and finds rank 0's share missing
        assert sorted(made) == [1235, 1237] and len(os.listdir(tmp_path)) == 2
        f0, note = bench.make_inputs(Spy, 4, 0, 2, 2, NoComm())        # rank 0 makes the other two
        assert sorted(made) == [1234, 1235, 1236, 1237] and "once per box" in note
        assert sorted(f0) == sorted(real(64, 48, "420", 90, seed=s) for s in (1234, 1235, 1236, 1237))
        f1, _ = bench.make_inputs(Spy, 4, 1, 2, 2, NoComm())           # everything cached now
        assert len(made) == 4 and sorted(f1) == sorted(f0) and f1 != f0    # same files, another rotation
    finally:
        bench.W, bench.H = saved
        del os.environ["JGA_BENCH_CACHE"]


def test_rank_cpu_shares():
    """Ranks on one NUMA node split that node's cores, SMT siblings together; without topology
    information they split everything evenly."""
    from jpeg_gpu_amd.shard import parse_cpulist, rank_cpu_share
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)),
                 1: list(range(64, 128)) + list(range(192, 256))}
    cores = lambda pool: [[c, c + 128] for c in pool if c < 128]
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    shares = [rank_cpu_share(r, gpu_nodes, range(256), node_cpus, cores) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert sorted(sum(shares, [])) == list(range(256))                 # a partition of the box
    for r, s in enumerate(shares):
        assert set(s) <= set(node_cpus[gpu_nodes[r]])                  # on the GPU's own socket
        assert all((c + 128) in s for c in s if c < 128)               # both halves of each core
    assert rank_cpu_share(0, [0], range(256), node_cpus, cores) == sorted(node_cpus[0])
    # no topology (numa_node = -1), or a cpuset that excludes the GPU's node: even split of what we may use
    assert rank_cpu_share(1, [-1, -1], range(8), {}) == [4, 5, 6, 7]
    assert rank_cpu_share(1, [0, 0], [200, 201, 202, 203], {0: list(range(64))}) == [202, 203]
    assert rank_cpu_share(3, [0, 0, 0, 0], [5], {0: [5]}) == [5]       # never empty


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys(gpu):
    d = run_bench("--steps", "2", "--warmup", "1", "--batch", "4", "--group", "2", "--distinct", "4", "--lanes", "2",
                  "--prewarm", "0", "--kernel-reps", "3", "--kernel-batch", "4", "--cpu-rounds", "1", "--cpu-frames", "1",
                  "--no-e2e", "--no-pack", "--no-other", "--no-gpu-entropy", "--quick-configs",
                  "--no-measure-traffic", "--scale-proxy", "8", timeout=1500)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs", "per_rank", "details"):
        assert k in d, k                                       # the line itself
    line, d = d, details(d)
    assert line["value"] == d["value"] and line["roofline"]["frac"] == d["roofline"]["frac"]
    assert line["cpu_baseline"]["value"] == d["cpu_baseline"]["value"]
    assert set(line["configs"]) == set(d["configs"]) - {"note"}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline", "value_pageable", "value_pinned_ingest", "per_rank", "configs", "scale_proxy",
              "host_cost"):
        assert k in d, k
    # the scale proxy: the same steps by a child confined to one rank-of-8's CPUs, every image verified;
    # with that few cores the clean-up runs on the device and pageable files go through the input cache:
    # registered at first sight INSIDE the timed region (4 distinct files), DMA'd in place afterwards
    sp = d["scale_proxy"]
    assert sp["as_rank_of"] == 8 and sp["cpus"] == sp["cpu_budget"] == len(sp["cpu_list"]) >= 1
    assert sp["images"] == 8 and sp["images_verified"] == 16
    for v in ("pageable", "pinned"):
        assert sp[v]["Mpixel_s"] > 0 and sp[v]["cpu_ms_per_image"] >= 0 and sp[v]["scan_cleanup"] in ("host", "device")
    if sp["pageable"]["scan_cleanup"] == "device":
        # (a lane that meets a file another lane is registering at that moment copies it: rare, allowed)
        assert sp["pageable"]["registered_in_timed_region"] == 4 and sp["pageable"]["jobs_dma_in_place"] >= 6
        assert sp["pageable"]["host_bytes_per_image"] < 1_000_000 and sp["pinned"]["host_bytes_per_image"] == 0
    assert d["host_cost"]["pageable"]["cpu_ms_per_image"] >= 0
    # `value` is the conservative variant (ordinary pageable files); the pinned-ingest rate sits beside it
    assert d["value"] == d["value_pageable"] and d["value_pinned_ingest"] > 0
    # EVERY image of the timed region kept its pixels and was compared with the oracle
    assert d["config"]["images_verified"] == d["config"]["images_timed_per_gpu"] == 8
    assert d["config"]["images_verified_pinned_ingest"] == 8
    pr = d["per_rank"]
    assert len(pr["ranks"]) == 1 and pr["ranks"][0]["rank"] == 0 and pr["ranks"][0]["Mpixel_s"] > 0
    assert pr["min_Mpixel_s"] == pr["max_Mpixel_s"] == pr["ranks"][0]["Mpixel_s"]
    for key in ("gpu", "numa_node", "cpus", "ms_per_step", "scan_cleanup", "images_verified"):
        assert key in pr["ranks"][0], key
    # every BASELINE.json config beside its CPU path
    cf = d["configs"]
    assert {"headline_4k_420", "config2_1080p_420_one_image", "config3_4k_444", "config4_batch_1080p_420",
            "config5_8k_420_dri"} <= set(cf)
    for name in ("config2_1080p_420_one_image", "config3_4k_444", "config5_8k_420_dri"):
        e = cf[name]
        assert e["bit_exact_vs_oracle"] is True, name
        assert e["to_rgb_hbm"]["latency_ms"] > 0 and e["to_rgb_hbm"]["Mpixel_s"] > 0
        assert e["to_host_pixels"]["ms_per_frame"] > 0
        assert e["device"]["kernel_ms"] > 0 and 0 < e["device"]["kernel_hbm_frac"] < 1
        assert e["device"]["one_frame"]["ms"] > 0
        assert e["cpu"]["cores"] >= 1 and "libjpeg_turbo_rgb" in e["cpu"]
    e4 = cf["config4_batch_1080p_420"]
    assert e4["bit_exact_vs_oracle"] is True
    assert any(k.startswith("all_") for k in e4["to_rgb_hbm"]) and "rank3_shard_of_8" in e4["to_rgb_hbm"]
    for v in e4["to_rgb_hbm"].values():
        assert v["pageable_files"]["Mpixel_s"] > 0 and v["pinned_files"]["Mpixel_s"] > 0
    # ... and the photograph-like content class: verified, with what bounds it
    for name in ("photo_like_4k_420", "photo_like_1080p_420"):
        e = cf[name]
        assert e["bit_exact_vs_oracle"] is True and 0.08 < e["bytes_per_pixel"] < 0.25, name
        assert e["bound_by"] in ("link", "entropy stage", "block decode") and set(e["per_image_ms"]) == {"link", "entropy_stage", "block_decode"}
        assert e["device"]["huffman_ms"] > 0 and e["device"]["idct_rgb_ms"] > 0 and e["to_rgb_hbm"]["steady"]["Mpixel_s"] > 0
        assert line["configs"][name]["bound_by"] == e["bound_by"]
    hl = cf["headline_4k_420"]
    assert hl["to_rgb_hbm"]["Mpixel_s"] == d["value"] and hl["device"]["kernel_hbm_frac"] == d["roofline"]["frac"]
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "Mpixel/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["metric"].startswith("Mpixel/s end-to-end decode, 4K 4:2:0 baseline JPEG")
    cfg = d["config"]
    assert "host RAM" in cfg["workload"] and "-> RGB8 in HBM" in cfg["workload"] and "model" not in cfg
    assert cfg["scan_cleanup"] in ("host", "device")
    assert cfg["bit_exact_vs_oracle"] is True and cfg["images_timed_per_gpu"] == 8
    # value = pixels of the timed region / its wall time
    assert abs(d["value"] - 8 * 3840 * 2160 / (d["ms_per_step"] * 2 * 1e-3) / 1e6) < 0.01 * d["value"]
    # compressed bytes crossed PCIe, not 24.9 MB of planes
    assert 0 < cfg["h2d_bytes_per_image"] < 8_000_000
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / measured launch time
    want = rf["algorithmic_bytes_per_launch"] / rf["kernel_ms_per_launch"] / 1e6
    assert abs(rf["achieved"] - want) < 0.005 * want          # (the JSON rounds the milliseconds)
    assert rf["kernel_Mpixel_s"] > d["value"]                 # the kernel alone outruns the whole path
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["oracle_port_rgb"]["value"] > 0 and "libjpeg_turbo_rgb" in cb
    assert d["value"] > 0 and d["ms_per_step"] > 0


@pytest.mark.gpu
def test_one_wrong_byte_in_the_timed_regions_outputs_fails_the_bench(gpu):
    """The pixels of every image of the timed region are compared with the oracle's: flip one byte of
    one kept output (JGA_BENCH_CORRUPT = slot) and the run must exit non-zero, saying which job."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["JGA_BENCH_CORRUPT"] = "5"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--batch", "4",
                        "--group", "2", "--distinct", "4", "--lanes", "2", "--prewarm", "0", "--no-cpu", "--no-e2e",
                        "--no-pack", "--no-other", "--no-gpu-entropy", "--no-configs", "--no-measure-traffic", "--scale-proxy", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "differ from the oracle" in r.stderr and "job 5" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]      # no line is printed


@pytest.mark.gpu
def test_a_fault_in_the_configs_process_fails_the_bench(gpu):
    """the per-config legs run in a process of their own; when that process dies (a GPU memory fault
    aborts it), bench.py still prints its line — with the error in `configs` — and exits NON-ZERO."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["JGA_CONFIGS_FAULT"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--batch", "4",
                        "--group", "2", "--distinct", "4", "--lanes", "2", "--prewarm", "0", "--no-cpu", "--no-e2e",
                        "--no-pack", "--no-other", "--no-gpu-entropy", "--quick-configs", "--no-measure-traffic",
                        "--kernel-reps", "3", "--kernel-batch", "4", "--scale-proxy", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode != 0 and "configs leg failed" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and "error" in d["configs"]


@pytest.mark.gpu
def test_eight_ranks_sharing_the_one_gpu(gpu):
    """The driver's N = 8 launch on a one-GPU box (JGA_BENCH_SHARE_GPUS: the ranks share the device and
    talk over gloo; the rates mean nothing, the code path is the 8-rank one): inputs synthesised once
    per box, every rank verifies its own images, and the line carries each rank's own numbers."""
    os.environ["JGA_BENCH_SHARE_GPUS"] = "1"
    try:
        d = run_bench("--gpus", "8", "--steps", "1", "--warmup", "0", "--batch", "2", "--group", "2", "--distinct", "8",
                      "--lanes", "1", "--prewarm", "0", "--kernel-reps", "2", "--kernel-batch", "2", "--no-e2e",
                      timeout=1500)
    finally:
        del os.environ["JGA_BENCH_SHARE_GPUS"]
    assert len(d["per_rank"]["Mpixel_s"]) == 8 and d["per_rank"]["verified"] == 32
    d = details(d)
    assert d["n_gpus"] == 8 and d["config"]["images_verified"] == 2
    assert d["config"]["ranks_talk_over"] == "gloo" and "once per box" in d["config"]["inputs"]
    ranks = d["per_rank"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8)) and all(r["images_verified"] == 4 for r in ranks)
    assert d["per_rank"]["min_Mpixel_s"] <= d["per_rank"]["max_Mpixel_s"]
    assert abs(d["value"] - 8 * 2 * 3840 * 2160 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    assert "cpu_baseline" not in d and "configs" not in d


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself(gpu):
    """`python bench.py --gpus 2` with nobody's torchrun around it: two ranks, each with its own
    pipeline and its own share of the host cores, max-over-ranks timing, pixels of both summed.
    (On a one-GPU box the ranks share the device — JGA_BENCH_SHARE_GPUS — and use gloo: the
    rates mean nothing then, the code path is the N-rank one.)"""
    os.environ["JGA_BENCH_SHARE_GPUS"] = "1"
    try:
        d = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--distinct", "4",
                      "--lanes", "2", "--prewarm", "0", "--kernel-reps", "3", "--kernel-batch", "4")
    finally:
        del os.environ["JGA_BENCH_SHARE_GPUS"]
    d = details(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["images_timed_per_gpu"] == 8 and d["config"]["bit_exact_vs_oracle"] is True
    # whole-job pixels / max-over-ranks time
    assert abs(d["value"] - 2 * 8 * 3840 * 2160 / (d["ms_per_step"] * 2 * 1e-3) / 1e6) < 0.01 * d["value"]
    assert "cpu_baseline" not in d                                 # N = 1 only
    e = d["e2e"]
    assert e["north_star_host_huffman_to_rgb_hbm"]["ok"] and e["pageable_files_to_rgb_hbm"]["ok"]
    assert e["pageable_files_to_rgb_hbm"]["value"] == d["value"] and d["config"]["images_verified"] == 8
    assert [r["rank"] for r in d["per_rank"]["ranks"]] == [0, 1] and d["config"]["ranks_talk_over"] == "gloo"
    pin = d["config"]["cpu_pinning"]
    assert pin and 1 <= pin["cpus"] <= len(os.sched_getaffinity(0))


@pytest.mark.gpu
def test_graft_entry_smoke_runs(gpu):
    """__graft_entry__.smoke(): one small decode on cuda:0 checked against the oracle."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    __graft_entry__.smoke()


def test_graft_entry_build_is_idempotent():
    """build() with everything already built returns quickly and leaves the library loadable."""
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    __graft_entry__.build()
    assert os.path.exists(os.path.join(ROOT, "jpeg_gpu_amd", "libjpeg_gpu_amd.so"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))
