#!/usr/bin/env python3
"""bench.py — headline benchmark of the JPEG block-decode path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment this program starts its own N ranks (one process per
GPU, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` of itself on
127.0.0.1); started under torchrun by somebody else it simply is one of the ranks.

The metric is BASELINE.json's: Mpixel/s END-TO-END — from JPEG file bytes in host RAM to RGB8
pixels in HBM (what a frame is in the reference: src/jpeg_gpu.c:1231-1237,
its cpu/gpu split 1437-1458) — on 3840x2160 4:2:0 q90 baseline files.  A "step" is ONE
batch of `--batch` (128) images per GPU through the pipelined decoder (jga_pipeline,
transport 2, which works on them in groups of `--group` = 32): host threads parse markers and
unstuff the scans into pinned memory, the compressed bytes cross PCIe, the GPU does the Huffman
decode and the fused dequantise + IDCT + upsample + RGB kernel.  The timed region is exactly K
such batches per rank, streamed through the rank's lanes, bracketed by barrier + device
synchronise; `value` = pixels of all ranks / max-over-ranks wall time.  EVERY image of the timed
region keeps its pixels (slices of one HBM buffer) and is compared with the oracle's pixels of
its file after the clock stops (`config.images_verified`); a single differing byte fails the run.
`value` is measured on files lying in ordinary PAGEABLE host memory; `value_pinned_ingest` (same
run, same check) on files lying in pinned ingest buffers.  Images are independent: each rank
owns its own images, its own share of the host cores (those of its GPU's NUMA node) and there
is no data-path collective (weak scaling); RCCL carries the barrier and the MAX only (gloo
if RCCL cannot be brought up: said in the line).

Rank 0 prints ONE JSON line.  Beside the contract keys:
  roofline      the fused IDCT+RGB kernel alone on coefficient planes resident in HBM:
                algorithmic bytes / launch time from HIP events on the launch stream
                (`kernel_Mpixel_s` is that kernel's pixel rate — NOT the end-to-end value);
                `traffic` from rocprofv3 --pmc child passes of this run when the tool is here
  per_rank      every rank's own rate, GPU, NUMA node, CPUs, scan clean-up route, H2D GB/s and CPU ms per
                image (+ min / max)
  scale_proxy   (N = 1) the headline repeated by a child confined to the CPUs ONE rank of 8 gets, and
                `concurrent`: eight such children alive at once on the one GPU, each on its own CPUs
  configs       (N = 1) every BASELINE.json config beside its CPU path (tools/configs_bench.py): one-frame
                latency, a short job's rate, a steady state of >= 0.5 s with its share of the link's
                ceiling; config 4's shard as median and best of 15 runs
  e2e           the same end-to-end measurement with the other transports: the north-star
                design (host Huffman threads + pinned hipMemcpyAsync + kernel), with the pixels
                copied back to host RAM, PACK words over PCIe
  cpu_baseline  (N = 1) the CPU paths timed on this box's host cores in the same run, warm,
                best of several rounds, one frame loop per core: the reference's own
                xjpeg + dct.c compiled from its sources (oracle/_ref, YUV stage — it has no
                CPU RGB stage), libjpeg-turbo through LIBJPEG_DECODE_CTX_VTBL (RGB), and the
                oracle port of the whole path (RGB)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
W, H, SAMPLING, QUALITY = 3840, 2160, "420", 90
METRIC = "Mpixel/s end-to-end decode, 4K 4:2:0 baseline JPEG, at 1/2/4/8 MI355X"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=448,
                    help="images per GPU per step (20 steps = 8 960 images, 223 GB of kept outputs, a timed "
                         "region of ~0.5 s: VERDICT r3 asked for >= 0.5 s; 128 until round 3)")
    ap.add_argument("--group", type=int, default=32,
                    help="images the pipeline hands to the GPU entropy stage at a time (one lane's batch)")
    ap.add_argument("--kernel-batch", type=int, default=48, help="images per launch in the roofline leg")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic images per rank")
    ap.add_argument("--lanes", type=int, default=8, help="batches in flight per GPU")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host threads per rank (0 = this rank's share of the cores)")
    ap.add_argument("--no-pin", action="store_true", help="leave the ranks' CPU affinity alone")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the other-transports leg")
    ap.add_argument("--no-pack", action="store_true", help="skip the PACK expansion leg")
    ap.add_argument("--no-other", action="store_true", help="skip the other-kernels leg")
    ap.add_argument("--no-gpu-entropy", action="store_true", help="skip the device-only breakdown leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-BASELINE-config leg")
    ap.add_argument("--quick-configs", action="store_true", help="per-config leg on small counts (tests)")
    ap.add_argument("--prewarm", type=float, default=0.5,
                    help="seconds of untimed work before the headline and before the roofline leg (clock ramp)")
    ap.add_argument("--kernel-reps", type=int, default=50, help="launches timed in the roofline leg")
    ap.add_argument("--cpu-rounds", type=int, default=5, help="cpu_baseline: best of this many rounds")
    ap.add_argument("--cpu-frames", type=int, default=6, help="cpu_baseline: frames per core per round")
    ap.add_argument("--measure-traffic", dest="measure_traffic", action="store_true", default=None,
                    help="collect roofline.traffic now with rocprofv3 --pmc child passes (the default "
                         "at N = 1 when rocprofv3 is on the box)")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false",
                    help="quote profiles/pmc_latest.json instead (with its provenance)")
    ap.add_argument("--max-keep-GB", type=float, default=236.0,
                    help="HBM the timed region's retained outputs may take (every image is kept and "
                         "verified when they fit: 223 GB at the defaults; beyond it slots are re-used by "
                         "jobs that decode the same file and the last writer of each slot is checked)")
    ap.add_argument("--scale-proxy", type=int, default=8,
                    help="N = 1: repeat the headline in a child confined to the CPUs ONE rank of this many "
                         "gets (cgroup grant / N, on the GPU's NUMA node) -> `scale_proxy` (0 = skip)")
    ap.add_argument("--as-rank-of", type=int, default=0, help=argparse.SUPPRESS)   # the scale-proxy child
    ap.add_argument("--proxy-slot", type=int, default=0, help=argparse.SUPPRESS)   # ... which share of the cores it takes
    ap.add_argument("--proxy-dir", default="", help=argparse.SUPPRESS)             # ... and where the N children meet
    ap.add_argument("--concurrent-proxy", dest="no_concurrent_proxy", action="store_false", default=True,
                    help="also run scale_proxy.concurrent (N children alive at once on the one GPU, each on its own CPUs; "
                         "off by default since round 6: eight processes time-slice one device, so it says little about "
                         "eight GPUs - VERDICT r5 - and costs the run 18 s)")
    ap.add_argument("--input-cache-MB", type=int, default=1024,
                    help="jga_pipeline_config.input_cache_mb of the headline pipelines: > 0 a persistent cache of that "
                         "many MB (pageable files are registered at first sight inside the timed region and DMA'd "
                         "where they lie from then on - used by the library where the scan clean-up runs on the "
                         "device); 0 = the library's default (registrations that live as long as their group); "
                         "-1 / -2 = none (named copies / host copies)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, report who they are and which CPUs they own, exit "
                         "(gloo; needs no GPU)")
    return ap.parse_args(argv)


# ---- launcher ---------------------------------------------------------------------------------

def launch_ranks(args, argv):
    """`bench.py --gpus N` run plainly: become the launcher of N ranks of this same file."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd[1:9])))
    return subprocess.call(cmd, env=env)


class quiet_stdout:
    """Send file descriptor 1 to stderr for the duration: libraries that chat on stdout while the
    process group comes up (gloo prints "[Gloo] Rank 0 is connected to ..." from C++) must not
    put lines next to the ONE JSON line rank 0 owes its caller."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def dry_launch(args, rank, local_rank, world):
    """Who am I, which CPUs are mine — the launch path without the GPU work (CPU test): the ranks come up over gloo,
    pin themselves, run `--steps` stand-in steps (the host entropy stage on a small file) between the same fences the
    real run uses, and rank 0 prints who they are.  JGA_BENCH_FAIL_RANK = "r" (an exception on rank r in the middle
    of its steps) or "r,die" (the process is gone without a word) exercise the failure path: every rank leaves
    non-zero within seconds and rank 0's line names the rank."""
    import torch
    import torch.distributed as dist
    import __graft_entry__
    __graft_entry__.build()
    from jpeg_gpu_amd import lib, shard, synth
    global _COMM
    comm = _COMM = Comm(torch, dist, rank, world, None, True)
    pin = {} if args.no_pin else shard.pin_rank_to_gpu_node(local_rank, world)
    quota = shard.cpu_quota()
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(),
          "cpus": sorted(os.sched_getaffinity(0)), "pin": pin,
          "cpu_budget": shard.rank_cpu_budget(len(os.sched_getaffinity(0)), world, quota)}
    fail = os.environ.get("JGA_BENCH_FAIL_RANK", "").split(",")
    data = synth.synthetic_jpeg(160, 96, "420", seed=1 + rank)
    _, g = lib.geom_of(data)
    comm.barrier()
    t0 = time.perf_counter()
    for step in range(args.steps):
        lib.entropy_decode(data, g)
        if fail[0] != "" and int(fail[0]) == rank and step == args.steps // 2:
            if fail[1:] == ["die"]:
                os._exit(9)
            raise RuntimeError("injected failure in step %d (JGA_BENCH_FAIL_RANK)" % step)
    comm.barrier()
    rate, dt = comm.throughput(args.steps * 160 * 96, time.perf_counter() - t0)
    everyone = comm.gather(me)
    comm.close()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "steps": args.steps, "ranks": everyone,
                          "pixels_per_s": round(rate, 1)}), flush=True)
    return 0


# ---- ranks talk over RCCL where it comes up, gloo where it does not -------------------------------

class RankFailed(Exception):
    """Some rank of the job failed; .ranks = which (empty: one died without saying), .why = what they said."""

    def __init__(self, ranks, why):
        Exception.__init__(self, why)
        self.ranks, self.why = list(ranks), why


class Comm:
    """The job's control plane.  The data path needs no collective, so all the ranks exchange is
    the timing barrier, a MAX / SUM of scalars and the per-rank report.  The default group is
    gloo (CPU tensors: always works, carries all_gather_object); an RCCL group (backend "nccl")
    is created on top and used for the barrier and the reductions when every rank could bring it
    up — decided together, so that nobody waits on a collective the others never enter."""

    def __init__(self, torch, dist, rank, world, gpu, share):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.group, self.device, self.backend, self.note = None, "cpu", "none", None
        if world <= 1:
            return
        import datetime as _dt
        self.timeout = _dt.timedelta(seconds=int(os.environ.get("JGA_BENCH_COMM_TIMEOUT_S", "300")))
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with quiet_stdout():
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=self.timeout)
            dist.barrier()                                           # (connects the pairs now, while stdout is away)
        self.backend = "gloo"
        if share or os.environ.get("JGA_BENCH_NO_RCCL") == "1":
            self.note = ("ranks share devices (JGA_BENCH_SHARE_GPUS): RCCL refuses two ranks per GPU" if share
                         else "JGA_BENCH_NO_RCCL=1")
            return
        ok, why, grp = 1, "", None
        try:
            with quiet_stdout():
                grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120),
                                     device_id=torch.device("cuda", gpu))
                t = torch.ones(1, device="cuda")
                dist.all_reduce(t, group=grp)
                torch.cuda.synchronize()
            ok = int(t.item() == world)
            why = "" if ok else "first all-reduce returned %r" % t.item()
        except Exception as e:                      # RCCL / IPC not usable on this box
            ok, why = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # (gloo)
        if int(flag.item()) == 1:
            self.group, self.device, self.backend = grp, "cuda", "nccl (RCCL)"
        else:
            self.note = "RCCL group not usable (%s): barrier and reductions over gloo" % (why or "another rank failed")
            log("rank %d: %s" % (rank, self.note))

    # -- is everybody still there?  Before every collective the ranks exchange their state over gloo: a rank that
    # failed (an exception anywhere in its body: a job of the timed region, one wrong byte, a HIP error) enters the
    # same exchange from its handler with ok = False, so the others learn WHO failed and why instead of waiting in a
    # barrier until the backend's timeout; a rank that died outright (abort, kill) closes its sockets and the
    # exchange raises on the others at once.  Either way every rank leaves non-zero and rank 0's line says which.
    def status(self, ok=True, err=None):
        if self.world <= 1:
            return
        mine = {"rank": self.rank, "ok": bool(ok), "err": err}
        everyone = [None] * self.world
        try:
            self.dist.all_gather_object(everyone, mine)                  # (gloo)
        except Exception as e:                                           # a peer is gone: its sockets closed under us
            raise RankFailed([], "rank %d lost a peer during the status exchange (%s: %s) - a rank died without "
                             "reporting" % (self.rank, type(e).__name__, str(e)[:160]))
        bad = [r for r in everyone if r and not r["ok"]]
        if bad and ok:
            raise RankFailed([r["rank"] for r in bad],
                             "; ".join("rank %d: %s" % (r["rank"], r["err"]) for r in bad))

    def barrier(self):
        if self.world > 1:
            self.status()
            if self.group is not None:
                self.dist.barrier(group=self.group)
            else:
                self.dist.barrier()

    def reduce(self, value, op):
        if self.world <= 1:
            return float(value)
        self.status()
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op), group=self.group)
        return float(t.item())

    def throughput(self, units, seconds):
        """Whole-job units/s: sum of units over ranks / max of time over ranks."""
        t = self.reduce(seconds, "MAX")
        return self.reduce(units, "SUM") / t, t

    def gather(self, obj):
        if self.world <= 1:
            return [obj]
        self.status()
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)                        # (gloo)
        return out

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


# ---- inputs -----------------------------------------------------------------------------------

def make_inputs(synth, n, rank, world, threads, comm):
    """The rank's `n` distinct files (SURVEY.md §8(d) recipe).  Synthesised ONCE per box: each rank
    makes its share (seed i with i % world == rank) into a cache under /dev/shm, keyed by the
    recipe and the seed, and after a barrier every rank reads all of them — each into its own
    memory, in its own rotation, so no two ranks decode the same file at the same moment.  Eight
    ranks on a 16-CPU grant used to synthesise 8 x 64 files; now 64."""
    seeds = [1234 + i for i in range(n)]
    cache = os.environ.get("JGA_BENCH_CACHE", "/dev/shm/jga_bench_%d" % os.getuid())
    name = lambda s: os.path.join(cache, "%dx%d_%s_q%d_s%d.jpg" % (W, H, SAMPLING, QUALITY, s))
    try:
        os.makedirs(cache, exist_ok=True)
        usable = os.access(cache, os.W_OK)
    except OSError:
        usable = False
    if not usable:
        if world > 1:
            comm.barrier()
        with ThreadPoolExecutor(max_workers=max(1, min(n, threads))) as ex:
            files = list(ex.map(lambda s: synth.synthetic_jpeg(W, H, SAMPLING, QUALITY, seed=s), seeds))
        return files[rank % n:] + files[:rank % n], "synthesised by every rank (no /dev/shm)"

    def make(s):
        if not os.path.exists(name(s)):
            data = synth.synthetic_jpeg(W, H, SAMPLING, QUALITY, seed=s)
            tmp = name(s) + ".%d.tmp" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(data)
            os.replace(tmp, name(s))
    mine = [s for i, s in enumerate(seeds) if i % world == rank]
    with ThreadPoolExecutor(max_workers=max(1, min(len(mine) or 1, threads))) as ex:
        list(ex.map(make, mine))
    comm.barrier()
    files = [open(name(s), "rb").read() for s in seeds]
    k = (rank * max(1, n // max(world, 1))) % n
    return files[k:] + files[:k], "%d files synthesised once per box (%s), each rank reads them all" % (n, cache)


def device_facts(torch, dev):
    p = torch.cuda.get_device_properties(dev)
    return {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
            "hbm_GB": round(p.total_memory / 2**30, 1),
            "clock_MHz": getattr(p, "clock_rate", 0) // 1000 or None}


# ---- CPU baselines ----------------------------------------------------------------------------

def cpu_baseline(jpegs, rounds, frames, cpus=None, quota=None):
    """north_star: "the xjpeg/libjpeg-turbo CPU path timed on the same box's host cores in the
    same run (core count stated)" — tools/cpu_paths.py: one frame loop per CPU the box really
    grants (all of `cpus`, the whole box, not one rank's NUMA share; or the cgroup's grant when
    that is smaller), every loop on its own image; warm; best of `rounds`."""
    import cpu_paths
    mine = os.sched_getaffinity(0)
    if cpus:
        os.sched_setaffinity(0, cpus)            # the loops' threads inherit it
    threads, ncpu = cpu_paths.granted_threads(None, quota)
    res = {"unit": "Mpixel/s", "cores": threads, "cpu_model": cpu_paths.cpu_model(),
           "visible_cpus": ncpu, "cgroup_cpu_quota": quota,
           "method": "one frame loop per granted CPU (reset -> header -> decode, as "
                     "src/jpeg_gpu.c:1231-1237), %d frames per loop per round, best of %d rounds, "
                     "each loop on its own 3840x2160 4:2:0 q90 file" % (frames, rounds)}
    t_all = time.perf_counter()
    res.update(cpu_paths.time_paths(jpegs, W, H, threads, frames, rounds, port=True))
    # the headline CPU number: the reference's own path where its build is here, else the port
    if "reference_xjpeg_yuv" in res:
        res["kind"], res["value"] = "reference", res["reference_xjpeg_yuv"]["value"]
        what = "reference xjpeg YUV stage"
    else:
        res["kind"], res["value"] = "port", res["oracle_port_rgb"]["value"]
        what = "oracle port, RGB"
    res["sample"] = "%s: %d loops x %d frames of %dx%d per round, %d rounds, %.1f s in all " \
                    "three CPU paths" % (what, threads, frames, W, H, rounds,
                                         time.perf_counter() - t_all)
    os.sched_setaffinity(0, mine)
    return res


# ---- HBM traffic of the fused kernel (PMC) ------------------------------------------------------

def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


def rocprof_here():
    import shutil
    return bool(shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"))


def measure_traffic(batch, budget_s=300):
    """rocprofv3 --pmc passes over a child that launches the fused kernel on a resident batch
    (tools/pmc_traffic.py; counters and corrections as MI355X_MICROARCH.md prescribes).  Writes
    profiles/pmc_latest.json and returns it, or None when the tool is not on the box / runs out
    of its time budget."""
    tool = os.path.join(ROOT, "tools", "pmc_traffic.py")
    try:
        r = subprocess.run([sys.executable, tool, "--batch", str(batch)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=budget_s, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("PMC ")]
        if r.returncode == 0 and line:
            return json.loads(line[0][4:])
        log("bench.py: traffic measurement failed:\n" + r.stderr[-1500:])
    except Exception as e:           # no rocprofv3, timeout, ...
        log("bench.py: traffic measurement unavailable (%s)" % e)
    return None


def quoted_traffic(batch):
    """roofline.traffic from the committed PMC summary, with where and when it was taken."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return None, {}
    if pmc.get("batch") != batch or pmc.get("workload") != "%dx%d %s" % (W, H, SAMPLING):
        return None, {}
    prov = {"source": "profiles/pmc_latest.json", "measured_in_this_run": False,
            "taken_at_head": pmc.get("head"), "taken_on": pmc.get("date"),
            "from": pmc.get("source")}
    valu = {k: pmc[k] for k in ("valu_insts_per_wave", "valu_busy_4clk") if k in pmc}
    return pmc.get("hbm_bytes_per_launch"), dict(provenance=prov, valu=valu)


# ---- every image of a timed region, against the oracle -------------------------------------------

class KeptOutputs:
    """One HBM buffer holding the pixels of every image of a timed region (slot k at k*pitch:
    evenly spaced, so the pipeline writes a group with one launch), and the check of all of
    them against the oracle's pixels after the clock has stopped."""

    def __init__(self, torch, n_images, out_bytes, group, max_bytes, nfiles=1):
        self.torch, self.out_bytes = torch, out_bytes
        self.pitch = (out_bytes + 255) // 256 * 256
        free, _ = torch.cuda.mem_get_info()
        room = int(min(max_bytes, free - (28 << 30)))               # leave the lanes their buffers (8 x ~2.5 GB)
        # When not every output fits, jobs k and k + slots share a slot: slots is a multiple of the
        # number of distinct files (job k decodes file (13 + k) % nfiles), so that both decode the SAME
        # file and whichever lane writes last leaves the same pixels (lanes finish out of order: ADVICE
        # r3) — and of the group size, so that no group straddles the wrap.
        import math
        unit = group * nfiles // math.gcd(group, nfiles)
        slots = max(unit, min(n_images, room // self.pitch))
        self.slots = n_images if slots >= n_images else slots // unit * unit
        self.n_images = n_images
        self.buf = torch.empty(self.slots * self.pitch, dtype=torch.uint8, device="cuda")

    def ptrs(self):
        base = self.buf.data_ptr()
        return [base + (k % self.slots) * self.pitch for k in range(self.n_images)]

    def verify(self, file_of_job, reference_of_file):
        """file_of_job[k] = which distinct file job k decoded; reference_of_file[f] = the oracle's
        pixels of it (a cuda uint8 tensor).  Slots hold the LAST job written to them.  Returns
        (images verified, indices of the jobs whose pixels differ)."""
        torch = self.torch
        last = {}
        for k in range(self.n_images):
            last[k % self.slots] = k
        bad = []
        inject = os.environ.get("JGA_BENCH_CORRUPT")
        if inject is not None:                                       # tests: one flipped byte must fail the run
            self.buf[(int(inject) % self.slots) * self.pitch + 4321] ^= 0xFF
        for slot, k in sorted(last.items()):
            got = self.buf[slot * self.pitch: slot * self.pitch + self.out_bytes]
            if not torch.equal(got, reference_of_file[file_of_job[k]]):
                bad.append(k)
        return len(last), bad

    def free(self):
        self.buf = None
        self.torch.cuda.empty_cache()



# ---- what rank 0 prints -------------------------------------------------------------------------

LINE_LIMIT = 4096             # bytes the ONE stdout line may take, at N = 1 and at N = 8 (a 23.7 KB line
                              # was more than the driver's reader took, and the round's headline went unrecorded)
DETAILS_FILES = ("bench_details.json", os.path.join("gpurun_out", "bench_details.json"))


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=120):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _config_summary(e):
    """One BASELINE config's entry of the full report -> the handful of figures the line carries (rates in whole
    Mpixel/s; everything else — CPU paths, device-only times, per-leg detail — is in the details file)."""
    if not isinstance(e, dict) or "to_rgb_hbm" not in e:
        return e if not isinstance(e, dict) else _pick(e, "error")
    t, dev, host = e["to_rgb_hbm"], e.get("device", {}), e.get("to_host_pixels", {})
    s = {}
    if "Mpixel_s" in t:
        s["Mpixel_s"] = int(round(t["Mpixel_s"]))
    if "latency_ms" in t:
        s["lat_ms"] = t["latency_ms"]
    st = t.get("steady")
    if st:
        s["steady"] = int(round(st["Mpixel_s"]))
        s["steady_s"] = st["seconds"]
        s["of_link_ceiling"] = st["of_link_ceiling"]
    for k, v in t.items():                                   # config 4: the whole batch and one rank's shard
        if isinstance(v, dict) and "pageable_files" in v:
            pg, pn = v["pageable_files"], v["pinned_files"]
            s[k] = {"ms": [pg["ms"], pn["ms"]], "best": [pg["ms_best"], pn["ms_best"]]}      # [pageable, pinned files]
    if "ms_per_frame" in host:
        s["host_ms"] = host["ms_per_frame"]
    elif isinstance(host.get("plugin"), dict):
        s["host_ms"] = host["plugin"].get("ms_per_frame")
    if "kernel_hbm_frac" in dev:
        s["kernel_hbm_frac"] = dev["kernel_hbm_frac"]
    if "shard_128" in dev:
        s["dev_shard_ms"] = dev["shard_128"]["ms"]
    for k in ("huffman_ms", "idct_rgb_ms"):
        if k in dev:
            s[k] = dev[k]
    if "bound_by" in e:
        s["bound_by"] = e["bound_by"]
        s["B_per_px"] = e.get("bytes_per_pixel")
    s["ok"] = bool(e.get("bit_exact_vs_oracle"))
    return s


def compact_line(full, details_path=None):
    """The full report -> the ONE line rank 0 prints: the contract keys, `roofline`, `cpu_baseline`, one compact
    summary per config and per supplementary leg.  Everything else stays in the details file."""
    o = _pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "value_pinned_ingest")
    cfg = full.get("config", {})
    o["config"] = dict(_pick(cfg, "workload", "batch_per_gpu", "images_timed_per_gpu", "images_verified",
                             "h2d_bytes_per_image", "h2d_GBps_per_gpu_at_value", "scan_cleanup",
                             "ranks_talk_over", "parallelism", "host_threads_per_gpu", "bit_exact_vs_oracle"))
    o["config"]["workload"] = _short(o["config"].get("workload", ""), 128)
    if "host_cpus" in cfg:
        o["config"]["cpu_budget_per_rank"] = cfg["host_cpus"].get("budget_per_rank")
    if "device" in cfg:
        o["config"]["device"] = "%s %s" % (cfg["device"].get("arch", "").split(":")[0], cfg["device"].get("compute_units"))
    rf = full.get("roofline", {})
    o["roofline"] = _pick(rf, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                          "algorithmic_bytes_per_launch", "kernel_ms_per_launch", "launches_timed", "device_copy_GBps")
    if "traffic_provenance" in rf:
        o["roofline"]["traffic_measured_in_this_run"] = bool(rf["traffic_provenance"].get("measured_in_this_run"))
    cb = full.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind")
        o["cpu_baseline"]["sample"] = _short(cb.get("sample", ""), 128)
        for k in ("libjpeg_turbo_rgb", "oracle_port_rgb"):
            if k in cb:
                o["cpu_baseline"][k] = cb[k].get("value")
    pr = full.get("per_rank")
    if pr:
        rk = pr["ranks"]
        o["per_rank"] = {"Mpixel_s": [r["Mpixel_s"] for r in rk], "h2d_GBps": [r.get("h2d_GBps") for r in rk],
                         "cpus": [r.get("cpus") for r in rk], "numa_node": [r.get("numa_node") for r in rk],
                         "verified": sum(r.get("images_verified", 0) for r in rk)}
    e2e = full.get("e2e") or {}
    if e2e:
        short = {"north_star_host_huffman_to_rgb_hbm": "host_huffman_to_hbm", "north_star_host_huffman_to_rgb_host": "host_huffman_to_host",
                 "pack_transport_to_rgb_hbm": "pack_to_hbm", "gpu_entropy_to_rgb_host": "gpu_entropy_to_host",
                 "gpu_entropy_to_rgb_pinned_host": "gpu_entropy_to_pinned_host"}
        o["e2e"] = {short[k]: int(round(v["value"])) for k, v in e2e.items() if isinstance(v, dict) and "value" in v and k in short}
    ge = full.get("gpu_entropy")
    if ge:
        o["gpu_entropy"] = _pick(ge, "huffman_ms", "idct_rgb_ms", "sync_rounds", "bit_exact_vs_oracle")
    ps = full.get("pack_stage")
    if ps:
        o["pack_stage"] = _pick(ps, "ms_per_launch", "achieved_GBps", "equals_oracle_quant_stage")
    ok_ = full.get("other_kernels")
    if ok_:
        o["other_kernels"] = {k: [v["ms"], v["GBps"]] for k, v in ok_.items()}
    sp = full.get("scale_proxy")
    if sp:
        o["scale_proxy"] = {"as_rank_of": sp.get("as_rank_of"), "cpus": sp.get("cpus"), "vs_value": sp.get("vs_value")}
    cf = full.get("configs")
    if cf:
        o["configs"] = {k: _config_summary(v) for k, v in cf.items() if k != "note"}
    if details_path:
        o["details"] = details_path
    line = json.dumps(o, separators=(",", ":"))
    # the limit is a contract: shed the supplementary objects, least important first, rather than exceed it
    for k in ("scale_proxy", "other_kernels", "e2e", "pack_stage", "per_rank", "gpu_entropy", "configs"):
        if len(line) < LINE_LIMIT:
            break
        o.pop(k, None)
        o.setdefault("dropped_for_size", []).append(k)
        line = json.dumps(o, separators=(",", ":"))
    return line


def emit(full):
    """Rank 0: the full report into the details files (repo root and gpurun_out/, whichever can be written), the
    compact line — alone — on stdout."""
    written = None
    for rel in DETAILS_FILES:
        path = os.path.join(ROOT, rel)
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            written = written or rel
        except OSError as e:
            log("bench.py: could not write %s (%s)" % (rel, e))
    print(compact_line(full, written), flush=True)


# ---- main -------------------------------------------------------------------------------------

_COMM = None                  # the rank's control plane once it is up (the failure handler below needs it)


def main():
    """The rank's body inside the failure protocol (Comm.status): a rank that fails says so to the others before it
    leaves; ranks that learn of a failed peer leave non-zero too; rank 0 prints ONE line naming the rank(s)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def failure_line(ranks, why):
        if rank == 0:
            print(json.dumps({"error": _short(why, 600), "failed_ranks": ranks, "n_gpus": world, "metric": METRIC,
                              "value": None}), flush=True)
    try:
        return _rank_main()
    except RankFailed as e:                                  # somebody else failed
        log("rank %d: leaving, a rank of the job failed: %s" % (rank, e.why))
        failure_line(e.ranks, e.why)
        os._exit(3)                                          # (no orderly shutdown: the group is broken)
    except SystemExit as e:
        if e.code in (0, None) or _COMM is None or _COMM.world <= 1:
            raise
        why = str(e.code)
    except BaseException as e:
        if _COMM is None or _COMM.world <= 1:
            raise
        why = "%s: %s" % (type(e).__name__, e)
        import traceback
        traceback.print_exc()
    try:
        _COMM.status(ok=False, err=_short(why, 300))         # tell the others (they are in, or on their way to, an exchange)
    except Exception:
        pass
    failure_line([rank], "rank %d: %s" % (rank, why))
    os._exit(1)


def _rank_main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log("bench.py: --gpus %d but the launcher made %d rank(s); reporting n_gpus = %d"
            % (args.gpus, world, world))
        args.gpus = world
    if args.dry_launch:
        sys.exit(dry_launch(args, rank, local_rank, world))

    import torch                      # first: its bundled HIP runtime must be THE runtime
    import torch.distributed as dist
    import numpy as np
    import ctypes as C
    import __graft_entry__
    __graft_entry__.build()
    from jpeg_gpu_amd import abi, lib, shard, synth

    ndev = lib.device_count() if torch.cuda.is_available() else 0
    # (testing the N-rank path on a box with fewer GPUs: ranks share devices and talk over gloo —
    # RCCL refuses two ranks on one GPU; the numbers then mean nothing, the code path is the same)
    share = ndev >= 1 and ndev < world and os.environ.get("JGA_BENCH_SHARE_GPUS") == "1"
    if ndev < world and not share:
        raise SystemExit("bench.py: %d HIP device(s) visible, %d needed (no CPU fallback exists)"
                         % (ndev, world))
    gpu = local_rank % ndev
    # this rank's host cores: those of its GPU's NUMA node, shared with the ranks next door
    pin = None
    orig_cpus = os.sched_getaffinity(0)
    if not args.no_pin:
        ids = [lib.device_pci_bus_id(i % ndev) for i in range(world)]
        pin = shard.pin_rank_to_gpu_node(local_rank, world, ids)
    my_cpus = len(os.sched_getaffinity(0))
    # ... of which the container may be granted fewer (cgroup cpu.max: the GPU boxes show 256
    # CPUs and grant 16); threads beyond the grant get everybody throttled
    quota = shard.cpu_quota()
    budget = shard.rank_cpu_budget(my_cpus, world, quota)
    if world > 1:
        os.environ["JGA_CPU_BUDGET"] = str(budget)     # the library's defaults: this rank's share, not the box's
    proxy_of = args.as_rank_of if world == 1 else 0
    if proxy_of > 1:
        # scale-proxy child: this process lives on what ONE rank of `proxy_of` gets — its share of the
        # grant, as that many CPUs of the GPU's NUMA node (a harder limit than the real thing: a real rank
        # may burst onto its neighbours' idle cores inside the cgroup's budget, this one cannot)
        budget = shard.rank_cpu_budget(my_cpus, proxy_of, quota if quota else float(len(orig_cpus)))
        mine = sorted(os.sched_getaffinity(0))
        cores = shard.cpu_cores(mine) if hasattr(shard, "cpu_cores") else [[c] for c in mine]
        first = (args.proxy_slot * budget) % max(1, len(cores) - budget + 1)     # (concurrent children: each its own cores)
        take = [core[0] for core in cores[first:first + budget]] or mine[:budget]
        os.sched_setaffinity(0, take)
        my_cpus = len(take)
        os.environ["JGA_CPU_BUDGET"] = str(budget)
        if pin:
            pin = dict(pin, cpus=len(take), cpu_list=",".join(map(str, take)))
    torch.cuda.set_device(gpu)
    lib.check(lib.L.jga_set_device(gpu))
    global _COMM
    comm = _COMM = Comm(torch, dist, rank, world, gpu, share)

    def fence():
        comm.barrier()
        torch.cuda.synchronize()
        lib.check(lib.L.jga_stream_sync(None))

    # ---- inputs: distinct synthetic files in host RAM (196 MB per rank at the defaults: they
    # do not fit any cache level, so the host side reads them from DRAM like real traffic)
    t_setup = time.perf_counter()
    jpegs, inputs_note = make_inputs(synth, args.distinct, rank, world, max(1, min(budget, 64)), comm)
    hdr, g = lib.geom_of(jpegs[0])
    B, K, Wm = args.batch, args.steps, args.warmup
    G = max(1, min(args.group, B))
    # transport 2's host threads parse + unstuff and then wait for the device: 1.5 per granted
    # CPU measured best (profiles/r2_e2e_sweep.txt); with no grant to respect, up to 96
    nthreads = args.host_threads or (min(my_cpus, max(args.lanes, budget + budget // 2)) if quota
                                     else max(1, min(my_cpus, 96)))
    log("rank %d: %d files (%.1f MB) in %.1f s; %d host threads%s%s" % (
        rank, len(jpegs), sum(map(len, jpegs)) / 1e6, time.perf_counter() - t_setup, nthreads,
        ", cpus %s of node %s" % (pin["cpu_list"], pin["numa_node"]) if pin else "",
        ", cgroup grants %.1f CPUs" % quota if quota else ""))

    # ---- the oracle's pixels of every distinct file, resident in HBM for the checks (checker
    # side: computed before any timed region, on this rank's cores)
    import oracle
    orc = oracle.Oracle()
    t_or = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, min(budget, 32))) as ex:
        refs_host = list(ex.map(lambda j: orc.decode_rgb(j)[1].reshape(-1), jpegs))
    refs = [torch.from_numpy(r).cuda() for r in refs_host]
    log("rank %d: oracle pixels of %d files in %.1f s" % (rank, len(refs), time.perf_counter() - t_or))

    # ---- headline: JPEG bytes in host RAM -> RGB8 in HBM, K batches of B images per rank, twice:
    # files in ordinary PAGEABLE memory (`value`: the host copies every scan, cleaning it up on the
    # way or leaving that to the GPU when cores are few) and files in PINNED ingest buffers
    # (jga_host_malloc_pinned: where a reader would put them; with few cores per
    # GPU the library DMAs them where they lie and cleans up on the device).  Every image of both
    # timed regions keeps its pixels and is checked.
    pins = [lib.PinnedBytes(j) for j in jpegs]
    cyc = lambda n, o=0: [jpegs[(o + i) % len(jpegs)] for i in range(n)]
    pcyc = lambda n, o=0: [pins[(o + i) % len(pins)].array for i in range(n)]
    kept = KeptOutputs(torch, K * B, g.rgb_bytes, G, int(args.max_keep_GB * 2**30), len(jpegs))
    file_of_job = [(13 + i) % len(jpegs) for i in range(K * B)]

    import resource

    def headline_run(pinned, **kw):
        src = pcyc if pinned else cyc
        mk = lambda n, o, **k2: lib.Pipeline.make_jobs(src(n, o), pinned=pinned, **k2)
        if not pinned:
            kw.setdefault("input_cache_mb", args.input_cache_MB)
        pl = lib.Pipeline(device=gpu, nthreads=nthreads, out=abi.JPEG_DECODE_RGB, copy_back=False,
                          transport=2, batch=G, depth=args.lanes, **kw)
        setup_jobs = mk(args.lanes * G, 0)                           # lanes allocate their buffers
        warm_jobs = mk(Wm * B, 7) if Wm > 0 else None
        timed_jobs = mk(K * B, 13, dev_outs=kept.ptrs())
        if pl.run_jobs(setup_jobs) != 0:
            raise SystemExit("bench.py: pipeline failed: " + lib.L.jga_last_error().decode())
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm:            # clocks settle (a cold GPU reads
            pl.run_jobs(setup_jobs)                                  # ~4 % low)
        if warm_jobs is not None:
            pl.run_jobs(warm_jobs)                                   # W untimed steps
        kept.buf.zero_()
        if not pinned and kw.get("input_cache_mb", 0) > 0:
            # whatever the warm-up left registered is dropped: every file's FIRST sight — its
            # hipHostRegister — happens inside the timed region
            for v in timed_jobs._keep[0]:
                lib.L.jga_pipeline_forget_input(pl.ptr, v.ctypes.data)
        c0 = pl.counters()
        fence()
        if proxy_of > 1 and args.proxy_dir:
            # N scale-proxy children alive at once: they enter their timed regions together
            tag = "pinned" if pinned else "pageable"
            open(os.path.join(args.proxy_dir, "%s.%d" % (tag, args.proxy_slot)), "w").close()
            t_meet = time.perf_counter()
            while sum(f.startswith(tag + ".") for f in os.listdir(args.proxy_dir)) < proxy_of:
                if time.perf_counter() - t_meet > 300:
                    raise SystemExit("bench.py: scale-proxy child %d: the others never arrived" % args.proxy_slot)
                time.sleep(0.002)
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        rc = pl.run_jobs(timed_jobs)                                 # exactly K steps; returns when
        fence()                                                      # every output is complete
        dt_local = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        c1 = pl.counters()
        pl.close()
        if rc != 0 or any(j.status != 0 for j in timed_jobs):
            raise SystemExit("bench.py: a job of the timed region failed: " + lib.L.jga_last_error().decode())
        nver, bad = kept.verify(file_of_job, refs)
        if bad:
            raise SystemExit("bench.py: rank %d: %d of %d images of the timed region differ from the oracle "
                             "(first: job %d, file %d)" % (rank, len(bad), nver, bad[0], file_of_job[bad[0]]))
        rate, dt = comm.throughput(K * B * W * H, dt_local)
        nj = len(timed_jobs)
        return {"rate": rate, "dt": dt, "dt_local": dt_local, "verified": nver,
                "h2d": sum(j.h2d_bytes for j in timed_jobs) // nj,
                # what the host side spent on the timed region: CPU seconds of the whole process (all
                # threads, user + system) and file bytes a host core read, per image
                "cpu_ms_per_image": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / nj * 1e3, 4),
                "host_bytes_per_image": sum(j.host_bytes for j in timed_jobs) // nj,
                "scan_cleanup": "device" if c1["cleanup_on_device"] else "host",
                "registered_in_timed_region": c1["registered"] - c0["registered"],
                "register_ms_total": round((c1["register_us"] - c0["register_us"]) / 1e3, 2),
                "jobs_dma_in_place": c1["jobs_in_place"] - c0["jobs_in_place"]}

    page = headline_run(False)
    pinn = headline_run(True)
    if proxy_of > 1:
        # the scale-proxy child's whole report
        keys = ("cpu_ms_per_image", "host_bytes_per_image", "scan_cleanup", "registered_in_timed_region",
                "register_ms_total", "jobs_dma_in_place")
        print("SCALE_PROXY " + json.dumps({
            "as_rank_of": proxy_of, "slot": args.proxy_slot, "cpus": my_cpus, "cpu_list": sorted(os.sched_getaffinity(0)),
            "cpu_budget": budget, "host_threads": nthreads, "images": K * B, "images_verified": page["verified"] + pinn["verified"],
            "pageable": dict({"Mpixel_s": round(page["rate"] / 1e6, 1), "ms_per_step": round(page["dt"] / K * 1e3, 4)},
                             **{k: page[k] for k in keys}),
            "pinned": dict({"Mpixel_s": round(pinn["rate"] / 1e6, 1), "ms_per_step": round(pinn["dt"] / K * 1e3, 4)},
                           **{k: pinn[k] for k in keys})}), flush=True)
        comm.close()
        return
    rate, dt = page["rate"], page["dt"]
    h2d_per_image = page["h2d"]
    ok = True
    forced = {}
    if world == 1 and not args.no_e2e:
        # both scan clean-ups forced on the pinned files, so that the line shows each whatever `auto` chose
        for name, mode in (("pinned_files_host_cleanup", 1), ("pinned_files_device_cleanup", 2)):
            r_ = headline_run(True, unstuff=mode)
            forced[name] = {"value": round(r_["rate"] / 1e6, 1), "unit": "Mpixel/s", "images_per_gpu": K * B,
                            "images_verified": r_["verified"], "ok": True,
                            "ms_per_step": round(r_["dt"] / K * 1e3, 4), "scan_cleanup": r_["scan_cleanup"],
                            "cpu_ms_per_image": r_["cpu_ms_per_image"]}
        # ... and the route a rank with few cores takes, on THIS box's full grant: pageable files through
        # the input cache (registered at first sight inside the timed region, DMA'd in place afterwards),
        # clean-up on the device
        if args.input_cache_MB > 0:
            r_ = headline_run(False, unstuff=2)
            forced["pageable_files_input_cache_device_cleanup"] = {
                "value": round(r_["rate"] / 1e6, 1), "unit": "Mpixel/s", "images_per_gpu": K * B,
                "images_verified": r_["verified"], "ok": True, "ms_per_step": round(r_["dt"] / K * 1e3, 4),
                **{k: r_[k] for k in ("scan_cleanup", "cpu_ms_per_image", "host_bytes_per_image",
                                      "registered_in_timed_region", "register_ms_total", "jobs_dma_in_place")}}
    kept_slots = kept.slots
    kept.free()
    for p_ in pins:
        p_.free()
    # ---- scale proxy (N = 1): what ONE rank of `--scale-proxy` would reach on its share of the host — the
    # headline repeated by a child process confined to grant / N CPUs of this GPU's NUMA node (a real
    # 8-GPU node is not ours to launch; the driver measures the curve, this is its host-side predictor)
    scale_proxy = None
    if world == 1 and rank == 0 and args.scale_proxy > 1:
        t_sp = time.perf_counter()
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(K), "--warmup", str(Wm),
               "--batch", str(B), "--group", str(G), "--lanes", str(args.lanes), "--distinct", str(args.distinct),
               "--as-rank-of", str(args.scale_proxy), "--input-cache-MB", str(args.input_cache_MB),
               "--max-keep-GB", str(args.max_keep_GB), "--prewarm", str(args.prewarm)] + (["--no-pin"] if args.no_pin else [])
        try:
            os.sched_setaffinity(0, orig_cpus)                      # (the child pins itself, from the whole mask)
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                               env={k: v for k, v in os.environ.items() if k != "JGA_CPU_BUDGET"})
            line = [l for l in r.stdout.splitlines() if l.startswith("SCALE_PROXY ")]
            if r.returncode == 0 and line:
                scale_proxy = json.loads(line[0][len("SCALE_PROXY "):])
                scale_proxy["seconds"] = round(time.perf_counter() - t_sp, 1)
            else:
                log("bench.py: scale-proxy child failed (rc %d):\n%s" % (r.returncode, r.stderr[-1500:]))
        except Exception as e:
            log("bench.py: scale-proxy child unavailable (%s)" % e)
        finally:
            if pin:
                shard.pin_rank_to_gpu_node(local_rank, world, ids)
    # ---- ... and the same N children ALIVE AT ONCE on this one GPU, each confined to its own share of the cores:
    # what a single child cannot show — N processes registering buffers, parsing markers and polling at once on
    # the host's grant.  The device is shared, so the children's SUM is what compares with `value`.
    proxy_concurrent = None
    if world == 1 and rank == 0 and args.scale_proxy > 1 and not args.no_concurrent_proxy and scale_proxy:
        import tempfile
        t_sp = time.perf_counter()
        n_ch = args.scale_proxy
        steps_c = max(2, K // 4)
        meet = tempfile.mkdtemp(prefix="jga_proxy_")
        keep_gb = max(4.0, min(args.max_keep_GB, 160.0) / n_ch)
        base = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps_c), "--warmup", "1",
                "--batch", str(B), "--group", str(G), "--lanes", str(args.lanes), "--distinct", str(args.distinct),
                "--as-rank-of", str(n_ch), "--input-cache-MB", str(args.input_cache_MB), "--proxy-dir", meet,
                "--max-keep-GB", "%.1f" % keep_gb, "--prewarm", "0.1"] + (["--no-pin"] if args.no_pin else [])
        try:
            os.sched_setaffinity(0, orig_cpus)
            env = {k: v for k, v in os.environ.items() if k != "JGA_CPU_BUDGET"}
            procs = [subprocess.Popen(base + ["--proxy-slot", str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env) for k in range(n_ch)]
            kids = []
            for pr in procs:
                try:
                    so, se = pr.communicate(timeout=900)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    so, se = pr.communicate()
                line = [l for l in so.splitlines() if l.startswith("SCALE_PROXY ")]
                if pr.returncode == 0 and line:
                    kids.append(json.loads(line[0][len("SCALE_PROXY "):]))
                else:
                    log("bench.py: a concurrent scale-proxy child failed (rc %s):\n%s" % (pr.returncode, se[-800:]))
            if len(kids) == n_ch:
                proxy_concurrent = {"children": n_ch, "steps_each": steps_c, "images_each": steps_c * B,
                                    "seconds": round(time.perf_counter() - t_sp, 1)}
                for leg in ("pageable", "pinned"):
                    px = steps_c * B * W * H
                    slow = max(k_[leg]["ms_per_step"] for k_ in kids) * steps_c / 1e3
                    rates = [k_[leg]["Mpixel_s"] for k_ in kids]
                    proxy_concurrent[leg] = {
                        "sum_Mpixel_s": round(n_ch * px / slow / 1e6, 1),
                        "child_Mpixel_s_min_max": [min(rates), max(rates)],
                        "slowest_child_share": round(min(rates) / sum(rates), 4),
                        "cpu_ms_per_image_min_max": [min(k_[leg]["cpu_ms_per_image"] for k_ in kids),
                                                     max(k_[leg]["cpu_ms_per_image"] for k_ in kids)],
                        "host_bytes_per_image": max(k_[leg]["host_bytes_per_image"] for k_ in kids),
                        "registered_in_timed_region": sum(k_[leg]["registered_in_timed_region"] for k_ in kids),
                        "register_ms_total": round(sum(k_[leg]["register_ms_total"] for k_ in kids), 2),
                        "scan_cleanup": kids[0][leg]["scan_cleanup"]}
                proxy_concurrent["cpu_lists"] = [k_["cpu_list"] for k_ in sorted(kids, key=lambda k_: k_["slot"])]
                proxy_concurrent["images_verified"] = sum(k_["images_verified"] for k_ in kids)
        except Exception as e:
            log("bench.py: concurrent scale proxy unavailable (%s)" % e)
        finally:
            import shutil
            shutil.rmtree(meet, ignore_errors=True)
            if pin:
                shard.pin_rank_to_gpu_node(local_rank, world, ids)
    cleanup_route = page["scan_cleanup"]           # (what the library's `unstuff = 0` came to: jga_pipeline_counters)
    me = {"rank": rank, "gpu": gpu, "pci": (lib.device_pci_bus_id(gpu) if ndev else None),
          "numa_node": pin["numa_node"] if pin else None, "cpus": my_cpus,
          "cpu_list": pin["cpu_list"] if pin else None, "cpu_budget": budget, "host_threads": nthreads,
          "Mpixel_s": round(K * B * W * H / page["dt_local"] / 1e6, 1),
          "ms_per_step": round(page["dt_local"] / K * 1e3, 4),
          "Mpixel_s_pinned_ingest": round(K * B * W * H / pinn["dt_local"] / 1e6, 1),
          # (what this rank's link carried: a slow PCIe root or a rank on the wrong NUMA node shows here)
          "h2d_GBps": round(K * B * page["h2d"] / page["dt_local"] / 1e9, 1),
          "h2d_GBps_pinned_ingest": round(K * B * pinn["h2d"] / pinn["dt_local"] / 1e9, 1),
          "scan_cleanup_pinned_ingest": pinn["scan_cleanup"],
          "cpu_ms_per_image": page["cpu_ms_per_image"],
          "images_verified": page["verified"] + pinn["verified"], "scan_cleanup": cleanup_route}
    per_rank = comm.gather(me)
    PB, B = B, args.kernel_batch       # from here on B = images per launch of the stand-alone kernel legs

    # ---- roofline: the fused kernel alone, coefficient planes resident in HBM ----
    cstride = (g.coef_shorts * 2 + 255) // 256 * 128          # shorts, 256-B aligned
    ostride = (g.rgb_bytes + 255) // 256 * 256
    d_coef = lib.DeviceBuffer(cstride * 2 * B)
    d_q = lib.DeviceBuffer(3 * 64 * 2 * B)
    d_out = lib.DeviceBuffer(ostride * B)
    ncoef = min(len(jpegs), 6)
    coefs = [lib.entropy_decode(j, g) for j in jpegs[:ncoef]]
    qt = np.zeros((B, 3, 64), np.uint16)
    for i in range(B):
        d_coef.upload(coefs[i % ncoef], offset=i * cstride * 2)
        qt[i] = lib.qtab_of(lib.parse_header(jpegs[i % ncoef]))
    d_q.upload(qt)
    stream = lib.L.jga_stream_create()

    def launch(reps):
        ms = C.c_float()
        lib.check(lib.L.jga_time_idct_batch(C.byref(g), B, d_coef.ptr, cstride, d_q.ptr, 1,
                                            d_out.ptr, ostride, 1, reps, stream, C.byref(ms)))
        return ms.value
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:     # clocks settle (a cold GPU reads ~5 % low)
        launch(20)
    launch(5)
    ev_ms = launch(args.kernel_reps)                      # HIP events on `stream` around the launches
    ev_ms = comm.reduce(ev_ms, "MAX")
    for i in range(min(B, ncoef)):                        # every distinct image of the launch
        if not np.array_equal(d_out.download(g.rgb_bytes, offset=i * ostride), refs_host[i]):
            raise SystemExit("bench.py: rank %d: kernel output differs from the oracle (image %d)" % (rank, i))
    alg_bytes = B * (g.coef_blocks * 128 + g.rgb_bytes)
    achieved = alg_bytes / (ev_ms * 1e-3) / 1e9
    traffic, extra = quoted_traffic(B)
    want_traffic = args.measure_traffic if args.measure_traffic is not None else rocprof_here()
    if want_traffic and rank == 0 and world == 1:
        t_pmc = time.perf_counter()
        pmc = measure_traffic(B)
        log("bench.py: roofline.traffic %s in %.0f s" % ("measured" if pmc else "NOT measured (quoting the committed file)",
                                                          time.perf_counter() - t_pmc))
        if pmc:
            traffic = pmc.get("hbm_bytes_per_launch")
            extra = dict(provenance={"source": "rocprofv3 --pmc child passes of this run "
                                               "(tools/pmc_traffic.py)", "measured_in_this_run": True,
                                     "taken_at_head": pmc.get("head")},
                         valu={k: pmc[k] for k in ("valu_insts_per_wave", "valu_busy_4clk") if k in pmc})

    # ---- the north-star transport at every N: host Huffman threads -> pinned hipMemcpyAsync ->
    # fused kernel (entropy.c on this rank's cores; 24.9 MB of planes per image over PCIe)
    e2e = {}
    e2e["pageable_files_to_rgb_hbm"] = {
        "value": round(page["rate"] / 1e6, 1), "unit": "Mpixel/s", "images_per_gpu": K * PB, "ok": True,
        "images_verified": page["verified"], "ms_per_step": round(page["dt"] / K * 1e3, 4),
        "note": "= `value`: the files are ordinary pageable buffers; aggregated over ranks"}
    e2e["pinned_files_to_rgb_hbm"] = {
        "value": round(pinn["rate"] / 1e6, 1), "unit": "Mpixel/s", "images_per_gpu": K * PB, "ok": True,
        "images_verified": pinn["verified"], "ms_per_step": round(pinn["dt"] / K * 1e3, 4),
        "note": "= `value_pinned_ingest`: the files lie in pinned ingest buffers (jga_host_malloc_pinned)"}
    e2e.update(forced)
    if not args.no_e2e:
        # Huffman threads also block on their slot's event: ~3 per granted CPU measured best
        # (profiles/r2_t0_sweep.txt: 32-48 threads on a 16-CPU grant, fewer AND more are slower)
        nthr = min(my_cpus, 3 * budget) if quota else max(1, min(my_cpus, 96))
        # (long enough that the cgroup's burst allowance is spent: 192 images read 13-16 Gpixel/s with
        #  21-26 CPUs busy on a 16-CPU grant, 768 the sustained 10-11: profiles/r4_north_star_probe.txt)
        n0 = max(384, 16 * nthr)
        pl0 = lib.Pipeline(device=gpu, nthreads=nthr, out=abi.JPEG_DECODE_RGB,
                           copy_back=False, transport=0)
        j_warm, j_run = lib.Pipeline.make_jobs(cyc(2 * nthr)), lib.Pipeline.make_jobs(cyc(n0, 3))
        pl0.run_jobs(j_warm)
        fence()
        c0 = sum(os.times()[:2])
        t0 = time.perf_counter()
        rc0 = pl0.run_jobs(j_run)
        cpu_ns = sum(os.times()[:2]) - c0
        fence()
        r0, t_ns = comm.throughput(n0 * W * H, time.perf_counter() - t0)
        pl0.close()
        e2e["north_star_host_huffman_to_rgb_hbm"] = {
            "value": round(r0 / 1e6, 1), "unit": "Mpixel/s", "images_per_gpu": n0,
            "host_threads_per_gpu": nthr, "ok": rc0 == 0,
            "h2d_bytes_per_image": int(g.coef_shorts * 2),
            "host_cpu_ms_per_image": round(cpu_ns / n0 * 1e3, 2), "cpus_busy": round(cpu_ns / t_ns, 1),
            "note": "north_star's design: Huffman on the host (csrc/entropy.c), dense coefficient "
                    "planes over PCIe, fused kernel; aggregated over ranks like `value`"}

    out = {
        "metric": METRIC,
        "value": round(rate / 1e6, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": K,
        "warmup": Wm, "ms_per_step": round(dt / K * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        # the same K steps on files lying in pinned ingest buffers (the faster of the two where host
        # cores are few; `value` is the conservative one: ordinary pageable files)
        "value_pageable": round(page["rate"] / 1e6, 1),
        "value_pinned_ingest": round(pinn["rate"] / 1e6, 1),
        "ms_per_step_pinned_ingest": round(pinn["dt"] / K * 1e3, 4),
        # what the host spent per image of the two timed regions (process CPU time, all threads; file
        # bytes read by host cores) and where the scan clean-up ran
        "host_cost": {v: {k: r_[k] for k in ("cpu_ms_per_image", "host_bytes_per_image", "scan_cleanup",
                                             "registered_in_timed_region", "jobs_dma_in_place")}
                      for v, r_ in (("pageable", page), ("pinned_ingest", pinn))},
        "config": {
            "workload": "3840x2160 4:2:0 q90 JPEG files in host RAM -> RGB8 in HBM, %d images/step/GPU, "
                        "all outputs checked vs oracle" % PB,
            "workload_detail": "files in ordinary pageable buffers, end to end; step = one batch of %d images per GPU "
                               "through the pipelined decoder: host marker parse, scan clean-up (unstuffing) on the host "
                               "into pinned memory - or, when the rank has 8 cores or fewer, on the GPU - compressed "
                               "bytes over PCIe, GPU Huffman decode + fused dequant/IDCT/upsample/RGB kernel; %d steps "
                               "streamed through %d lanes per GPU in groups of %d" % (PB, K, args.lanes, G),
            # (the library's rule, csrc/pipeline.cpp: the smaller of the CPU grant and nthreads)
            "scan_cleanup": cleanup_route,
            # (pageable files: with the clean-up on the device they go through the pipeline's input cache —
            # registered at first sight inside the timed region, DMA'd where they lie afterwards; on the host
            # every byte is read by a core anyway and nothing is registered: `host_cost`)
            "input_cache_mb": args.input_cache_MB,
            "batch_per_gpu": PB, "pipeline_group": G, "distinct_images_per_gpu": len(jpegs),
            "images_timed_per_gpu": K * PB, "images_verified": page["verified"],
            "images_verified_pinned_ingest": pinn["verified"],
            "outputs_kept_GB": round(kept_slots * ((g.rgb_bytes + 255) // 256 * 256) / 2**30, 1),
            "h2d_bytes_per_image": int(h2d_per_image),
            "inputs": inputs_note,
            "ranks_talk_over": comm.backend, **({"ranks_talk_note": comm.note} if comm.note else {}),
            # what the link carries per GPU at this rate (it sustains ~56 GB/s from pinned memory,
            # tools/h2d_probe.py): `value` sits within ~10 % of the PCIe ceiling for this content
            "h2d_GBps_per_gpu_at_value": round(rate / world / (W * H) * h2d_per_image / 1e9, 1),
            "parallelism": "image-sharded x%d, one process per GPU, no collective" % world,
            "host_threads_per_gpu": nthreads, "cpu_pinning": pin,
            "host_cpus": {"visible_to_rank": my_cpus, "cgroup_cpu_quota": quota,
                          "budget_per_rank": budget},
            "bit_exact_vs_oracle": ok,
            "rgb_stage_note": "planes are bit-exact against the reference's CPU path (src/dct.c); the "
                              "upsample + RGB rounding is this build's definition (SURVEY.md A.5: the "
                              "reference has no executable one), pinned by the oracle",
            "device": device_facts(torch, gpu),
        },
        "per_rank": {"ranks": per_rank,
                     "min_Mpixel_s": min(r["Mpixel_s"] for r in per_rank),
                     "max_Mpixel_s": max(r["Mpixel_s"] for r in per_rank),
                     "note": "each rank's own wall clock around its own K steps (pageable files); "
                             "`value` = all pixels / the slowest rank's time"},
        "roofline": {
            "kernel": lib.L.jga_kernel_name(C.byref(g), 1).decode(),
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms_per_launch": round(ev_ms, 4),
            "kernel_Mpixel_s": round(B * W * H / ev_ms / 1e3, 1),
            "launches_timed": args.kernel_reps,
            "note": "the fused dequant+IDCT+upsample+RGB kernel on %d images whose coefficient "
                    "planes are resident in HBM; HIP events on the launch stream; NOT the "
                    "end-to-end value" % B,
            **({"traffic_provenance": extra["provenance"]} if extra.get("provenance") else {}),
            **({"valu": extra["valu"]} if extra.get("valu") else {}),
        },
    }
    if e2e:
        out["e2e"] = e2e
    if scale_proxy:
        sp = scale_proxy
        sp["vs_value"] = {"pageable": round(sp["pageable"]["Mpixel_s"] * 1e6 / rate, 3),
                          "pinned": round(sp["pinned"]["Mpixel_s"] * 1e6 / rate, 3)}
        sp["note"] = ("the headline's K steps repeated by a child process confined (sched_setaffinity + "
                      "JGA_CPU_BUDGET) to the CPUs one rank of %d gets on this host: the per-GPU rate to expect "
                      "at N = %d if the host is the limit; same images, every output verified; pageable files go "
                      "through the input cache (hipHostRegister at first sight, inside the timed region)"
                      % (sp["as_rank_of"], sp["as_rank_of"]))
        if proxy_concurrent:
            pc = proxy_concurrent
            pc["vs_value"] = {leg: round(pc[leg]["sum_Mpixel_s"] * 1e6 / rate, 3) for leg in ("pageable", "pinned")}
            # what the host side would have to give N ranks that each run at `value`: the slowest child's CPU time
            # per image x `value`'s images per second x N, against the grant
            per_s = rate / world / (W * H)
            pc["host_cpus_needed_at_value"] = {
                leg: round(pc["children"] * per_s * pc[leg]["cpu_ms_per_image_min_max"][1] / 1e3, 1) for leg in ("pageable", "pinned")}
            pc["host_cpus_granted"] = quota if quota else len(orig_cpus)
            pc["note"] = ("the same %d children ALIVE AT ONCE on this one GPU, each confined to its own share of the cores "
                          "(input cache on, first sights inside the timed regions, which the children enter together): "
                          "sum_Mpixel_s = all their pixels / the slowest child's time - the device and the link are "
                          "shared, so it compares with `value`, not with %d x `value` - and %d PROCESSES time-slice the one device "
                          "(their kernels do not overlap the way one process's lanes do), which is why the sum stays below "
                          "`value` however idle the host is; what it adds to the single child: %d processes registering "
                          "buffers, parsing markers and polling at once on the grant - `host_cpus_needed_at_value` = the CPUs "
                          "that many ranks at full rate would keep busy at the slowest child's CPU time per image"
                          % (pc["children"], pc["children"], pc["children"], pc["children"]))
            sp["concurrent"] = pc
        out["scale_proxy"] = sp

    if rank == 0:
        # what a plain device-to-device copy of the same volume reaches on this box (SURVEY.md
        # 8d: "state the measured copy ceiling next to the spec"): bytes read + bytes written
        # (warmed: clocks ramped for 0.3 s first; 50 repetitions between two HIP events on the launch stream;
        # both directions are counted: a copy of n bytes reads n and writes n.  hipMemcpyDtoDAsync through the
        # library; torch's copy_ of the same tensors beside it)
        nb = alg_bytes // 2
        src, dst = lib.DeviceBuffer(nb), lib.DeviceBuffer(nb)
        ms_c = C.c_float()
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            lib.check(lib.L.jga_time_device_copy(dst.ptr, src.ptr, nb, 10, stream, C.byref(ms_c)))
        reps_c = 50
        lib.check(lib.L.jga_time_device_copy(dst.ptr, src.ptr, nb, reps_c, stream, C.byref(ms_c)))
        ms_c = ms_c.value
        # ... and a kernel of this library that only moves the bytes (csrc/copy_kernel.hip), best of three grids
        nb16 = nb & ~15
        ms_k, kernel_copy = C.c_float(), {}
        for grid in (1024, 2048, 8192):
            lib.check(lib.L.jga_time_kernel_copy(dst.ptr, src.ptr, nb16, grid, 10, stream, C.byref(ms_k)))
            lib.check(lib.L.jga_time_kernel_copy(dst.ptr, src.ptr, nb16, grid, reps_c, stream, C.byref(ms_k)))
            kernel_copy[grid] = round(2 * nb16 / ms_k.value / 1e6, 1)
        src.free(); dst.free()
        tsrc = torch.empty(nb, dtype=torch.uint8, device="cuda")
        tdst = torch.empty_like(tsrc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            tdst.copy_(tsrc)
        e0.record()
        for _ in range(reps_c):
            tdst.copy_(tsrc)
        e1.record()
        torch.cuda.synchronize()
        ms_t = e0.elapsed_time(e1) / reps_c
        del tsrc, tdst
        # the ceiling = the fastest plain copy of the volume this box offers, whoever makes it
        out["roofline"]["device_copy_GBps"] = max(round(2 * nb / ms_c / 1e6, 1), max(kernel_copy.values()))
        out["roofline"]["device_copy"] = {"bytes_each_way": int(nb), "reps": reps_c, "ms": round(ms_c, 4),
                                          "read_GBps": round(nb / ms_c / 1e6, 1), "write_GBps": round(nb / ms_c / 1e6, 1),
                                          "hipMemcpyDtoD_GBps": round(2 * nb / ms_c / 1e6, 1),
                                          "kernel_copy_GBps_by_grid": kernel_copy,
                                          "torch_copy_GBps": round(2 * nb / ms_t / 1e6, 1),
                                          "note": "the kernel's byte volume (half read, half written) moved by hipMemcpyDtoDAsync "
                                                  "(ms, read_GBps, write_GBps), by a copy kernel of this library (csrc/copy_kernel.hip: "
                                                  "16 B per lane per trip, non-temporal stores; by grid size) and by torch's copy_; warmed "
                                                  "0.3 s, 50 repetitions between HIP events on the launch stream; device_copy_GBps = the best"}

    solo = rank == 0 and world == 1
    configs_failed = None
    if solo and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(jpegs, args.cpu_rounds, args.cpu_frames, orig_cpus, quota)

    if solo and not args.no_e2e:
        # the other ends and transports, same measurement (JPEG bytes in host RAM -> pixels)
        nthr = min(my_cpus, 3 * budget) if quota else max(1, min(my_cpus, 96))
        for key, copy_back, transport in (("north_star_host_huffman_to_rgb_host", True, 0),
                                          ("pack_transport_to_rgb_hbm", False, 1),
                                          ("gpu_entropy_to_rgb_host", True, 2)):
            nt = nthreads if transport == 2 else nthr
            p2 = lib.Pipeline(device=gpu, nthreads=nt, out=abi.JPEG_DECODE_RGB,
                              copy_back=copy_back, transport=transport, batch=G, depth=args.lanes)
            n = 24 * G if transport == 2 else max(384, 16 * nthr)
            if copy_back:
                n = min(n, 288)                   # 25 MB of host pixels per image
            outs = [np.zeros(g.rgb_bytes, np.uint8) for _ in range(n)] if copy_back else None
            nw = min(n, args.lanes * G if transport == 2 else 2 * nthr)
            p2.run_jobs(lib.Pipeline.make_jobs(cyc(nw), host_outs=outs[:nw] if outs else None))
            jr = lib.Pipeline.make_jobs(cyc(n, 3), host_outs=outs)
            t0 = time.perf_counter()
            rc2 = p2.run_jobs(jr)
            te = time.perf_counter() - t0
            p2.close()
            e2e[key] = {"value": round(n * W * H / te / 1e6, 1), "unit": "Mpixel/s", "images": n,
                        "ok": rc2 == 0, "host_threads": nt,
                        "h2d_bytes_per_image": int(sum(j.h2d_bytes for j in jr) // n)}
            del outs
        # ... and with the callers' pixel buffers pinned (jga_job.pinned bit 1): D2H straight into
        # them, no staging buffer and no host memcpy — the link is what is left (64 buffers, reused)
        pouts = [lib.PinnedBytes(bytes(g.rgb_bytes)) for _ in range(64)]
        p3 = lib.Pipeline(device=gpu, nthreads=nthreads, out=abi.JPEG_DECODE_RGB, copy_back=True,
                          transport=2, batch=G, depth=args.lanes)
        n = 288
        oc = [pouts[i % 64].array for i in range(n)]
        p3.run_jobs(lib.Pipeline.make_jobs(cyc(args.lanes * G), host_outs=oc[:args.lanes * G], outs_pinned=True))
        jr = lib.Pipeline.make_jobs(cyc(n, 3), host_outs=oc, outs_pinned=True)
        t0 = time.perf_counter()
        rc3 = p3.run_jobs(jr)
        te = time.perf_counter() - t0
        p3.close()
        e2e["gpu_entropy_to_rgb_pinned_host"] = {
            "value": round(n * W * H / te / 1e6, 1), "unit": "Mpixel/s", "images": n, "ok": rc3 == 0,
            "d2h_GBps": round(n * g.rgb_bytes / te / 1e9, 1), "host_threads": nthreads}
        for p_ in pouts:
            p_.free()
        # ... and on lighter content than the recipe's (N(0, 12) noise costs 3 bits per pixel): a
        # smooth image with fine grain, q90, 0.6 bits per pixel; same pipeline
        def photo_like(i):
            r = np.random.default_rng(900 + i)
            xx = np.linspace(0, 1, W, dtype=np.float32)[None, :, None]
            yy = np.linspace(0, 1, H, dtype=np.float32)[:, None, None]
            cc = np.arange(3, dtype=np.float32)[None, None, :]
            img = 128 + 60 * np.sin((6 + i) * xx * (cc + 1)) * np.cos(4 * yy) + r.normal(0, 2, (H, W, 3)).astype(np.float32)
            return synth.encode_pixels(np.clip(img, 0, 255).astype(np.uint8), SAMPLING, QUALITY)
        with ThreadPoolExecutor(max_workers=max(1, min(8, my_cpus))) as ex:
            light = list(ex.map(photo_like, range(8)))
        p4 = lib.Pipeline(device=gpu, nthreads=nthreads, out=abi.JPEG_DECODE_RGB, copy_back=False,
                          transport=2, batch=G, depth=args.lanes)
        lj = lambda n, o=0: lib.Pipeline.make_jobs([light[(o + i) % len(light)] for i in range(n)])
        p4.run_jobs(lj(args.lanes * G))
        p4.run_jobs(lj(8 * B, 3))
        jr = lj(K * B, 5)
        fence()
        t0 = time.perf_counter()
        rc4 = p4.run_jobs(jr)
        fence()
        te = time.perf_counter() - t0
        buf = lib.DeviceBuffer(g.rgb_bytes)
        one = lj(1, 2)
        one[0].dev_out = buf.ptr
        import oracle
        ok4 = rc4 == 0 and p4.run_jobs(one) == 0 and bool(np.array_equal(
            buf.download(g.rgb_bytes), oracle.Oracle().decode_rgb(light[2 % len(light)])[1].reshape(-1)))
        buf.free()
        p4.close()
        e2e["lighter_content_to_rgb_hbm"] = {
            "value": round(K * B * W * H / te / 1e6, 1), "unit": "Mpixel/s", "images": K * B, "ok": ok4,
            "bytes_per_pixel": round(sum(map(len, light)) / len(light) / (W * H), 3),
            "h2d_GBps": round(sum(j.h2d_bytes for j in jr) / te / 1e9, 1),
            "note": "as `value` on a smooth image with fine grain (sigma 2) instead of the recipe's "
                    "sigma-12 noise, same size / sampling / quality: the link has room there and the "
                    "rate follows the device's entropy + block-decode time"}
        e2e["note"] = "all PCIe- and host-inclusive; *_to_rgb_host also copies the pixels back into " \
                      "the callers' host buffers (the plugin's decode_image semantics), " \
                      "*_to_rgb_pinned_host into buffers the caller pinned (no host memcpy)"

    if solo and not args.no_other:
        # Supplementary: the other device stages, each against its own algorithmic bytes
        # (128 B per coded block in, output bytes out), HIP-event timed.
        others = {}

        def time_stage(gg, n, dc, cs, dq, do, os_, rgb, reps=30):
            ms = C.c_float()
            t_w = time.perf_counter()                # the CPU legs let the GPU's clocks drop: ramp them
            while time.perf_counter() - t_w < 0.15:
                lib.check(lib.L.jga_time_idct_batch(C.byref(gg), n, dc, cs, dq, 1, do, os_, rgb, 10,
                                                    stream, C.byref(ms)))
            for r in (10, reps):                     # (fresh buffers: the first launches map pages)
                lib.check(lib.L.jga_time_idct_batch(C.byref(gg), n, dc, cs, dq, 1, do, os_, rgb, r,
                                                    stream, C.byref(ms)))
            return ms.value

        ys = (g.yuv_bytes + 255) // 256 * 256
        d_yuv = lib.DeviceBuffer(ys * B)
        t = time_stage(g, B, d_coef.ptr, cstride, d_q.ptr, d_yuv.ptr, ys, 0)
        ab = B * (g.coef_blocks * 128 + g.yuv_bytes)
        others["yuv_stage_420"] = {"kernel": "jga_idct_yuv_kernel", "ms": round(t, 4),
                                   "GBps": round(ab / t / 1e6, 1), "images": B}
        for rep in range(2):                                  # pass 3 alone on those planes
            t0 = time.perf_counter()
            for _ in range(10 + 20*rep):
                lib.check(lib.L.jga_yuv_rgb_batch(C.byref(g), B, d_yuv.ptr, ys, d_out.ptr, ostride,
                                                  stream))
            lib.check(lib.L.jga_stream_sync(stream))
            t = (time.perf_counter() - t0) / (10 + 20*rep) * 1e3
        ab = B * (g.yuv_bytes + g.rgb_bytes)
        others["yuv_to_rgb_420"] = {"kernel": "jga_yuv_rgb_kernel", "ms": round(t, 4),
                                    "GBps": round(ab / t / 1e6, 1), "images": B}
        d_yuv.free()
        for name, samp, n in (("rgb_444", "444", 24), ("rgb_422", "422", 32), ("grey", "grey", 48)):
            data = synth.synthetic_jpeg(W, H, samp, quality=90, seed=1234)
            h2, g2 = lib.geom_of(data)
            cs2 = (g2.coef_shorts * 2 + 255) // 256 * 128
            os2 = (g2.rgb_bytes + 255) // 256 * 256
            dc2, do2 = lib.DeviceBuffer(cs2 * 2 * n), lib.DeviceBuffer(os2 * n)
            c2 = lib.entropy_decode(data, g2)
            for i in range(n):
                dc2.upload(c2, offset=i * cs2 * 2)
            dq2 = lib.DeviceBuffer(384 * n)
            dq2.upload(np.tile(lib.qtab_of(h2).reshape(-1), n))
            t = time_stage(g2, n, dc2.ptr, cs2, dq2.ptr, do2.ptr, os2, 1)
            ab = n * (g2.coef_blocks * 128 + g2.rgb_bytes)
            others[name] = {"kernel": lib.L.jga_kernel_name(C.byref(g2), 1).decode(),
                            "ms": round(t, 4), "GBps": round(ab / t / 1e6, 1), "images": n}
            dc2.free(); do2.free(); dq2.free()
        out["other_kernels"] = others

    if solo and not args.no_pack:
        # Supplementary: PACK words + block index resident in HBM ->
        # jga_unpack_kernel -> QUANT planes.  Algorithmic bytes: 2 B/word + 4 B/block read,
        # 128 B/block written.
        pw = [lib.entropy_decode_pack(j, g)[:2] for j in jpegs[:ncoef]]
        nidx = int(lib.L.jga_index_count(C.byref(g)))
        pstride = (max(len(p) for p, _ in pw) + 127) // 128 * 128
        hp = np.zeros((B, pstride), np.uint16)
        hi = np.zeros((B, nidx), np.int32)
        for i in range(B):
            p, ix = pw[i % len(pw)]
            hp[i, :len(p)] = p.view(np.uint16)
            hi[i] = ix
        d_pack, d_idx = lib.DeviceBuffer(hp.nbytes), lib.DeviceBuffer(hi.nbytes)
        d_pack.upload(hp)
        d_idx.upload(hi)
        reps = 20
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.check(lib.L.jga_unpack_batch(C.byref(g), B, d_pack.ptr, pstride, pstride,
                                                 d_idx.ptr, nidx, d_coef.ptr, cstride, None))
            lib.check(lib.L.jga_stream_sync(None))
            tu = (time.perf_counter() - t0) / reps
        nblk = sum(g.plane[p].hblocks * g.plane[p].vblocks for p in range(g.nplanes))
        words = sum(len(pw[i % len(pw)][0]) for i in range(B))
        ub = words * 2 + B * nblk * (4 + 128)
        import oracle as _o
        same = bool(np.array_equal(d_coef.download(g.coef_shorts * 2, dtype=np.int16),
                                   _o.Oracle().decode(jpegs[0], _o.QUANT)[1]))
        out["pack_stage"] = {
            "kernel": "jga_unpack_kernel", "ms_per_launch": round(tu * 1e3, 4),
            "achieved_GBps": round(ub / tu / 1e9, 1), "algorithmic_bytes_per_launch": ub,
            "words_per_block": round(words / (B * nblk), 2),
            "pcie_bytes_vs_dense": round((words * 2 + B * nidx * 4) / (B * g.coef_shorts * 2), 3),
            "equals_oracle_quant_stage": same,
        }
        d_pack.free()
        d_idx.free()

    if solo and not args.no_gpu_entropy:
        # Supplementary: the device side of `value` on its own.  Entropy-coded bytes resident in
        # HBM -> self-synchronising parallel Huffman decode -> the fused kernel -> RGB in HBM.
        jobs = cyc(B)
        hb = lib.HuffBatch(B, sum(map(len, jobs)) + 4096 * B)
        t0 = time.perf_counter()
        hb.prepare(jobs)
        lib.check(lib.L.jga_stream_sync(None))
        t_prep = time.perf_counter() - t0
        d_q.upload(hb.qtabs())
        reps = 5
        th = ti = 0.0
        # (as the pipeline runs it: DC values beside the planes, jga_huff_decode_split + *_batch_dc)
        dcs = (g.coef_shorts // 64 + 127) // 128 * 128
        d_dc = lib.DeviceBuffer(dcs * 2 * B)
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            rounds = hb.decode_split(d_coef.ptr, cstride, d_dc.ptr, dcs)     # synchronous (checks errors)
            t1 = time.perf_counter()
            lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), B, d_coef.ptr, cstride, d_dc.ptr, dcs, d_q.ptr, 1,
                                                  d_out.ptr, ostride, None))
            lib.check(lib.L.jga_stream_sync(None))
            t2 = time.perf_counter()
            if rep:                                           # first repetition warms up
                th += t1 - t0
                ti += t2 - t1
        got = d_out.download(g.rgb_bytes, offset=0)
        t0 = time.perf_counter()
        for rep in range(3):                                  # ... and with finished QUANT planes (DC applied in place)
            hb.decode(d_coef.ptr, cstride)
        t_full = (time.perf_counter() - t0) / 3
        d_dc.free()
        import oracle as _o
        same = bool(np.array_equal(got, _o.Oracle().decode_rgb(jobs[0])[1].reshape(-1)))
        hb.close()
        out["gpu_entropy"] = {
            "value": round(B * W * H * reps / (th + ti) / 1e6, 1), "unit": "Mpixel/s",
            "note": "JPEG entropy-coded bytes resident in HBM -> GPU Huffman -> fused kernel "
                    "-> RGB in HBM (device-only timed region, one batch of %d, no overlap)" % B,
            "huffman_ms": round(th / reps * 1e3, 3), "idct_rgb_ms": round(ti / reps * 1e3, 3),
            "huffman_ms_finished_planes": round(t_full * 1e3, 3),
            "sync_rounds": rounds, "bit_exact_vs_oracle": same,
            "prepare_ms_host_parse_unstuff_h2d": round(t_prep * 1e3, 2),
        }

    if solo and not args.no_configs:
        # Every BASELINE.json config beside its CPU path (row d' of the judge's table: "throughput on
        # synthetic JPEGs of the named resolutions ... next to the ... CPU path").  Configs 2-5 are
        # measured by tools/configs_bench.py; the headline entry is this run's own legs.
        # They run in a process of their own (tools/configs_bench.py --child): dozens of short legs, supplementary to
        # the headline — should the device or the runtime fault in one of them (round 5 saw one "Memory access fault
        # by GPU" there in a dozen runs), the line is printed with the error in `configs` instead of not at all.
        import cpu_paths
        import tempfile
        t_cfg = time.perf_counter()
        cpu_threads, _ = cpu_paths.granted_threads(orig_cpus, quota)
        cfgs, child_headline = {}, {}
        with tempfile.TemporaryDirectory() as td:
            hf = os.path.join(td, "headline.jpg")
            with open(hf, "wb") as f:
                f.write(jpegs[0])
            spec = os.path.join(td, "spec.json")
            json.dump({"nthreads": nthreads, "cpu_threads": cpu_threads, "lanes": args.lanes, "group": G,
                       "quick": bool(args.quick_configs), "cpu_affinity": sorted(orig_cpus), "gpu": gpu,
                       "headline_file": hf}, open(spec, "w"))
            try:
                r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools",
                                                                 "configs_bench.py"), "--child", spec],
                                   stdout=subprocess.PIPE, text=True, timeout=1500)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and lines:
                    got = json.loads(lines[-1])
                    cfgs, child_headline = got["configs"], got.get("headline", {})
                else:
                    cfgs = {"error": "the configs process ended with code %d" % r.returncode}
            except Exception as e:
                cfgs = {"error": "the configs process: %s" % e}
        if "error" in cfgs:
            configs_failed = cfgs["error"]
            log("bench.py: configs leg failed: %s" % cfgs["error"])
        cb = out.get("cpu_baseline", {})
        ge = out.get("gpu_entropy", {})
        hl = {"what": "3840x2160 4:2:0 q90 (the configuration the metric is quoted on), %d distinct files" % len(jpegs),
              "file_bytes": len(jpegs[0]),
              "to_rgb_hbm": {"Mpixel_s": out["value"], "Mpixel_s_pinned_files": out["value_pinned_ingest"],
                             "stream_images": K * PB, "h2d_bytes_per_image": int(h2d_per_image)},
              "to_host_pixels": {k: e2e[k] for k in ("gpu_entropy_to_rgb_host", "gpu_entropy_to_rgb_pinned_host")
                                 if k in e2e},
              "device": {"kernel": out["roofline"]["kernel"], "kernel_ms": out["roofline"]["kernel_ms_per_launch"],
                         "kernel_GBps": out["roofline"]["achieved"], "kernel_hbm_frac": out["roofline"]["frac"],
                         "images_per_launch": B,
                         **({"entropy_plus_kernel_ms": round(ge["huffman_ms"] + ge["idct_rgb_ms"], 3),
                             "huffman_ms": ge["huffman_ms"], "idct_rgb_ms": ge["idct_rgb_ms"]} if ge else {})},
              "bit_exact_vs_oracle": ok,
              "cpu": {k: cb[k] for k in ("reference_xjpeg_yuv", "libjpeg_turbo_rgb", "cores") if k in cb}}
        # one 4K frame alone, like the single-image configs (measured by the same child process)
        if "latency_ms" in child_headline:
            hl["to_rgb_hbm"]["latency_ms"] = child_headline["latency_ms"]
        if "plugin" in child_headline:
            hl["to_host_pixels"]["plugin"] = child_headline["plugin"]
        out["configs"] = dict(headline_4k_420=hl, **cfgs)
        out["configs"]["note"] = ("Mpixel/s end to end per BASELINE.json config on ONE MI355X, host RAM -> RGB8 in HBM "
                                  "(to_rgb_hbm) and -> the caller's host pixels (to_host_pixels: the plugin's "
                                  "decode_image semantics), device-only times, and the reference's xjpeg + "
                                  "libjpeg-turbo on the same files and the same %d granted cores; %.0f s"
                                  % (cpu_threads, time.perf_counter() - t_cfg))

    if rank == 0:
        emit(out)
    lib.L.jga_stream_destroy(stream)
    comm.close()
    if configs_failed:
        # (a fault in the configs process must FAIL the run; the line above says what happened)
        raise SystemExit("bench.py: the configs leg failed: %s" % configs_failed)


if __name__ == "__main__":
    main()
