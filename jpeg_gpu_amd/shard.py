"""Image-level sharding for multi-GPU runs: images are independent
units, rank r of W owns a contiguous chunk, there is no data-path collective —
only a host-side barrier and a max-over-ranks of the elapsed time."""


def shard_range(n_items, rank, world):
    """Contiguous, balanced partition: sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_items(items, rank, world):
    r = shard_range(len(items), rank, world)
    return [items[i] for i in r]


def band_of_rank(jpeg, rank, world):
    """ONE frame across `world` GPUs (SURVEY.md §8e's note; include/jpeg_gpu_amd.h: jga_band_plan):
    the band of MCU rows rank `rank` decodes, as (y0, rows, file) — a baseline JPEG file of its
    own that any decode entry point takes; its pixels are rows [y0, y0 + rows) of the frame.  A
    frame with fewer independent rows of MCUs than ranks (no restart markers: one) leaves the
    last ranks without work: (0, 0, None).  No collective: the bands stay where they are decoded."""
    from . import lib
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    bands = lib.band_plan(jpeg, world)
    if rank >= len(bands):
        return 0, 0, None
    b = bands[rank]
    return b.y0, b.rows, lib.band_file(jpeg, b)


def aggregate_throughput(local_units, local_seconds, dist=None, device=None):
    """Whole-job units/s: sum of units over ranks / max of time over ranks.

    `dist` is torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_units / local_seconds, local_units, local_seconds
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return u.item() / t.item(), u.item(), t.item()


# ---- one process per GPU: which host CPUs belong to which rank -----------------------------
# A node has two sockets; a GPU's PCIe root hangs off one of them, and the pipeline's host
# threads (marker parse + unstuffing, or the host Huffman threads of the north-star
# transport) read the JPEG bytes and write pinned memory the GPU then pulls: they should run
# on the GPU's socket (profiles/design_diary_r3_r5.md §5.1: 86-106 vs 108-117 Gpixel/s), and N ranks must share
# the cores, not each start one thread per core of the box.

def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa_node(pci_bus_id, sysfs="/sys/bus/pci/devices"):
    """NUMA node of a PCI device ('0000:c1:00.0'), -1 if the kernel does not say."""
    try:
        with open("%s/%s/numa_node" % (sysfs, pci_bus_id.lower())) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def numa_node_cpus(node, sysfs="/sys/devices/system/node"):
    try:
        with open("%s/node%d/cpulist" % (sysfs, node)) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return []


def cpu_cores(cpus, sysfs="/sys/devices/system/cpu"):
    """Group logical CPUs into physical cores (SMT siblings together), in CPU order."""
    cpus = sorted(cpus)
    have, cores, seen = set(cpus), [], set()
    for c in cpus:
        if c in seen:
            continue
        try:
            with open("%s/cpu%d/topology/thread_siblings_list" % (sysfs, c)) as f:
                sib = [x for x in parse_cpulist(f.read()) if x in have]
        except (OSError, ValueError):
            sib = []
        core = sorted(set(sib) | {c})
        seen.update(core)
        cores.append(core)
    return cores


def rank_cpu_share(local_rank, gpu_nodes, allowed, node_cpus, cores_of=None):
    """CPUs for the rank that drives GPU `local_rank`.

    gpu_nodes[i] = NUMA node of GPU i (-1 unknown), one entry per local rank;
    allowed      = CPUs this job may use (os.sched_getaffinity);
    node_cpus    = {node: [cpus]};
    cores_of     = function grouping a CPU list into physical cores (default: one CPU each).
    Ranks whose GPUs sit on the same node split that node's allowed cores into equal
    contiguous shares, SMT siblings staying together.  With no topology information every
    rank gets an equal share of `allowed`.  Never returns an empty list."""
    allowed = sorted(allowed)
    node = gpu_nodes[local_rank]
    mates = [r for r, nd in enumerate(gpu_nodes) if nd == node]
    pool = [c for c in node_cpus.get(node, []) if c in set(allowed)] if node >= 0 else []
    if not pool:                                  # unknown topology: even split of everything
        pool, mates = allowed, list(range(len(gpu_nodes)))
    cores = cores_of(pool) if cores_of else [[c] for c in pool]
    k, m = mates.index(local_rank), len(mates)
    mine = cores[k * len(cores) // m:(k + 1) * len(cores) // m] or [cores[k % len(cores)]]
    return sorted(c for core in mine for c in core)


def _dir_quota(d):
    """quota / period of one cgroup directory (v2 cpu.max, v1 cpu.cfs_*), None = no limit there."""
    try:
        with open(d + "/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open(d + "/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open(d + "/cpu.cfs_period_us") as f:
            period = float(f.read())
        return None if quota <= 0 or period <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_quota(cgroup="/sys/fs/cgroup", proc="/proc/self/cgroup"):
    """CPUs' worth of run time the process may use per unit of wall time, or None when
    unlimited: the tightest limit of its own cgroup (resolved from /proc/self/cgroup, v2
    "0::/path" or v1 "N:cpu,cpuacct:/path") and of every ancestor up to the mount point —
    the same walk csrc/layout.c:jga_cpu_budget() does, so bench.py and the library agree.  A
    box can show 256 CPUs and grant 16: threads beyond the grant only get the whole group
    throttled, so thread counts have to be sized by this, not by the CPU count."""
    rel = {"": "", "cpu": ""}
    try:
        with open(proc) as f:
            for line in f:
                parts = line.rstrip("\n").split(":", 2)
                if len(parts) != 3:
                    continue
                if parts[1] == "":
                    rel[""] = parts[2]
                elif "cpu" in parts[1].split(",") or "cpuacct" in parts[1].split(","):
                    rel["cpu"] = parts[2]
    except OSError:
        pass
    best = None
    for root, r in ((cgroup, rel[""]), (cgroup + "/cpu", rel["cpu"])):
        d = root + (r if r != "/" else "")
        while True:
            q = _dir_quota(d)
            if q is not None and (best is None or q < best):
                best = q
            if len(d) <= len(root):
                break
            d = d.rsplit("/", 1)[0]
    return best


def rank_cpu_budget(ncpus, world, quota="auto"):
    """How many CPUs' worth of work one of `world` ranks can count on: its share of the
    affinity mask, cut down to its share of the cgroup grant."""
    q = cpu_quota() if quota == "auto" else quota
    return max(1, min(int(ncpus), int(q / world + 0.5))) if q else max(1, int(ncpus))


def pin_rank_to_gpu_node(local_rank, nlocal, pci_bus_ids=None):
    """Restrict the calling thread (and every thread it creates from now on) to this rank's
    share of the host CPUs.  Returns a description for the bench line."""
    import os
    allowed = os.sched_getaffinity(0)
    ids = list(pci_bus_ids or [])
    gpu_nodes = [device_numa_node(ids[i]) if i < len(ids) else -1 for i in range(nlocal)]
    cpus = {nd: numa_node_cpus(nd) for nd in set(gpu_nodes) if nd >= 0}
    share = rank_cpu_share(local_rank, gpu_nodes, allowed, cpus, cpu_cores)
    os.sched_setaffinity(0, share)
    return {"numa_node": gpu_nodes[local_rank], "cpus": len(share),
            "cpu_list": "%d-%d%s" % (share[0], share[-1], "" if share[-1] - share[0] + 1 == len(share)
                                      else " (%d of them)" % len(share))}
