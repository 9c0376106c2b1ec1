#!/bin/bash
# What the GPU box's host really offers (CPU quota, topology, memory bandwidth by thread count).
echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|Hypervisor|Virtualization|L3" 
free -g | head -2
cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
python3 - <<'PY'
import numpy as np, threading, time
def bw(n):
    srcs=[np.ones(1<<27,np.uint8) for _ in range(n)]; dsts=[np.empty(1<<27,np.uint8) for _ in range(n)]
    for s,d in zip(srcs,dsts): np.copyto(d,s)
    def body(i):
        for _ in range(4): np.copyto(dsts[i],srcs[i])
    ts=[threading.Thread(target=body,args=(i,)) for i in range(n)]
    t0=time.perf_counter()
    for t in ts:t.start()
    for t in ts:t.join()
    dt=time.perf_counter()-t0
    return n*4*2*(1<<27)/dt/1e9
for n in (1,4,16,32,64,128):
    print("memcpy %3d threads: %.1f GB/s (read+write)"%(n,bw(n)),flush=True)
PY
