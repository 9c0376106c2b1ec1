"""What the host-to-device link sustains from pinned memory (torch, HIP events): one stream and
several streams in flight, 96 MB per copy (a group of 32 bench files).  Usage: python tools/h2d_probe.py"""
import torch
n = 96 << 20
srcs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(8)]
dsts = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(8)]
for k in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(k)]
    for rep in range(2):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(4):
            for i, s in enumerate(streams):
                s.wait_event(e0) if r == 0 else None
                with torch.cuda.stream(s):
                    dsts[i].copy_(srcs[i], non_blocking=True)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%d stream(s): %.1f GB/s host -> device" % (k, 4 * k * n / ms / 1e6))
d2h = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(4):
    d2h.copy_(dsts[0], non_blocking=True)
e1.record(); torch.cuda.synchronize()
print("device -> host: %.1f GB/s" % (4 * n / e0.elapsed_time(e1) / 1e6))
