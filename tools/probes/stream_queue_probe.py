#!/usr/bin/env python3
"""How many latency-bound launch chains overlap?  n streams each run a chain of `L` tiny dependent kernels (a one-frame forward looks
like this: ~45 dependent launches of a few dozen blocks).  Perfect overlap = flat time as n grows.  Run with GPU_MAX_HW_QUEUES=k in the
environment to see the effect of the runtime's stream -> hardware-queue multiplexing.  usage: stream_queue_probe.py [L]"""
import os
import sys
import time
import torch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
x = [torch.zeros(N, device=dev) for _ in range(8)]
y = [torch.zeros(N, device=dev) for _ in range(8)]
streams = [torch.cuda.Stream(dev) for _ in range(8)]


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(L):
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                torch.cumsum(x[i], 0, out=y[i])          # a long scan: tens of microseconds on a handful of blocks (host launch cost ~10 us)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


run(1)
base = min(run(1) for _ in range(3))
print("GPU_MAX_HW_QUEUES=%s  L=%d  1 stream: %.3f ms (%.2f us per launch)" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), L, base, 1e3 * base / L))
for n in (2, 3, 4, 5, 6, 8):
    t = min(run(n) for _ in range(3))
    print("  %d streams: %.3f ms  = %.2f x one stream (1.0 = perfect overlap, %d = serial)" % (n, t, t / base, n))
