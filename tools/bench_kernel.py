#!/usr/bin/env python3
"""Time one pointwise GEMM shape through the C ABI (tuning aid).  usage: bench_kernel.py M K N [split|split3|f16|f16p]
(f16: two fp16 parts, operand split in the kernel; f16p: operand pre-packed as fp16 pairs)"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4])
split = len(sys.argv) > 4
split3 = split and sys.argv[4] == "split3"
f16 = split and sys.argv[4] in ("f16", "f16p")
f16p = split and sys.argv[4] == "f16p"
lib = hip.lib()
dev = "cuda:0"
x = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) / K ** 0.5
sc = torch.rand(N, device=dev) + 0.5
sh = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
Kp = (K + 31) // 32 * 32
panels = torch.zeros(3 * N * Kp, dtype=torch.int16, device=dev)


def run():
    if f16:
        hip.check(lib.ams_k_pointwise_split_f16(P(x), M, K, P(w), N, P(sc), P(sh), hip.ACT_RELU6, None, P(y), P(panels), panels.numel(), P(x) if f16p else None,
                                                None, st))
    elif split3:
        hip.check(lib.ams_k_pointwise_split3(P(x), M, K, P(w), N, P(sc), P(sh), hip.ACT_RELU6, None, P(y), P(panels), panels.numel(), st))
    elif split:
        hip.check(lib.ams_k_pointwise_split(P(x), M, K, P(w), N, P(sc), P(sh), hip.ACT_RELU6, None, P(y), P(panels), panels.numel(), st))
    else:
        hip.check(lib.ams_k_pointwise(P(x), M, K, P(w), N, 0, None, 1, P(sc), P(sh), hip.ACT_RELU6, None, P(y), st))


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
nbytes = 4.0 * (M * (K + N) + K * N)
print("M=%d K=%d N=%d %s: %.1f us  %.0f GB/s  %.1f TFLOP/s" % (M, K, N, (sys.argv[4] if f16 else "split3" if split3 else "split") if split else "f32", us, nbytes / us / 1e3, 2.0 * M * K * N / us / 1e6))
