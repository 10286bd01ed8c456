#!/usr/bin/env python3
"""Time the input-gradient 1x1 GEMMs of the fine-tune step's early blocks with the BN-backward reduction in their epilogue
(ams_k_pointwise_red mode 2, exact-f32 streaming kernel) and, beside them, the same product without the reduction.  usage: bench_red.py [split]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

lib = hip.lib()
dev = "cuda:0"
split = 1 if len(sys.argv) > 1 else 0
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(8 * 257 * 513, 16, 32), (8 * 129 * 257, 24, 96), (8 * 129 * 257, 24, 144), (8 * 65 * 129, 32, 144), (8 * 65 * 129, 32, 192), (8 * 65 * 129, 32, 192),
          (8 * 33 * 65, 64, 192)]
import os
if os.environ.get("BENCH_RED_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["BENCH_RED_SHAPES"].split(";")]
if os.environ.get("BENCH_RED_ONLY"):
    SHAPES = [SHAPES[int(os.environ["BENCH_RED_ONLY"])]]
for M, K, N in SHAPES:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5          # dgrad orientation: w is [N_out_of_forward = K here ...]; trans_w = 1 -> w [N, K]
    z = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    vec = [torch.rand(N, device=dev) + 0.5 for _ in range(4)]
    part = torch.empty((M // 64 + 2048) * 2 * N, device=dev)
    rows = C.c_int32(0)
    Kp = (K + 31) // 32 * 32
    panels = torch.zeros(3 * N * Kp, dtype=torch.int16, device=dev)

    def red():
        hip.check(lib.ams_k_pointwise_red(P(x), M, K, P(w), N, 1, split, 2, None, P(z), P(vec[0]), P(vec[1]), P(vec[2]), P(vec[3]), hip.ACT_RELU6, None,
                                          P(y), P(part), part.numel(), C.byref(rows), P(panels), panels.numel(), st))

    def plain():
        if split:
            hip.check(lib.ams_k_pointwise_split3(P(x), M, K, P(w.t().contiguous()), N, None, None, hip.ACT_NONE, None, P(y), P(panels), panels.numel(), st))
        else:
            hip.check(lib.ams_k_pointwise(P(x), M, K, P(w), N, 1, None, 1, None, None, hip.ACT_NONE, None, P(y), st))

    for name, fn in (("red2", red), ("plain", plain)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        nbytes = 4.0 * M * (K + N * (2 if name == "red2" else 1))
        print("M=%d K=%d N=%d %s: %.1f us  %.0f GB/s  rows=%d" % (M, K, N, name, us, nbytes / us / 1e3, rows.value))
