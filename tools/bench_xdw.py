#!/usr/bin/env python3
"""Time the early blocks' recompute kernels of the fine-tune step alone (k_xdw_train.hip): backward reduce and dx passes at the step's shapes.
usage: bench_xdw.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

lib = hip.lib()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
# (H, W, Cin, Cexp, stride) of blocks 1..6 at 512 x 1024
SHAPES = [(257, 513, 16, 96, 2), (129, 257, 24, 144, 1), (129, 257, 24, 144, 2), (65, 129, 32, 192, 1), (65, 129, 32, 192, 1), (65, 129, 32, 192, 2)]
for (H, W, Cin, Cexp, s) in SHAPES:
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    x = torch.randn(B, H, W, Cin, device=dev)
    we = torch.randn(Cin, Cexp, device=dev) / Cin ** 0.5
    wd = torch.randn(9, Cexp, device=dev)
    dz = torch.randn(B, Ho, Wo, Cexp, device=dev)
    vec = [torch.rand(Cexp, device=dev) + 0.5 for _ in range(7)]
    res = torch.randn(B, H, W, Cin, device=dev)
    dx = torch.empty_like(x)
    n_scr = lib.ams_k_xdw_train_scratch(B, H, W, Cin, Cexp)
    scr = torch.empty(n_scr, device=dev)
    rows, stride_out = C.c_int32(0), C.c_int64(0)

    def red():
        hip.check(lib.ams_k_xdw_bwd_reduce(P(x), B, H, W, Cin, P(we), Cexp, P(vec[0]), P(vec[1]), P(vec[2]), P(vec[3]), hip.ACT_RELU6, P(wd), s, P(dz),
                                           P(scr), n_scr, C.byref(rows), C.byref(stride_out), st))

    def dxp():
        hip.check(lib.ams_k_xdw_bwd_dx(P(x), B, H, W, Cin, P(we), Cexp, P(vec[0]), P(vec[1]), hip.ACT_RELU6, P(wd), s, P(dz), P(vec[4]), P(vec[5]), P(vec[6]),
                                       P(res), P(dx), st))

    for name, fn in (("reduce", red), ("dx", dxp)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        nbytes = 4.0 * (dz.numel() + x.numel() * (3 if name == "dx" else 1))
        print("B=%d %dx%d %d->%d s%d %s: %.1f us  %.0f GB/s (algorithmic)" % (B, H, W, Cin, Cexp, s, name, us, nbytes / us / 1e3))
