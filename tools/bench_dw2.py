#!/usr/bin/env python3
"""Time the tile-form depthwise backward / forward kernels of the fine-tune step (k_dw_train.hip) alone: usage bench_dw2.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

lib = hip.lib()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (H, W, Cn, rate) in [(33, 65, 960, 2), (33, 65, 576, 1), (33, 65, 384, 1)]:
    dy, zd, ze = (torch.randn(B, H, W, Cn, device=dev) for _ in range(3))
    out = torch.empty_like(dy)
    vec = [torch.rand(Cn, device=dev) + 0.5 for _ in range(7)]
    w = torch.randn(9, Cn, device=dev)
    n_scr = lib.ams_k_depthwise3x3_dgrad_bn_apply_scratch(B, H, W, Cn, rate)
    scr = torch.empty(n_scr, device=dev)
    rows = C.c_int32(0)
    n_f = lib.ams_k_depthwise3x3_fwd_bn_tiles_scratch(B, H, W, Cn, rate)
    scr_f = torch.empty(n_f, device=dev)

    def bwd():
        hip.check(lib.ams_k_depthwise3x3_dgrad_bn_apply(P(dy), P(zd), P(vec[0]), P(vec[1]), P(vec[2]), B, H, W, Cn, P(w), rate, P(ze), P(vec[3]), P(vec[4]),
                                                        hip.ACT_RELU6, P(vec[5]), P(vec[6]), P(out), P(scr), n_scr, C.byref(rows), st))

    def fwd():
        hip.check(lib.ams_k_depthwise3x3_fwd_bn_tiles(P(ze), B, H, W, Cn, P(w), rate, P(vec[3]), P(vec[4]), hip.ACT_RELU6, P(vec[5]), P(out), P(scr_f), n_f,
                                                      C.byref(rows), st))

    for name, fn, nt in (("bwd", bwd, 4), ("fwd", fwd, 2)):
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
        print("B=%d %dx%dx%d rate %d %s: %.1f us  %.0f GB/s (algorithmic)  rows=%d" % (B, H, W, Cn, rate, name, us, nt * dy.numel() * 4 / us / 1e3, rows.value))
