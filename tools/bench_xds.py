#!/usr/bin/env python3
"""Streaming expand+depthwise kernel (k_xdw_stream.hip) against the two kernels it replaces, on the stride-16 block shapes of
the 512x1024 student.  usage: bench_xds.py [B ...]   (tuning aid; AMS_XDS_FORCE = "tiles,row segments,column strips")"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

lib = hip.lib()
dev = "cuda:0"
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ.get("XDS_SHAPES", "33,65,64,384,1;33,65,96,576,1;33,65,160,960,2").split(";")]
batches = [int(v) for v in sys.argv[1:]] or [32, 8, 1]
PRE = int(os.environ.get("XDS_PRE", "1"))
forces = os.environ.get("XDS_FORCES", "auto;2,1,1;2,2,1;2,4,1;4,1,1;4,2,1;4,4,1").split(";")


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in batches:
    for H, W, Cin, Cexp, rate in SHAPES:
        M = B * H * W
        x = torch.randn(B, H, W, Cin, device=dev)
        we = torch.randn(Cin, Cexp, device=dev) / Cin ** 0.5
        wd = torch.randn(3, 3, Cexp, 1, device=dev) * 0.4
        se, sd = torch.rand(Cexp, device=dev) + 0.5, torch.rand(Cexp, device=dev) + 0.5
        he, hd = torch.randn(Cexp, device=dev), torch.randn(Cexp, device=dev)
        e = torch.empty(M, Cexp, device=dev)
        y0 = torch.empty(B, H, W, Cexp, device=dev)
        y1 = torch.empty(B, H, W, Cexp, device=dev)
        panels = torch.zeros(3 * Cexp * Cin + 3 * M * Cin, dtype=torch.int16, device=dev)
        for parts in ((3, 2) if Cin >= 64 else (3,)):
            gemm = lib.ams_k_pointwise_split3 if parts == 3 else lib.ams_k_pointwise_split

            def unfused():
                if Cin <= 32:
                    hip.check(lib.ams_k_pointwise(P(x), M, Cin, P(we), Cexp, 0, None, 1, P(se), P(he), hip.ACT_RELU6, None, P(e), st))
                else:
                    hip.check(gemm(P(x), M, Cin, P(we), Cexp, P(se), P(he), hip.ACT_RELU6, None, P(e), P(panels), panels.numel(), st))
                hip.check(lib.ams_k_depthwise3x3(P(e), B, H, W, Cexp, P(wd), 1, rate, P(sd), P(hd), hip.ACT_RELU6, P(y0), st))

            def fused():
                hip.check(lib.ams_k_expand_dw_stream(P(x), B, H, W, Cin, P(we), P(se), P(he), Cexp, P(wd), rate, P(sd), P(hd), P(y1),
                                                     P(panels), panels.numel(), parts, PRE, st))

            t0 = timeit(unfused)
            out_mb = M * Cexp * 4 / 1e6
            line = "B=%2d %dx%d %3d->%3d r%d parts=%d  unfused %6.1f us |" % (B, H, W, Cin, Cexp, rate, parts, t0)
            for f in forces:
                if f == "auto":
                    os.environ.pop("AMS_XDS_FORCE", None)
                else:
                    os.environ["AMS_XDS_FORCE"] = f
                try:
                    t1 = timeit(fused)
                except hip.AmsHipError:
                    line += " %s: n/a |" % f
                    continue
                same = bool(torch.equal(y0, y1))
                line += " %s: %6.1f us (%.2f TB/s out)%s |" % (f, t1, out_mb / t1, "" if same else " MISMATCH")
            os.environ.pop("AMS_XDS_FORCE", None)
            for wf in [] if Cin < 64 else os.environ.get("XWR_FORCES", "4,1,1,0;8,1,1,0;4,1,1,32;8,1,1,64;4,2,1,0").split(";"):      # the weight-register form
                os.environ["AMS_XWR_FORCE"] = wf
                keep, PRE = PRE, 2
                try:
                    t1 = timeit(fused)
                    line += " wreg %s: %6.1f us%s |" % (wf, t1, "" if bool(torch.equal(y0, y1)) else " MISMATCH")
                except hip.AmsHipError as e:
                    line += " wreg %s: n/a |" % wf
                PRE = keep
            os.environ.pop("AMS_XWR_FORCE", None)
            print(line, flush=True)
