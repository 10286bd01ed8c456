#!/usr/bin/env python3
"""One configuration of the streaming expand+depthwise kernel, a few launches (for rocprofv3 passes).
usage: xds_one.py B Cin Cexp rate parts [iters]   (AMS_XDS_FORCE selects the geometry)"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

B, Cin, Cexp, rate, parts = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
PRE = int(sys.argv[7]) if len(sys.argv) > 7 else 1
lib = hip.lib()
dev = "cuda:0"
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
H, W = 33, 65
x = torch.randn(B, H, W, Cin, device=dev)
we = torch.randn(Cin, Cexp, device=dev) / Cin ** 0.5
wd = torch.randn(3, 3, Cexp, 1, device=dev) * 0.4
se, sd = torch.rand(Cexp, device=dev) + 0.5, torch.rand(Cexp, device=dev) + 0.5
he, hd = torch.randn(Cexp, device=dev), torch.randn(Cexp, device=dev)
y = torch.empty(B, H, W, Cexp, device=dev)
panels = torch.zeros(3 * Cexp * Cin + 3 * B * H * W * Cin, dtype=torch.int16, device=dev)
for _ in range(iters):
    hip.check(lib.ams_k_expand_dw_stream(P(x), B, H, W, Cin, P(we), P(se), P(he), Cexp, P(wd), rate, P(sd), P(hd), P(y), P(panels),
                                         panels.numel(), parts, PRE, st))
torch.cuda.synchronize()
print("done")
