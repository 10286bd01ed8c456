#!/usr/bin/env python3
"""One shape of the whole-block kernel (k_block.hip), a few launches: timing with HIP events, or the body of a rocprofv3 pass.
usage: block_one.py B H W Cin Cexp Cout stride res [iters]     (AMS_BLK_TILE=THxTW selects the tile)"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

B, H, W, Cin, Cexp, Cout, stride, res = (int(v) for v in sys.argv[1:9])
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 10
lib = hip.lib()
dev = "cuda:0"
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(B, H, W, Cin, device=dev)
we = torch.randn(Cin, Cexp, device=dev) / Cin ** 0.5
wd = torch.randn(3, 3, Cexp, 1, device=dev) * 0.4
wp = torch.randn(Cexp, Cout, device=dev) / Cexp ** 0.5
se, sd, sp = (torch.rand(n, device=dev) + 0.5 for n in (Cexp, Cexp, Cout))
he, hd, hp = (torch.randn(n, device=dev) for n in (Cexp, Cexp, Cout))
Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
y = torch.empty(B, Ho, Wo, Cout, device=dev)
import os
X6 = os.environ.get("BLK_X6", "1") == "1"
panels = torch.zeros(3 * Cexp * 32, dtype=torch.int16, device=dev)


def run():
    hip.check(lib.ams_k_block_fused(P(x), B, H, W, Cin, P(we), P(se), P(he), Cexp, P(wd), stride, P(sd), P(hd), P(wp), Cout, P(sp), P(hp), res, P(y),
                                    P(panels) if X6 else None, panels.numel(), st))


for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
print("block %dx%d %d->%d->%d s%d B=%d: %.1f us" % (H, W, Cin, Cexp, Cout, stride, B, e0.elapsed_time(e1) * 1e3 / iters))
