#!/usr/bin/env python3
"""One shape of the whole-block kernel (k_block.hip), a few launches: timing with HIP events, or the body of a rocprofv3 pass.
usage: block_one.py B H W Cin Cexp Cout stride res [iters]     (AMS_BLK_TILE=THxTW selects the tile; BLK_F16=1: the fp16-pair form of the
default plan; with AMS_BLK_TIMED=1 the kernel's per-phase cycle sums are printed)"""
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
F16 = os.environ.get("BLK_F16", "0") == "1"
panels = torch.zeros(3 * Cexp * 32, dtype=torch.int16, device=dev)
hpan = torch.zeros(2 * Cexp * 32 + 2 * Cout * ((Cexp + 31) // 32 * 32), dtype=torch.int16, device=dev)


def run():
    if F16:
        hip.check(lib.ams_k_block_fused_f16(P(x), B, H, W, Cin, P(we), P(se), P(he), Cexp, P(wd), stride, P(sd), P(hd), P(wp), Cout, P(sp), P(hp), res, P(y),
                                            P(hpan), hpan.numel(), st))
        return
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
if os.environ.get("AMS_BLK_TIMED"):
    out = (C.c_uint64 * 8)()
    hip.check(lib.ams_debug_phase_cycles(1, out, 8))
    run()
    torch.cuda.synchronize()
    hip.check(lib.ams_debug_phase_cycles(1, out, 8))
    tot = sum(out[:4])
    print("  %d waves, %.0f cycles per wave:" % (out[6], tot / out[6]), ", ".join("%s %.0f (%.0f %%)" % (n, out[i] / out[6], 100.0 * out[i] / tot)
                                                                                for i, n in enumerate(["prologue", "expand", "depthwise+project", "epilogue"])))
