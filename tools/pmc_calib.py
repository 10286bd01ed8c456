#!/usr/bin/env python3
"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters: kernels of KNOWN byte counts, far larger than L2 (32 MB)
and the Infinity Cache (256 MB), in the access patterns of this library.  Run under `rocprofv3 --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` (tools/pmc_calib.sh); the expected bytes are printed so the ratios can be read off.
"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from ams_amd import hip  # noqa: E402

lib = hip.lib()
dev = "cuda:0"
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

# 1. depthwise 3x3, stride 1: reads [B,H,W,C] once (+ halo rows from L2), writes the same shape      (float4 per lane)
B, H, W, Cc = 16, 256, 512, 96
x = torch.randn(B, H, W, Cc, device=dev)
y = torch.empty_like(x)
wd = torch.randn(9, Cc, device=dev)
sc = torch.rand(Cc, device=dev) + 0.5
sh = torch.randn(Cc, device=dev)
for _ in range(2):
    hip.check(lib.ams_k_depthwise3x3(P(x), B, H, W, Cc, P(wd), 1, 1, P(sc), P(sh), hip.ACT_RELU6, P(y), st))
torch.cuda.synchronize()
print("dw3x3_fwd: read %.1f KB write %.1f KB" % (x.numel() * 4 / 1024, y.numel() * 4 / 1024))

# 2. streaming pointwise GEMM (persistent S variant): reads [M,K], writes [M,N]
M, K, N = 4 * 1024 * 1024, 32, 96
a = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) / K ** 0.5
o = torch.empty(M, N, device=dev)
scn = torch.rand(N, device=dev) + 0.5
shn = torch.randn(N, device=dev)
for _ in range(2):
    hip.check(lib.ams_k_pointwise(P(a), M, K, P(w), N, 0, None, 1, P(scn), P(shn), hip.ACT_RELU6, None, P(o), st))
torch.cuda.synchronize()
print("pw_gemm_f32_s: read %.1f KB write %.1f KB" % (a.numel() * 4 / 1024, o.numel() * 4 / 1024))

# 3. torch copy and reduction of 1 GiB (16 B per lane)
big = torch.randn(256 * 1024 * 1024, device=dev)
dst = torch.empty_like(big)
dst.copy_(big)
s = big.sum()
torch.cuda.synchronize()
print("torch copy: read %.1f KB write %.1f KB; torch sum: read %.1f KB" % (big.numel() * 4 / 1024, big.numel() * 4 / 1024, big.numel() * 4 / 1024))
