#!/usr/bin/env python3
"""Plain frozen-inference loop for rocprofv3: usage infer_loop.py [B] [H] [steps] [warmup] [dual].  Prints ms per call."""
import os
import sys
import time
sys.path.insert(0, ".")
import torch
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from bench import CI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dual = int(sys.argv[5]) if len(sys.argv) > 5 else None
W0 = Wt.synthetic_weights(S.build_spec(), 0)
fr, _ = synth.SyntheticVideo(H, B, CI).clip()
eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
eng.load_variables(W0)
eng.freeze()
if dual is not None:
    eng.set_dual_stream(dual)
f = torch.from_numpy(fr).cuda()
for _ in range(warm):
    out = eng.predict(f)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = eng.predict(f)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
print("infer B=%d: %.3f ms per call, %.1f frames/s, checksum %d" % (B, ms, 1e3 * B / ms, int(out.sum().item())))
