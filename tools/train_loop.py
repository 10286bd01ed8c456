#!/usr/bin/env python3
"""Plain fine-tune loop for rocprofv3 (no engine-side profiler): usage train_loop.py [B] [H] [steps] [warmup].
Prints the HIP-event time per step; under `rocprofv3 --kernel-trace` tools/trace_timeline.py turns the CSV into a per-launch timeline."""
import sys
import time
sys.path.insert(0, ".")
import torch
from ams_amd import spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from bench import CI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 3
W0 = Wt.synthetic_weights(S.build_spec(), 0)
fr, lb = synth.SyntheticVideo(H, B, CI).clip()
eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
eng.load_variables(W0)
f, l = torch.from_numpy(fr).cuda(), torch.from_numpy(lb).cuda()
import os
if os.environ.get("AMS_LOOP_OWN_STREAM"):        # a stream of its own instead of the null stream (a CU-masked side stream is a blocking stream)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.Stream())
for _ in range(warm):
    eng.train_step(f, l, 1e-3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(steps):
    loss = eng.train_step(f, l, 1e-3)
e1.record()
torch.cuda.synchronize()
print("train step: %.3f ms (events), %.3f ms (host), loss %s" % (e0.elapsed_time(e1) / steps, 1e3 * (time.perf_counter() - t0) / steps,
                                                                 loss.cpu().numpy().tolist()))
