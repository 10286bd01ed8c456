#!/usr/bin/env python3
"""Probe: the 8-frame fine-tune step captured as ONE hipGraph and replayed (the learning-rate argument is frozen into the capture, so this is a
timing experiment only).  Round 2 on MI355X: eager 11.19 ms per step, graph replay 11.64 ms — the step is not bound by launch gaps."""
import sys, time
sys.path.insert(0, ".")
import torch
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from bench import CI
B, H = 8, 512
W0 = Wt.synthetic_weights(S.build_spec(), 0)
fr, lb = synth.SyntheticVideo(H, B, CI).clip()
eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
eng.load_variables(W0)
f, l = torch.from_numpy(fr).cuda(), torch.from_numpy(lb).cuda()
for _ in range(5):
    eng.train_step(f, l, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.train_step(f, l, 1e-3)
torch.cuda.synchronize()
print("eager  %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            eng.train_step(f, l, 1e-3)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            eng.train_step(f, l, 1e-3)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph  %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
