#!/usr/bin/env python3
"""Host time to ENQUEUE one fine-tune step / one single-frame inference (the call returns before the GPU finishes): if it is close to the GPU
time of the step, the step is bound by the host's launch rate.  usage: host_enqueue.py [B] [H]"""
import sys
import time
sys.path.insert(0, ".")
import torch
from ams_amd import spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from bench import CI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W0 = Wt.synthetic_weights(S.build_spec(), 0)
fr, lb = synth.SyntheticVideo(H, B, CI).clip()
eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
eng.load_variables(W0)
f, l = torch.from_numpy(fr).cuda(), torch.from_numpy(lb).cuda()
for _ in range(3):
    eng.train_step(f, l, 1e-3)
torch.cuda.synchronize()
hs, gs = [], []
for _ in range(5):
    t0 = time.perf_counter()
    eng.train_step(f, l, 1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(1e3 * (t1 - t0)); gs.append(1e3 * (t2 - t0))
print("train step: host enqueue %.3f ms (min %.3f), until GPU done %.3f ms" % (sorted(hs)[2], min(hs), sorted(gs)[2]))
eng.freeze()
one = f[:1].contiguous()
for _ in range(5):
    eng.predict(one, 0)
torch.cuda.synchronize()
hs, gs = [], []
for _ in range(20):
    t0 = time.perf_counter()
    eng.predict(one, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(1e3 * (t1 - t0)); gs.append(1e3 * (t2 - t0))
print("predict(1 frame): host enqueue %.3f ms (min %.3f), until GPU done %.3f ms" % (sorted(hs)[10], min(hs), sorted(gs)[10]))
