#!/usr/bin/env python3
"""Frozen inference as 1 .. 4 part-batches on as many streams, per batch size (what AMS_OPT_DUAL_STREAM's static rule encodes): usage sweep_parts.py [B ...]"""
import sys
import time
sys.path.insert(0, ".")
import torch
from ams_amd import spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from bench import CI

W0 = Wt.synthetic_weights(S.build_spec(), 0)
for B in [int(v) for v in sys.argv[1:]] or [16, 24, 32, 40, 48, 64]:
    fr, _ = synth.SyntheticVideo(512, B, CI).clip()
    eng = StudentEngine(CI, 512, 1024, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    f = torch.from_numpy(fr).cuda()
    res = []
    for parts in (1, 2, 3, 4):
        if parts == 1:
            eng.set_dual_stream(0)
        else:
            eng.set_dual_stream(2, parts=parts)
        for _ in range(8):
            eng.predict(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            eng.predict(f)
        torch.cuda.synchronize()
        res.append(1e3 * (time.perf_counter() - t0) / n)
    eng.set_dual_stream(1)
    for _ in range(8):
        eng.predict(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.predict(f)
    torch.cuda.synchronize()
    st = 1e3 * (time.perf_counter() - t0) / 30
    print("B=%d  ms per step with 1 / 2 / 3 / 4 parts: %s   static rule: %.3f   best %.0f frames/s" % (B, " / ".join("%.3f" % r for r in res), st, 1e3 * B / min(res)))
    eng.close()
