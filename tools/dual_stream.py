#!/usr/bin/env python3
"""Two engines on two streams of one GPU, half the batch each, against one engine with the whole batch (experiment: do the
latency-bound kernels of one stream fill the gaps of the other?).  usage: dual_stream.py [B_total]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ams_amd import spec as S, synth, weights as Wt  # noqa: E402
from ams_amd.engine import StudentEngine  # noqa: E402

CI = [0, 1, 2, 10, 11, 13]
H = 512
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
frames, _ = synth.SyntheticVideo(H, B, CI, seed=1).clip()
dev = "cuda:0"
fr = torch.from_numpy(frames).to(dev)


def rate(engines, streams, parts, steps=10):
    for _ in range(3):
        for e, st, p in zip(engines, streams, parts):
            with torch.cuda.stream(st):
                e.predict(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, st, p in zip(engines, streams, parts):
            with torch.cuda.stream(st):
                e.predict(p)
    torch.cuda.synchronize()
    return steps * sum(p.shape[0] for p in parts) / (time.perf_counter() - t0)


one = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False, device=dev)
one.load_variables(W0); one.freeze()
print("one engine, %d frames per step: %.0f frames/s" % (B, rate([one], [torch.cuda.current_stream()], [fr])))
one.close(); del one
torch.cuda.empty_cache()
for n in (2, 4):
    engs = []
    for _ in range(n):
        e = StudentEngine(CI, H, 2 * H, max_batch=B // n, trainable=False, device=dev)
        e.load_variables(W0); e.freeze()
        engs.append(e)
    sts = [torch.cuda.Stream() for _ in range(n)]
    parts = [fr[i * (B // n):(i + 1) * (B // n)].contiguous() for i in range(n)]
    print("%d engines on %d streams, %d frames each: %.0f frames/s" % (n, n, B // n, rate(engs, sts, parts)))
    for e in engs:
        e.close()
    del engs
    torch.cuda.empty_cache()
