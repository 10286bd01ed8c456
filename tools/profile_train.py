#!/usr/bin/env python3
"""Per-kernel profile of one fine-tune step (engine's HIP-event profiler). usage: profile_train.py [B] [H]"""
import sys
from collections import defaultdict
sys.path.insert(0, ".")
import torch
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
sys.path.insert(0, ".")
from bench import read_profile, CI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W0 = Wt.synthetic_weights(S.build_spec(), 0)
fr, lb = synth.SyntheticVideo(H, B, CI).clip()
eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
eng.load_variables(W0)
f, l = torch.from_numpy(fr).cuda(), torch.from_numpy(lb).cuda()
for _ in range(2):
    eng.train_step(f, l, 1e-3)
hip.check(eng.lib.ams_student_profile(eng._h, 1))
eng.train_step(f, l, 1e-3)
rows = read_profile(eng)
hip.check(eng.lib.ams_student_profile(eng._h, 0))
agg = defaultdict(lambda: [0, 0.0, 0.0])
for name, layer, ms, nb, _fl, _fx in rows:
    a = agg[name.split("<")[0]]
    a[0] += 1; a[1] += ms; a[2] += nb
tot = sum(a[1] for a in agg.values())
print("train step kernels: %.2f ms over %d launches" % (tot, len(rows)))
for k, (n, ms, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %4d launches %8.3f ms %5.1f%%  %7.0f MB %6.0f GB/s" % (k, n, ms, 100 * ms / tot, nb / 1e6, nb / ms / 1e6))
if len(sys.argv) > 3:
    for name, layer, ms, nb, _fl, _fx in rows:
        if sys.argv[3] in name:
            print("%3d %-30s %8.1f us %7.0f GB/s %8.0f KB" % (layer, name, 1e3 * ms, nb / ms / 1e6, nb / 1e3))
