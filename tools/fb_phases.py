#!/usr/bin/env python3
"""Per-phase cycle sums of the walking first block (AMS_FB_ABL=32|bits in the environment): usage fb_phases.py [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
os.environ.setdefault("AMS_FB_ABL", "32")
_MEASURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ams_amd", "libams_hip_measure.so")
os.environ.setdefault("AMS_HIP_LIB", _MEASURE)      # the clocked / ablated kernels are in the measurement build only: make -C ams_amd/csrc measure
assert os.path.exists(os.environ["AMS_HIP_LIB"]), "build it first: make -C ams_amd/csrc measure"
import torch
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from bench import CI

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fr, _ = synth.SyntheticVideo(512, B, CI).clip()
eng = StudentEngine(CI, 512, 1024, max_batch=B, trainable=False)
eng.load_variables(Wt.synthetic_weights(S.build_spec(), 0))
eng.freeze()
eng.set_dual_stream(0)
f = torch.from_numpy(fr).cuda()
for _ in range(3):
    eng.predict(f)
torch.cuda.synchronize()
out = (C.c_uint64 * 8)()
hip.check(hip.lib().ams_debug_phase_cycles(0, out, 8))
n = 5
for _ in range(n):
    eng.predict(f)
torch.cuda.synchronize()
hip.check(hip.lib().ams_debug_phase_cycles(0, out, 8))
v = [x / n for x in out]
tiles = v[6]
names = ["tile decode", "stem", "barrier", "depthwise + project"]
tot = sum(v[:4])
print("AMS_FB_ABL=%s  wave-tiles per launch %.0f, cycles per wave-tile %.0f" % (os.environ["AMS_FB_ABL"], tiles, tot / tiles))
for nm, x in zip(names, v[:4]):
    print("  %-20s %8.0f cycles per wave-tile  %5.1f %%" % (nm, x / tiles, 100 * x / tot))
