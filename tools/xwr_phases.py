#!/usr/bin/env python3
"""Per-role cycle sums of the weight-register streaming kernel inside the 32-frame step (AMS_XWR_TIMED=1 is set here): usage xwr_phases.py [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
os.environ["AMS_XWR_TIMED"] = "1"
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
hip.check(hip.lib().ams_debug_phase_cycles(2, out, 8))
hip.check(hip.lib().ams_debug_phase_cycles(3, out, 8))
n = 5
for _ in range(n):
    eng.predict(f)
torch.cuda.synchronize()
hip.check(hip.lib().ams_debug_phase_cycles(2, out, 8))
def show(name, out, taps):
    es, ds = max(out[6], 1), max(out[7], 1)
    print("%s  E-waves: %d wave-steps per pass, %.0f cycles between barriers + %.0f at the barrier per step" % (name, es / n, out[0] / es, out[1] / es))
    if taps:
        print("%s  D-waves: %d wave-steps per pass, %.0f cycles until the taps have landed + %.0f FMAs + %.0f epilogue and stores + %.0f at the barrier per step" % (name, ds / n, out[4] / ds, out[5] / ds, out[2] / ds, out[3] / ds))
    else:
        print("%s  D-waves: %d wave-steps per pass, %.0f cycles between barriers + %.0f at the barrier per step" % (name, ds / n, out[2] / ds, out[3] / ds))


show("xdw_wreg  ", out, True)
out3 = (C.c_uint64 * 8)()
hip.check(hip.lib().ams_debug_phase_cycles(3, out3, 8))
show("xdw_stream", out3, False)
sys.exit(0)
es, ds = out[6], out[7]
print("E-waves: %d wave-steps per pass, %.0f cycles between barriers + %.0f at the barrier per step" % (es / n, out[0] / es, out[1] / es))
print("D-waves: %d wave-steps per pass, %.0f cycles until the taps have landed + %.0f arithmetic and stores + %.0f at the barrier per step" % (ds / n, out[4] / ds, out[2] / ds, out[3] / ds))
