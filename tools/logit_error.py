#!/usr/bin/env python3
"""Low-res logit error of frozen inference at 512x1024 against the f64 oracle, per kernel plan and batch composition.
usage: python tools/logit_error.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from oracle.student_torch import StudentOracle

CI = [0, 1, 2, 10, 11, 13]
H = 512
W0 = Wt.synthetic_weights(S.build_spec(), 0)
frames, labels = synth.SyntheticVideo(H, 8, CI, seed=1).clip()
o64 = StudentOracle(W0, CI, dtype=torch.float64)
o32 = StudentOracle(W0, CI)
with torch.no_grad():
    ref = o64.forward_lowres(frames[:2].astype(np.float32), "frozen").numpy()
    r32 = o32.forward_lowres(frames[:2].astype(np.float32), "frozen").numpy()
rel = lambda a, b: np.abs(a.astype(np.float64) - b).max() / np.abs(b).max()
print("f32 CPU oracle vs f64: %.2e" % rel(r32, ref))
eng = StudentEngine(CI, H, 2 * H, max_batch=8, trainable=False)
eng.load_variables(W0)
eng.freeze()
h, w = eng.lowres
low = lambda B: eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].cpu().numpy()
for name, mode, fb, fx in (("2-part bf16", hip.MATMUL_SPLIT_BF16, True, 1), ("3-part bf16", hip.MATMUL_SPLIT_BF16_X6, True, 1),
                           ("2-part fp16", hip.MATMUL_SPLIT_F16, True, 1),
                           ("exact f32 plan", hip.MATMUL_F32, False, 0)):
    eng.set_matmul_mode(mode); eng.set_fuse_first_block(fb); eng.set_fuse_expand_dw(fx)
    eng.predict(frames)
    e8 = rel(low(2), ref)
    eng.predict(frames[:2])
    e2 = rel(low(2), ref)
    eng.predict(frames[:1])
    e1 = rel(low(1), ref[:1])
    lab = lambda z: z[..., CI].argmax(-1)
    mism = int((lab(low(1)) != lab(ref[:1])).sum())
    print("%-16s vs f64: batch 8 %.2e   batch 2 %.2e   batch 1 %.2e   low-res label mismatches (frame 0) %d" % (name, e8, e2, e1, mism))
