#!/usr/bin/env python3
"""bf16 ACTIVATION STORAGE of the output-stride-16 section (the depthwise results `d` and the block inputs), judged on
"trained-like" weights instead of the random synthetic initialisation.

1. Trained-like weights: the synthetic seed-0 student fine-tuned on the synthetic video for N steps of 8 frames (Adam, lr 1e-3, the schedule
   of the stream leg): BN statistics and weights settle on the data the way a checkpoint's would (there is no real checkpoint here).
2. Frozen inference of 8 fresh 512 x 1024 frames twice: f32 storage (default) and with AMS_OPT_EMULATE_BF16_STORAGE = 1, which rounds `d`
   and the block inputs of blocks 7-16 to bf16 right after they are written (the values bf16 storage would hold; arithmetic unchanged).
3. Reports label mismatch, low-resolution logits deviation, mIoU vs the procedural teacher for both, for the random AND the trained-like
   weights.  usage: bf16_storage_study.py [steps] [out.json]"""
import json
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from ams_amd.utils import calculate_miou
from bench import CI

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r03_bf16_storage_study.json"
H, B = 512, 8
W0 = Wt.synthetic_weights(S.build_spec(), 0)
video = synth.SyntheticVideo(H, 64, CI, seed=21)
vf, vl = video.clip()
test_f, test_l = synth.SyntheticVideo(H, B, CI, seed=22).clip()


def miou(cm):
    return float(np.nanmean(calculate_miou(cm.astype(np.float64), nan=True)))


def evaluate(variables, tag):
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(variables)
    eng.freeze()
    res = {}
    h, w = eng.lowres
    ref = None
    for mode in (0, 1):
        hip.check(eng.lib.ams_student_set_option(eng._h, hip.OPT_EMULATE_BF16_STORAGE, mode))
        lab, conf, loss = eng.predict_with_metric(test_f, test_l)
        low = eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].clone()
        ls = loss.cpu().numpy()
        cur = {"miou_vs_teacher": round(miou(conf.cpu().numpy()), 6), "loss": round(float(ls[0] / ls[1]), 6)}
        if mode == 0:
            ref = (lab.clone(), low)
        else:
            cur["label_mismatch_fraction"] = float("%.3e" % (lab != ref[0]).float().mean().item())
            cur["logits_max_rel_dev"] = float("%.3e" % ((low - ref[1]).abs().max() / ref[1].abs().max()).item())
            cur["logits_rms_rel_dev"] = float("%.3e" % ((low - ref[1]).pow(2).mean().sqrt() / ref[1].pow(2).mean().sqrt()).item())
        res["bf16_storage_d_and_block_inputs" if mode else "f32_storage"] = cur
    eng.close()
    print(tag, json.dumps(res))
    return res


result = {"what": __doc__.split("\n\n")[0], "frames": B, "size": "512x1024", "train_steps": steps}
result["random_synthetic_weights"] = evaluate(W0, "random")
srv = StudentEngine(CI, H, 2 * H, max_batch=8, trainable=True)
srv.load_variables(W0)
f, l = torch.from_numpy(vf).cuda(), torch.from_numpy(vl).cuda()
rng = np.random.default_rng(0)
losses = []
for it in range(steps):
    idx = torch.from_numpy(rng.choice(64, 8, replace=False)).cuda()
    ls = srv.train_step(f[idx], l[idx], 1e-3)
    if it % 50 == 0 or it == steps - 1:
        v = ls.cpu().numpy()
        losses.append((it, round(float(v[0] / v[1]), 4)))
trained = srv.get_variables()
srv.close()
result["fine_tune_loss_curve"] = losses
result["trained_like_weights"] = evaluate(trained, "trained-like")
json.dump(result, open(out, "w"), indent=1)
print(json.dumps(result["fine_tune_loss_curve"]))
