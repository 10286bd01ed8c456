"""Not a test (not collected): how far one f32 evaluation of the fine-tune gradient lies from the f64 oracle, per tensor, over several
seeds — the HIP step in its fused and its layer-by-layer forms next to the f32 CPU oracle.  Run on the GPU box:
    python3 tests/grad_noise_study.py [n_seeds] [H]
Prints, per seed, the worst and the median relative L2 error over the trainable tensors for each evaluation."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ams_amd import spec as S, synth, weights as Wt  # noqa: E402
from ams_amd.engine import StudentEngine  # noqa: E402
from oracle.student_torch import StudentOracle  # noqa: E402

CI = [0, 1, 2, 10, 11, 13]


def errors(g, grads_o, spec):
    gnorm = max(float(gv.abs().max()) for gv in grads_o.values())
    out = []
    for v in spec.trainable:
        want = grads_o[v.name].numpy().reshape(-1)
        floor = max(np.linalg.norm(want), 1e-3 * gnorm * np.sqrt(want.size))
        out.append(np.linalg.norm(g[v.offset:v.offset + v.size] - want) / floor)
    return np.array(out)


def hip_grads(W, fr, lb, H, B, fused):
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W)
    recompute, dgrad, red = fused
    eng.set_train_recompute(recompute, fuse_dgrad_bn=dgrad, fuse_gemm_red=red)
    eng.train_step(fr, lb, 1e-3)
    g = eng.grads.cpu().numpy().astype(np.float64)
    spec = eng.spec
    eng.close()
    return g, spec


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = 4
    forms = {"default": (True, 2, 2), "dgrad1": (True, 1, 2), "layerwise": (False, 0, 0), "red3": (True, 2, 3)}
    worst = {k: [] for k in list(forms) + ["f32cpu"]}
    for seed in range(n):
        W = Wt.synthetic_weights(S.build_spec(), seed=seed)
        fr, lb = synth.SyntheticVideo(H, B, CI, seed=seed + 100).clip()
        o = StudentOracle(W, CI, dtype=torch.float64)
        _, grads_o = o.gradients(fr.astype(np.float32), lb)
        o32 = StudentOracle(W, CI, dtype=torch.float32)
        _, g32 = o32.gradients(fr.astype(np.float32), lb)
        line = ["seed %d" % seed]
        spec = None
        for name, f in forms.items():
            g, spec = hip_grads(W, fr, lb, H, B, f)
            e = errors(g, grads_o, spec)
            worst[name].append(e.max())
            line.append("%s max %.2e med %.2e" % (name, e.max(), np.median(e)))
        flat32 = np.concatenate([g32[v.name].numpy().reshape(-1).astype(np.float64) for v in spec.trainable])
        e = errors(flat32, grads_o, spec)
        worst["f32cpu"].append(e.max())
        line.append("f32cpu max %.2e med %.2e" % (e.max(), np.median(e)))
        print(" | ".join(line), flush=True)
    for k, v in worst.items():
        print("%-10s worst-tensor error over seeds: max %.2e  mean %.2e" % (k, max(v), float(np.mean(v))))


if __name__ == "__main__":
    main()
