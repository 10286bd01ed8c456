"""Which kernel owns the head-gradient gap?  (aspp0/weights 2.8e-3 from f64 where an f32 CPU evaluation is 9.5e-5.)

Runs ONE fine-tune step of the smoke configuration (64x128, 2 frames; also 4 frames / other seeds on request) under every switch the
engine has, and prints for each form the max-norm and L2 errors of the gradient tensors vs the f64 oracle beside the f32 CPU oracle's own
errors: the tensor named, and the five worst HIP / f32-CPU ratios among the tensors whose f32-CPU error is below 1e-3 (the well-conditioned
ones, where a ratio far above 1 is a kernel's arithmetic and not the graph's noise).

    python tools/grad_gap_bisect.py [H=64] [B=2] [seed=0] [tensor=aspp0/weights:0]
    python tools/grad_gap_bisect.py layers [H=64] [B=2] [seed=0]
    python tools/grad_gap_bisect.py seeds [H=64] [B=2] [n=8]

``layers``: the layer-by-layer step (every fine-tune fusion off, so that every tensor exists; ams_student_layer_tensor) compared tensor by
tensor with the f64 oracle: raw conv outputs z and activations a on the way forward, gradients with respect to z on the way back (autograd
with respect to the oracle's own z tensors), each beside the f32 CPU oracle's distance from f64 on the same tensor.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from ams_amd import hip, spec, synth, weights  # noqa: E402
from ams_amd.engine import StudentEngine  # noqa: E402
from oracle.student_torch import StudentOracle  # noqa: E402

CI = [0, 1, 2, 10, 11, 13]


def errors(spec_, g, grads_ref):
    """per tensor: (max-norm error / max |want|, L2 error / ||want||)"""
    out = {}
    for v in spec_.trainable:
        want = grads_ref[v.name].numpy().reshape(-1).astype(np.float64)
        got = g[v.name] if isinstance(g, dict) else g[v.offset:v.offset + v.size]
        got = np.asarray(got, dtype=np.float64).reshape(-1)
        mx = np.abs(want).max()
        out[v.name] = (np.abs(got - want).max() / max(mx, 1e-30), np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
    return out


def layer_view(eng, layer, which, B, h, w, c):
    import ctypes as C
    off, n = C.c_size_t(), C.c_size_t()
    hip.check(eng.lib.ams_student_layer_tensor(eng._h, layer, which, C.byref(off), C.byref(n)))
    return eng.arena[off.value:off.value + 4 * B * h * w * c].view(torch.float32).view(B, h, w, c).cpu().numpy().astype(np.float64)


def layers_main():
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sp = spec.build_spec()
    W0 = weights.synthetic_weights(sp, seed=seed)
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=7 + seed).clip()
    fr32 = frames.astype(np.float32)
    ref = {}
    for dt in (torch.float64, torch.float32):
        o = StudentOracle(W0, CI, dtype=dt)
        taps, ztaps = {}, {}
        params = dict(o.vars)
        for v in sp.trainable:
            params[v.name] = o.vars[v.name].clone().requires_grad_(True)
        z = o.reduced_logits(o.logits_full(fr32, "train", params, taps, ztaps))
        target, weight = o.label_targets(labels)
        loss = o.loss_from_reduced(z, target, weight)
        names = list(ztaps)
        gz = torch.autograd.grad(loss, [ztaps[k] for k in names])
        ref[dt] = ({k: v.detach().numpy().astype(np.float64) for k, v in taps.items()},
                   {k: v.detach().permute(0, 2, 3, 1).numpy().astype(np.float64) for k, v in ztaps.items()},
                   {k: g.permute(0, 2, 3, 1).numpy().astype(np.float64) for k, g in zip(names, gz)})
    t64, z64, g64 = ref[torch.float64]
    t32, z32, g32 = ref[torch.float32]
    for tag, matmul in (("layer-wise + matmul f32", hip.MATMUL_F32), ("layer-wise, three-part split", hip.MATMUL_SPLIT_BF16_X6)):
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W0)
        eng.set_train_recompute(False, fuse_dgrad_bn=0, fuse_gemm_red=0)
        eng.set_matmul_mode(matmul)
        eng.train_step(frames, labels, 1e-3)
        torch.cuda.synchronize()
        print("==", tag, "   (relative L2 vs the f64 oracle: HIP / f32 CPU oracle)")
        rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)  # noqa: E731
        for l in sp.layers:
            if l.scope not in z64:
                continue
            shp = z64[l.scope].shape
            zz = layer_view(eng, l.idx, 0, *shp)
            aa = layer_view(eng, l.idx, 1, *shp) if l.scope in t64 and t64[l.scope].shape == shp else None
            dz = layer_view(eng, l.idx, 3, *shp) if l.scope.startswith("MobilenetV2") else None
            line = "%2d %-44s z %.1e / %.1e" % (l.idx, l.scope.replace("MobilenetV2/", ""), rel(zz, z64[l.scope]), rel(z32[l.scope], z64[l.scope]))
            if aa is not None:
                line += "   a %.1e / %.1e" % (rel(aa, t64[l.scope]), rel(t32[l.scope], t64[l.scope]))
            if dz is not None:
                line += "   dz %.1e / %.1e" % (rel(dz, g64[l.scope]), rel(g32[l.scope], g64[l.scope]))
            print(line)
        eng.close()


def seeds_main():
    """The head's ReLUs as the source of the gap: per seed, the head tensors' gradient errors (HIP default step and f32 CPU oracle, vs f64)
    beside the number of aspp0 / concat_projection activations each evaluation puts on the other side of zero than the f64 oracle does."""
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    sp = spec.build_spec()
    heads = ["aspp0", "concat_projection"]
    hl = {l.scope: l for l in sp.layers}
    names = ["aspp0/weights:0", "concat_projection/weights:0", "logits/semantic/weights:0", "MobilenetV2/expanded_conv_16/project/weights:0"]
    print("seed | flips vs f64 (aspp0, concat_projection): HIP / f32 CPU | elements within 1e-4 rms of zero | L2 error of "
          + ", ".join(x.replace(":0", "") for x in names) + ": HIP / f32 CPU")
    for seed in range(n):
        W0 = weights.synthetic_weights(sp, seed=seed)
        frames, labels = synth.SyntheticVideo(H, B, CI, seed=7 + seed).clip()
        fr32 = frames.astype(np.float32)
        taps = {}
        grads = {}
        for dt in (torch.float64, torch.float32):
            o = StudentOracle(W0, CI, dtype=dt)
            tp = {}
            params = dict(o.vars)
            for v in sp.trainable:
                params[v.name] = o.vars[v.name].clone().requires_grad_(True)
            z = o.reduced_logits(o.logits_full(fr32, "train", params, tp))
            target, weight = o.label_targets(labels)
            loss = o.loss_from_reduced(z, target, weight)
            g = torch.autograd.grad(loss, [params[v.name] for v in sp.trainable])
            grads[dt] = {v.name: gg for v, gg in zip(sp.trainable, g)}
            taps[dt] = {k: tp[k].detach().numpy().astype(np.float64) for k in heads}
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W0)
        eng.train_step(frames, labels, 1e-3)
        torch.cuda.synchronize()
        g = eng.grads.cpu().numpy().astype(np.float64)
        e_hip = errors(sp, g, grads[torch.float64])
        e_32 = errors(sp, {k: v.numpy() for k, v in grads[torch.float32].items()}, grads[torch.float64])
        flips_hip, flips_32, risk = [], [], []
        for k in heads:
            t64 = taps[torch.float64][k]
            a_hip = layer_view(eng, hl[k].idx, 1, *t64.shape)
            flips_hip.append(int(((a_hip > 0) != (t64 > 0)).sum()))
            flips_32.append(int(((taps[torch.float32][k] > 0) != (t64 > 0)).sum()))
            pos = t64[t64 > 0]
            risk.append(int((pos < 1e-4 * np.sqrt((t64 ** 2).mean())).sum()))      # just above zero (those just below are invisible after the ReLU)
        eng.close()
        print("%4d | %s / %s | %s | %s" % (seed, flips_hip, flips_32, risk,
                                            "  ".join("%.1e / %.1e" % (e_hip[x][1], e_32[x][1]) for x in names)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "layers":
        return layers_main()
    if len(sys.argv) > 1 and sys.argv[1] == "seeds":
        return seeds_main()
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    name = sys.argv[4] if len(sys.argv) > 4 else "aspp0/weights:0"
    sp = spec.build_spec()
    W0 = weights.synthetic_weights(sp, seed=seed)
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=7 + seed).clip()
    fr32 = frames.astype(np.float32)
    _, g64 = StudentOracle(W0, CI, dtype=torch.float64).gradients(fr32, labels)
    _, g32 = StudentOracle(W0, CI).gradients(fr32, labels)
    e32 = errors(sp, {k: v.numpy() for k, v in g32.items()}, g64)
    print("f32 CPU oracle: %s max-norm %.2e  L2 %.2e" % (name, *e32[name]))

    forms = [
        ("default", {}),
        ("fuse_gemm_red=0", {"gemm_red": 0}),
        ("fuse_gemm_red=1 (fwd stats only)", {"gemm_red": 1}),
        ("fuse_gemm_red=2 (bwd sums only)", {"gemm_red": 2}),
        ("fuse_dgrad_bn=0", {"dgrad_bn": 0}),
        ("train_recompute=0", {"recompute": 0}),
        ("matmul f32", {"matmul": hip.MATMUL_F32}),
        ("layer-wise (recompute 0, dgrad_bn 0, gemm_red 0)", {"recompute": 0, "dgrad_bn": 0, "gemm_red": 0}),
        ("layer-wise + matmul f32", {"recompute": 0, "dgrad_bn": 0, "gemm_red": 0, "matmul": hip.MATMUL_F32}),
        ("layer-wise + matmul f32 + one stream", {"recompute": 0, "dgrad_bn": 0, "gemm_red": 0, "matmul": hip.MATMUL_F32, "overlap": 0}),
    ]
    well = [v.name for v in sp.trainable if e32[v.name][1] < 1e-3]
    print("%d of %d tensors have an f32-CPU L2 error < 1e-3" % (len(well), len(sp.trainable)))
    for tag, opt in forms:
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W0)
        eng.set_train_recompute(bool(opt.get("recompute", 1)), fuse_dgrad_bn=opt.get("dgrad_bn", 2), fuse_gemm_red=opt.get("gemm_red", 3))
        if "matmul" in opt:
            eng.set_matmul_mode(opt["matmul"])
        if "overlap" in opt:
            hip.check(eng.lib.ams_student_set_option(eng._h, hip.OPT_OVERLAP_WGRAD, opt["overlap"]))
        eng.train_step(frames, labels, 1e-3)
        g = eng.grads.cpu().numpy().astype(np.float64)
        eg = errors(sp, g, g64)
        ratios = sorted(((eg[n][1] / max(e32[n][1], 1e-12), n) for n in well), reverse=True)[:5]
        print("%-52s %s max-norm %.2e (x%.1f)  L2 %.2e (x%.1f) | worst ratios among well-conditioned: %s"
              % (tag, name, eg[name][0], eg[name][0] / e32[name][0], eg[name][1], eg[name][1] / e32[name][1],
                 ", ".join("%s x%.1f (%.1e)" % (n.replace("MobilenetV2/", "").replace(":0", ""), r, eg[n][1]) for r, n in ratios)))
        eng.close()


if __name__ == "__main__":
    main()
