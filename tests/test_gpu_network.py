"""Whole-network parity through the C ABI engine and the SemanticNetwork boundary, against the CPU oracle.

Bars (BASELINE.json north star): low-resolution logits within 1e-3 relative (f32), label maps identical to the
oracle's except where the oracle's own top-2 margin is below the logit tolerance (a tie the reference itself
would break by summation order), confusion matrix / loss consistent with the labels, one optimisation step's
gradients, Adam update and BN moving averages within 1e-3 relative.
"""
import ctypes as C
import random
from collections import deque

import numpy as np
import pytest
import torch

from ams_amd import exp_configs, hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from ams_amd.semantic_network import SemanticNetwork

pytestmark = pytest.mark.gpu

CI = [0, 1, 2, 10, 11, 13]


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


@pytest.fixture(scope="module")
def W0():
    return Wt.synthetic_weights(S.build_spec(), seed=0)


@pytest.fixture(scope="module")
def clip64():
    return synth.SyntheticVideo(64, 6, CI).clip()


def _oracle(W, dtype=torch.float32):
    from oracle.student_torch import StudentOracle
    return StudentOracle(W, CI, dtype=dtype)


def _check_labels(got, oracle_logits_sel, tol):
    """labels must equal the oracle's argmax wherever the oracle's top-2 margin exceeds ``tol``."""
    want = np.argmax(oracle_logits_sel, axis=-1)
    srt = np.sort(oracle_logits_sel, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    bad = (got != want)
    assert not np.any(bad & (margin > tol)), "label mismatch on a pixel with margin %g" % margin[bad].max()
    return bad.mean()


@pytest.mark.parametrize("H,B,matmul", [(64, 2, hip.MATMUL_F32), (48, 3, hip.MATMUL_F32), (64, 2, hip.MATMUL_SPLIT_BF16),
                                        (128, 2, hip.MATMUL_SPLIT_BF16), (64, 2, hip.MATMUL_SPLIT_BF16_X6),
                                        (128, 3, hip.MATMUL_SPLIT_BF16_X6), (62, 3, hip.MATMUL_SPLIT_BF16_X6), (34, 5, hip.MATMUL_SPLIT_BF16_X6),
                                        (50, 1, hip.MATMUL_SPLIT_BF16_X6), (64, 2, hip.MATMUL_SPLIT_F16), (128, 3, hip.MATMUL_SPLIT_F16),
                                        (62, 3, hip.MATMUL_SPLIT_F16), (34, 5, hip.MATMUL_SPLIT_F16), (50, 1, hip.MATMUL_SPLIT_F16)])
def test_frozen_inference_matches_oracle(W0, H, B, matmul):
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=3).clip()
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.set_matmul_mode(matmul)
    eng.set_fuse_expand_dw(2 if matmul != hip.MATMUL_F32 else 0)          # 2: every supported block fused; 0: layer-by-layer plan
    eng.set_fuse_dw_project(matmul == hip.MATMUL_SPLIT_BF16)              # the optional depthwise+project kernel too
    eng.set_fuse_first_block(matmul != hip.MATMUL_F32)                    # first block: one kernel vs three
    eng.set_fuse_block(matmul != hip.MATMUL_F32)                          # early blocks: one kernel each vs layer by layer
    eng.freeze()
    o = _oracle(W0)
    with torch.no_grad():
        low = o.forward_lowres(frames.astype(np.float32), "frozen").numpy()
        full = o.reduced_logits(o.logits_full(frames.astype(np.float32), "frozen")).numpy()
    got_lab, conf, loss = eng.predict_with_metric(frames, labels, hip.MODE_FROZEN)
    h, w = eng.lowres
    got_low = eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].cpu().numpy()
    err = rel(got_low, low)
    assert err < 1e-3, "low-res logits rel err %g" % err
    mismatch = _check_labels(got_lab.cpu().numpy(), full, tol=2e-3 * np.abs(low).max())
    # f32-level plans: at most one tie pixel of these small maps; the two-part split (logits 2e-4 .. 5e-4) a handful
    npix = got_lab.numel()
    assert mismatch * npix <= (1 if matmul != hip.MATMUL_SPLIT_BF16 else 8), "%d label mismatches of %d" % (round(mismatch * npix), npix)
    # metrics are consistent with the oracle
    p, cm, l = o.predict_with_metric(frames.astype(np.float32), labels, "frozen")
    got_cm = conf.cpu().numpy()
    assert got_cm.sum() == cm.sum()
    assert np.abs(got_cm - cm).sum() <= 2 * (got_lab.cpu().numpy() != p).sum()
    ls = loss.cpu().numpy()
    assert ls[0] / ls[1] == pytest.approx(l, rel=1e-3)
    # predict (no labels) returns the same label map; uint8 and float32 frames agree
    lab2 = eng.predict(frames, hip.MODE_FROZEN)
    assert torch.equal(lab2, got_lab)
    lab3 = eng.predict(frames.astype(np.float32), hip.MODE_FROZEN)
    assert torch.equal(lab3, got_lab)
    eng.close()


@pytest.mark.parametrize("H,W,B", [(64, 128, 3), (34, 68, 5), (50, 100, 1), (128, 256, 2), (256, 512, 3), (512, 1024, 2), (200, 72, 2)])
def test_first_block_walking_form_same_bits(W0, H, W, B, knobs):
    """first_block_walk_kernel (a block walks tiles; table, taps and weight fragments staged once, the bytes of the next tile's taps requested
    under this tile's phases) against one tile per block (AMS_FB_WALK=0): identical bits at any cap of tiles per block, border tiles,
    interior tiles and sizes that are no multiple of the tile included."""
    frames = np.random.default_rng(H + B).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = StudentEngine(CI, H, W, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    h, w = eng.lowres
    low = lambda: eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].cpu().numpy().copy()
    knobs(AMS_FB_WALK="0")
    lab0 = eng.predict(frames)
    low0 = low()
    for cap in (None, "1", "2", "5"):
        knobs(AMS_FB_WALK=cap)
        lab = eng.predict(frames)
        assert torch.equal(lab, lab0) and np.array_equal(low(), low0), cap
    eng.close()


def test_live_forward_uses_batch_statistics(W0, clip64):
    frames, labels = clip64
    B = 3
    eng = StudentEngine(CI, 64, 128, max_batch=B, trainable=True)
    eng.load_variables(W0)
    o = _oracle(W0)
    with torch.no_grad():
        low = o.forward_lowres(frames[:B].astype(np.float32), "train").numpy()
    stats_before = eng.stats.clone()
    eng.predict(frames[:B], hip.MODE_LIVE)
    h, w = eng.lowres
    got_low = eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].cpu().numpy()
    assert rel(got_low, low) < 2e-3
    assert torch.equal(stats_before, eng.stats), "inference on the live graph must not touch the moving averages"
    eng.close()


def _compare_train_state(eng, o, before, lr, steps, tag, grads64):
    """Parameters after `steps` Adam iterations vs the f64 oracle.

    Adam's normalised step (m / sqrt(v)) is +-lr on the first iteration whatever the gradient's size.  Entries whose
    true gradient is zero or at round-off level (e.g. the beta of a BN that feeds another training-mode BN: exactly zero
    gradient) therefore take noise-driven +-lr steps in ANY f32 implementation, the reference included.  The bar: no
    entry may differ by more than the 2*lr*steps an opposite sign can produce; among entries with a significant
    gradient (> 5 % of a tensor whose own maximum is > 1e-3 of the global one) at most 1 % may differ by more than 10 %
    of lr*steps.  BN moving statistics (plain averages) must agree to 1e-3."""
    got = eng.get_variables()
    want = o.get_vars()
    gnorm = max(float(gv.abs().max()) for gv in grads64.values())
    for v in eng.spec.trainable:
        d_got = got[v.name].astype(np.float64) - before[v.name]
        d_want = want[v.name].astype(np.float64) - before[v.name]
        diff = np.abs(d_got - d_want)
        assert diff.max() <= 2.05 * lr * steps, "%s: %s update off by %g" % (tag, v.name, diff.max())
        g = np.abs(grads64[v.name].numpy())
        if g.max() < 1e-3 * gnorm:
            continue
        sig = g > 0.05 * g.max()
        frac = (diff[sig] > 0.1 * lr * steps).mean()
        assert frac <= 0.01 + 1.0 / sig.sum(), "%s: %s %.3f%% of significant updates differ" % (tag, v.name, 100 * frac)
    for v in eng.spec.stats:
        assert rel(got[v.name], want[v.name]) < 1e-3, "%s: %s" % (tag, v.name)


def test_train_step_matches_oracle(W0, clip64):
    """Gradient bar.  On this graph (54 training-mode BNs over as few as 180 samples, ReLU6 clipping, random weights with
    dead channels) an f32 evaluation is itself 1-6 % away from the f64 one on the worst tensors, CPU and GPU alike, so the
    f32 torch oracle cannot arbitrate at 1e-3 (measured: per-tensor relative L2 error vs f64 has median ~1e-2 for BOTH the
    f32 CPU oracle and the HIP path, at 64x128 and at 128x256; tools/debug_grads.py).  The HIP path is therefore held to
    the f32 error class: (a) per tensor, relative L2 error vs the f64 oracle no worse than max(6e-2, 4x the f32 CPU
    oracle's own error) — the gross-error bar: the worst tensor of ONE f32 evaluation scatters between 4e-3 and 3e-2 from seed
    to seed for every form of the step and for the f32 CPU oracle alike (tests/grad_noise_study.py), so the error class itself is
    held statistically by test_gradient_error_class_over_seeds below — and a median over tensors no worse than 3x the f32 CPU
    oracle's median + 5e-3; (b) cosine similarity of the whole gradient with
    the f64 gradient >= 0.9995; (c) loss within 1e-3; (d) the Adam update and BN moving averages as in
    _compare_train_state."""
    frames, labels = clip64
    B = 4
    eng = StudentEngine(CI, 64, 128, max_batch=B, trainable=True)
    eng.load_variables(W0)
    o = _oracle(W0, torch.float64)
    o32 = _oracle(W0, torch.float32)
    lr = 1e-3
    for step in range(3):
        # two f32/f64 runs drift apart through the noise-driven updates described in _compare_train_state, so every step
        # starts from the oracle's weights (Adam moments and step count stay the engine's own)
        before = {k: np.asarray(v, dtype=np.float64) for k, v in o.get_vars().items()}
        eng.load_variables(o.get_vars())
        fr, lb = frames[step:step + B], labels[step:step + B]
        loss_o, grads_o = o.gradients(fr.astype(np.float32), lb)
        o32.restore(o.get_vars())
        _, grads_32 = o32.gradients(fr.astype(np.float32), lb)
        ls = eng.train_step(fr, lb, lr).cpu().numpy()
        assert ls[0] / ls[1] == pytest.approx(loss_o, rel=1e-3), "loss at step %d" % step
        g = eng.grads.cpu().numpy().astype(np.float64)
        gnorm = max(float(gv.abs().max()) for gv in grads_o.values())
        flat_want = np.concatenate([grads_o[v.name].numpy().reshape(-1) for v in eng.spec.trainable])
        cos = float(g @ flat_want / (np.linalg.norm(g) * np.linalg.norm(flat_want)))
        assert cos > 0.9995, "step %d: gradient cosine %.6f" % (step, cos)
        errs_gpu, errs_f32 = [], []
        for v in eng.spec.trainable:
            want = grads_o[v.name].numpy().reshape(-1)
            floor = max(np.linalg.norm(want), 1e-3 * gnorm * np.sqrt(want.size))
            e_gpu = np.linalg.norm(g[v.offset:v.offset + v.size] - want) / floor
            e_f32 = np.linalg.norm(grads_32[v.name].numpy().reshape(-1).astype(np.float64) - want) / floor
            errs_gpu.append(e_gpu)
            errs_f32.append(e_f32)
        order = np.argsort(errs_gpu)[::-1][:4]
        print("step %d worst gradient tensors vs f64 (HIP, f32 CPU):" % step,
              [(eng.spec.trainable[k].name, "%.2e" % errs_gpu[k], "%.2e" % errs_f32[k]) for k in order])
        well = [k for k in range(len(errs_f32)) if errs_f32[k] < 1e-3]
        ratios = sorted(((errs_gpu[k] / max(errs_f32[k], 1e-12), eng.spec.trainable[k].name) for k in well), reverse=True)[:5]
        print("step %d worst HIP / f32-CPU ratios among the %d tensors whose f32-CPU error is < 1e-3:" % (step, len(well)),
              [(n, "x%.1f" % r) for r, n in ratios])
        for k, v in enumerate(eng.spec.trainable):
            # a tensor the f32 CPU evaluation gets to < 7.5e-3 must come out below 3e-2 (one flipped ReLU6 element upstream costs up to
            # ~1e-2 on a small tensor); the noisier ones are held to 4x the f32 CPU oracle's own error, 6e-2 at least
            bar = max(3e-2, 4 * errs_f32[k]) if errs_f32[k] < 7.5e-3 else max(6e-2, 4 * errs_f32[k])
            assert errs_gpu[k] <= bar, "gradient of %s: HIP %.2e vs f32 CPU %.2e (bar %.1e)" % (v.name, errs_gpu[k], errs_f32[k], bar)
        assert np.median(errs_gpu) <= 3 * np.median(errs_f32) + 5e-3, (np.median(errs_gpu), np.median(errs_f32))
        o.train_step(fr.astype(np.float32), lb, lr)
        _compare_train_state(eng, o, before, lr, 1, "after step %d" % (step + 1), grads_o)
    assert eng.adam_step == 3
    # Adam moments accumulated over the three steps (linear in the gradients, so they track the oracle closely)
    m = eng.adam_m.cpu().numpy()
    for name in ("aspp0/weights:0", "MobilenetV2/expanded_conv_3/depthwise/depthwise_weights:0", "logits/semantic/biases:0"):
        v = eng.spec.by_name[name]
        want = o.adam_m[name].numpy().reshape(-1)
        assert np.linalg.norm(m[v.offset:v.offset + v.size] - want) / np.linalg.norm(want) < 5e-2, name
    eng.close()



def test_gradient_error_class_over_seeds():
    """One f32 evaluation of this gradient is a noisy measurement of the f64 one (ReLU6 masks flip on last-bit differences and the 54
    training-mode BNs amplify them), so the per-tensor error of any single seed is a lottery ticket.  Over several seeds the
    DISTRIBUTION is what identifies the error class: the HIP step's worst-tensor and median-tensor relative L2 errors vs the f64
    oracle, averaged over the seeds, must stay within 1.5x (+2e-3) of the f32 CPU oracle's own on the same inputs."""
    H, B, seeds = 64, 4, range(6)
    worst_gpu, worst_f32, med_gpu, med_f32 = [], [], [], []
    for seed in seeds:
        W = Wt.synthetic_weights(S.build_spec(), seed=seed)
        fr, lb = synth.SyntheticVideo(H, B, CI, seed=seed + 100).clip()
        _, grads_o = _oracle(W, torch.float64).gradients(fr.astype(np.float32), lb)
        _, grads_32 = _oracle(W, torch.float32).gradients(fr.astype(np.float32), lb)
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W)
        eng.train_step(fr, lb, 1e-3)
        g = eng.grads.cpu().numpy().astype(np.float64)
        gnorm = max(float(gv.abs().max()) for gv in grads_o.values())
        e_gpu, e_f32 = [], []
        for v in eng.spec.trainable:
            want = grads_o[v.name].numpy().reshape(-1)
            floor = max(np.linalg.norm(want), 1e-3 * gnorm * np.sqrt(want.size))
            e_gpu.append(np.linalg.norm(g[v.offset:v.offset + v.size] - want) / floor)
            e_f32.append(np.linalg.norm(grads_32[v.name].numpy().reshape(-1).astype(np.float64) - want) / floor)
        eng.close()
        worst_gpu.append(max(e_gpu)); worst_f32.append(max(e_f32)); med_gpu.append(np.median(e_gpu)); med_f32.append(np.median(e_f32))
    print("worst tensor, mean over seeds: HIP %.2e  f32 CPU %.2e;  median tensor: HIP %.2e  f32 CPU %.2e"
          % (np.mean(worst_gpu), np.mean(worst_f32), np.mean(med_gpu), np.mean(med_f32)))
    assert max(worst_gpu) < 6e-2
    assert np.mean(worst_gpu) <= 1.5 * np.mean(worst_f32) + 2e-3
    assert np.mean(med_gpu) <= 1.5 * np.mean(med_f32) + 2e-3

def _layer_a(eng, layer_idx, shape):
    import ctypes as C
    off, n = C.c_size_t(), C.c_size_t()
    hip.check(eng.lib.ams_student_layer_tensor(eng._h, layer_idx, 1, C.byref(off), C.byref(n)))
    b, h, w, c = shape
    return eng.arena[off.value:off.value + 4 * b * h * w * c].view(torch.float32).view(b, h, w, c).cpu().numpy()


def test_head_gradients_are_f32_level_unless_a_relu_element_flips():
    """Where the "30x head-gradient gap" of round 3 comes from, as a test.  At 64x128 x 2 frames the head's two ReLUs see
    90 x 256 values whose forward f32 error is ~2e-5 relative, so about ONE element per evaluation sits within that of zero; an f32
    evaluation that puts it on the other side than the f64 oracle changes the head's gradient tensors by 1e-3 .. 1e-2 (one element of 23 040
    is 1 / sqrt(23040) = 6.6e-3 of the tensor's norm).  It happens to the HIP path and to the f32 CPU oracle alike, on different seeds
    (tools/grad_gap_bisect.py seeds).  So: on every seed where NEITHER evaluation flips a head ReLU, every head tensor of the HIP step must be
    within 4x the f32 CPU oracle's own distance from f64 (no floor); with a flip on either side the bar is the gross one."""
    H, B = 64, 2
    sp = S.build_spec()
    heads = {l.scope: l for l in sp.layers if l.scope in ("aspp0", "concat_projection")}
    names = [v.name for v in sp.trainable if v.name.split("/")[0] in ("aspp0", "concat_projection", "logits")] + \
            ["MobilenetV2/expanded_conv_16/project/weights:0"]
    tight = 0
    for seed in range(8):
        W = Wt.synthetic_weights(sp, seed=seed)
        fr, lb = synth.SyntheticVideo(H, B, CI, seed=seed + 7).clip()
        ref = {}
        for dt in (torch.float64, torch.float32):
            o = _oracle(W, dt)
            taps = {}
            params = dict(o.vars)
            for v in sp.trainable:
                params[v.name] = o.vars[v.name].clone().requires_grad_(True)
            z = o.reduced_logits(o.logits_full(fr.astype(np.float32), "train", params, taps))
            target, weight = o.label_targets(lb)
            g = torch.autograd.grad(o.loss_from_reduced(z, target, weight), [params[n] for n in names])
            ref[dt] = ({n: x.numpy().astype(np.float64).reshape(-1) for n, x in zip(names, g)}, {k: taps[k].detach().numpy() for k in heads})
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W)
        eng.train_step(fr, lb, 1e-3)
        torch.cuda.synchronize()
        g = eng.grads.cpu().numpy().astype(np.float64)
        flips_hip = sum(int(((_layer_a(eng, l.idx, ref[torch.float64][1][k].shape) > 0) != (ref[torch.float64][1][k] > 0)).sum()) for k, l in heads.items())
        flips_f32 = sum(int(((ref[torch.float32][1][k] > 0) != (ref[torch.float64][1][k] > 0)).sum()) for k in heads)
        eng_spec = eng.spec
        eng.close()
        rows = []
        for n in names:
            v = eng_spec.by_name[n]
            want = ref[torch.float64][0][n]
            e_hip = np.linalg.norm(g[v.offset:v.offset + v.size] - want) / np.linalg.norm(want)
            e_f32 = np.linalg.norm(ref[torch.float32][0][n] - want) / np.linalg.norm(want)
            rows.append((n, e_hip, e_f32))
        worst = max(rows, key=lambda r: r[1] / max(r[2], 1e-12))
        print("seed %d: head ReLU elements on the other side of zero than f64: HIP %d, f32 CPU %d; worst ratio %s HIP %.1e / f32 CPU %.1e"
              % (seed, flips_hip, flips_f32, worst[0], worst[1], worst[2]))
        for n, e_hip, e_f32 in rows:
            if flips_hip == 0 and flips_f32 == 0:
                assert e_hip <= 4 * e_f32 + 1e-7, "seed %d, no flipped head ReLU: %s HIP %.2e vs f32 CPU %.2e" % (seed, n, e_hip, e_f32)
            else:
                assert e_hip <= max(3e-2, 4 * e_f32), "seed %d: %s HIP %.2e vs f32 CPU %.2e" % (seed, n, e_hip, e_f32)
        tight += int(flips_hip == 0 and flips_f32 == 0)
    assert tight >= 1, "no seed without a flipped head ReLU: the tight bar was never exercised"


def test_free_running_schedule_tracks_oracle(W0):
    """a short free-running distillation schedule (no re-synchronisation with the oracle between steps).  Two
    evaluations of this graph drift apart from the second step on (noise-driven sign flips of Adam's normalised steps on
    zero-gradient entries, see _compare_train_state), so the bar is the f32 error class again: the HIP path's loss curve may
    leave the f64 oracle's by no more than 3x what the f32 CPU oracle's does (+1 %), the first step agrees to 1e-3, the loss
    falls, and the student's mIoU against the teacher labels on held-out frames ends as close to the f64 oracle's as the f32 CPU
    oracle's does (3x + 0.5 point: on 3 x 8192 pixels a few hundred flipped labels move the mIoU by points in any f32 run)."""
    from ams_amd.utils import calculate_miou
    frames, labels = synth.SyntheticVideo(64, 7, CI, seed=5).clip()
    B, steps, lr = 4, 6, 1e-3
    eng = StudentEngine(CI, 64, 128, max_batch=B, trainable=True)
    eng.load_variables(W0)
    o64, o32 = _oracle(W0, torch.float64), _oracle(W0, torch.float32)
    fr, lb = frames[:B], labels[:B]
    held_f, held_l = frames[4:7], labels[4:7]
    curve_g, curve_64, curve_32 = [], [], []
    for _ in range(steps):
        ls = eng.train_step(fr, lb, lr).cpu().numpy()
        curve_g.append(ls[0] / ls[1])
        curve_64.append(float(o64.train_step(fr.astype(np.float32), lb, lr)))
        curve_32.append(float(o32.train_step(fr.astype(np.float32), lb, lr)))
    curve_g, curve_64, curve_32 = np.array(curve_g), np.array(curve_64), np.array(curve_32)
    assert abs(curve_g[0] - curve_64[0]) <= 1e-3 * curve_64[0]
    dev_g, dev_32 = np.abs(curve_g - curve_64) / curve_64, np.abs(curve_32 - curve_64) / curve_64
    assert np.all(dev_g <= 3 * np.maximum.accumulate(dev_32) + 1e-2), (dev_g, dev_32)
    assert curve_g[-1] < 0.5 * curve_g[0] and curve_64[-1] < 0.5 * curve_64[0]
    # held-out frame, live graph (BN batch statistics, as the server evaluates it)
    _, conf, _ = eng.predict_with_metric(held_f, held_l, hip.MODE_LIVE)
    _, cm_64, _ = o64.predict_with_metric(held_f.astype(np.float32), held_l, "live")
    _, cm_32, _ = o32.predict_with_metric(held_f.astype(np.float32), held_l, "live")
    miou_g = np.nanmean(calculate_miou(conf.cpu().numpy().astype(np.float64), nan=True))
    miou_64 = np.nanmean(calculate_miou(np.asarray(cm_64, dtype=np.float64), nan=True))
    miou_32 = np.nanmean(calculate_miou(np.asarray(cm_32, dtype=np.float64), nan=True))
    assert abs(miou_g - miou_64) <= 3 * abs(miou_32 - miou_64) + 0.005, (miou_g, miou_32, miou_64)
    eng.close()


def test_parameter_delta_after_ten_free_running_steps(W0):
    """SURVEY 8 d6, "parameter delta after 1 and 10 steps", with a bar on the entries that carry signal instead of the
    2*lr*steps worst case.  Adam moves EVERY entry by ~lr per step whatever its gradient's size, so an f32 and an f64 evaluation of
    this graph drift apart quickly: measured here for the f32 CPU oracle against the f64 one, after 1 step 0.2 % of the consistently
    moving entries differ by more than 10 % of their own displacement, after 2 steps 30 %, after 10 steps 70 % (mean relative
    deviation 0.004 -> 0.12 -> 0.28).  SURVEY's literal "rel <= 1e-3" is therefore not reachable by any f32 run, the reference's
    own GPU/CPU pair included; what CAN be held is the error class, step by step: on the entries the f64 oracle has moved by more
    than half the maximum (0.5 * lr * k after k steps), the HIP run's mean relative deviation and its fraction of entries off by
    more than 10 % stay within 1.5x the f32 CPU oracle's (+1 %), and after the first step they are below 1 % / 0.5 % outright."""
    frames, labels = synth.SyntheticVideo(64, 4, CI, seed=8).clip()
    B, steps, lr = 4, 10, 1e-3
    eng = StudentEngine(CI, 64, 128, max_batch=B, trainable=True)
    eng.load_variables(W0)
    o64, o32 = _oracle(W0, torch.float64), _oracle(W0, torch.float32)
    names = [v.name for v in eng.spec.trainable]
    cat = lambda d: np.concatenate([np.asarray(d[k], np.float64).reshape(-1) for k in names])  # noqa: E731
    w0 = cat(W0)
    rows = []
    for k in range(1, steps + 1):
        eng.train_step(frames, labels, lr)
        o64.train_step(frames.astype(np.float32), labels, lr)
        o32.train_step(frames.astype(np.float32), labels, lr)
        d_g = eng.params.cpu().numpy().astype(np.float64) - w0
        d_64, d_32 = cat(o64.get_vars()) - w0, cat(o32.get_vars()) - w0
        sig = np.abs(d_64) > 0.5 * lr * k
        assert sig.sum() > 0.05 * sig.size, "step %d: too few consistently moving entries (%d)" % (k, sig.sum())
        dev_g = np.abs(d_g[sig] - d_64[sig]) / np.abs(d_64[sig])
        dev_32 = np.abs(d_32[sig] - d_64[sig]) / np.abs(d_64[sig])
        rows.append((k, int(sig.sum()), dev_g.mean(), (dev_g > 0.1).mean(), dev_32.mean(), (dev_32 > 0.1).mean()))
    for r in rows:
        print("step %2d: %7d entries | HIP mean dev %.4f, >10%%: %.4f | f32 CPU oracle mean dev %.4f, >10%%: %.4f" % r)
    k, _, mg, og, m32, o32f = rows[0]
    assert mg < 1e-2 and og < 5e-3, rows[0]
    for k, _, mg, og, m32, o32f in rows:
        assert mg <= 1.5 * m32 + 0.01, "step %d: mean deviation %.4f vs f32 CPU oracle %.4f" % (k, mg, m32)
        assert og <= 1.5 * o32f + 0.01, "step %d: fraction off by >10%% %.4f vs f32 CPU oracle %.4f" % (k, og, o32f)
    eng.close()


def test_masked_step_reverts_weights_but_advances_moments(W0, clip64):
    frames, labels = clip64
    B = 2
    eng = StudentEngine(CI, 64, 128, max_batch=B, trainable=True)
    eng.load_variables(W0)
    rng = np.random.default_rng(0)
    mask = (rng.random(eng.spec.n_trainable) < 0.1).astype(np.uint8)
    before = eng.params.clone()
    eng.train_step(frames[:B], labels[:B], 1e-3, mask=torch.as_tensor(mask).to(eng.device))
    after = eng.params.cpu().numpy()
    b = before.cpu().numpy()
    assert np.array_equal(after[mask == 0], b[mask == 0])
    changed = (after != b)
    assert changed[mask == 1].mean() > 0.9
    assert (eng.adam_m.cpu().numpy() != 0).mean() > 0.9
    eng.close()


@pytest.mark.parametrize("nan_grads", [False, True])
def test_batch_without_a_valid_pixel(W0, clip64, nan_grads):
    """utils/graph_utils.py:408 on a batch whose every teacher label is ignored (255): tf.reduce_mean over the empty tf.boolean_mask is NaN, but
    its backward is an empty tensor that the mask's gather gradient densifies to zeros — TensorFlow's step leaves NaN loss and ZERO gradients.
    Default: exactly that (weights, Adam moments, moving statistics finite, and the next ordinary step trains); AMS_OPT_NAN_GRADS = 1 turns
    the gradients into NaN for callers that want such a batch to fail loudly."""
    frames, labels = clip64
    B = 2
    eng = StudentEngine(CI, 64, 128, max_batch=B, trainable=True)
    eng.load_variables(W0)
    if nan_grads:
        eng.set_nan_grads(True)
    before = eng.params.clone()
    ls = eng.train_step(frames[:B], np.full_like(labels[:B], 255), 1e-3).cpu().numpy()
    assert ls[1] == 0                                               # no valid pixel: the API's loss is sum / count = NaN (semantic_network.py)
    g = eng.grads.cpu().numpy()
    if nan_grads:
        assert np.isnan(g).any()
        eng.close()
        return
    assert np.isfinite(g).all() and not g.any()                     # zero gradients everywhere, BN parameters included
    assert torch.isfinite(eng.params).all() and torch.isfinite(eng.adam_m).all() and torch.isfinite(eng.adam_v).all()
    assert torch.isfinite(eng.stats).all()
    # Adam with g = 0 from zero moments: m = v = 0, step = 0 / (0 + eps) = 0 — the weights do not move at all
    assert torch.equal(eng.params, before)
    ls = eng.train_step(frames[:B], labels[:B], 1e-3).cpu().numpy()          # ... and the student still trains afterwards
    assert ls[1] > 0 and np.isfinite(ls[0])
    assert torch.isfinite(eng.params).all() and not torch.equal(eng.params, before)
    eng.close()


def test_semantic_network_surface(W0, tmp_path):
    np.random.seed(0)
    random.seed(0)
    H = 32
    vid = synth.SyntheticVideo(H, 12, CI, seed=1)
    frames, labels = vid.clip()
    cw = exp_configs.class_weights(25)
    meta = str(tmp_path / "student")
    Wt.save_npy(meta + ".npy", W0)
    net = SemanticNetwork(meta, class_weights_exp=cw, height=H, frozen=False, scale=[1], mini_batch_size=4, lr=1e-3,
                          train_biases_only=False, regularize=False, masked_gradients=False)
    assert net.class_count == 6 and net.take_array.tolist() == [0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 3, 4, 0, 5, 0, 0, 0, 0, 0]
    lab = net.predict_input(frames[:2])
    assert lab.dtype == np.int32 and lab.shape == (2, H, 2 * H) and lab.max() < 6
    out = net.predict_with_metric(frames[:1], labels[:1])
    assert len(out) == 5 and out[1].shape == (6, 6) and out[1].dtype == np.float64 and isinstance(out[4], np.float32)
    fm, lm = deque(frames, maxlen=20), deque(labels, maxlen=20)
    net.train_with_deque(fm, lm, 3, 'full_model')
    assert len(net.last_losses) == 3 and all(np.isfinite(net.last_losses))
    assert len(net.train_params) == 272 and all(m.all() for m in net.curr_mask)
    v_trained = net.get_vars()
    assert "MobilenetV2/Conv/weights/Adam:0" in v_trained
    # restore_initial resets the weights but not Adam
    step_before = net.engine.adam_step
    net.restore_initial()
    assert net.engine.adam_step == step_before == 3
    assert np.array_equal(net.get_vars()["aspp0/weights:0"], W0["aspp0/weights:0"])
    # coordinate descent
    net.train_with_deque(fm, lm, 2, 'coord_desc_rand')
    assert len(net.train_params) == 164 and len(net.curr_mask) == 164
    frac = sum(int(m.sum()) for m in net.curr_mask) / sum(m.size for m in net.curr_mask)
    assert 0.08 < frac < 0.12
    net.restore_initial()
    net.train_with_deque(fm, lm, 2, 'coord_desc_auto')
    frac = sum(int(m.sum()) for m in net.curr_mask) / sum(m.size for m in net.curr_mask)
    assert 0.05 < frac < 0.15
    with pytest.raises(NameError):
        net.train_with_deque(fm, lm, 1, 'no_such_strategy')
    assert not net.process_lock.locked()
    # mem_frac (per_process_gpu_memory_fraction, SemanticNetwork.py:73): a student whose arena does not fit the share is refused
    with pytest.raises(MemoryError):
        SemanticNetwork(meta, class_weights_exp=cw, height=H, frozen=False, scale=[1], mini_batch_size=4, lr=1e-3, mem_frac=1e-7)
    # server -> edge hand-off
    net.save_to_frozen_graph(str(tmp_path / "edge"))
    edge = SemanticNetwork(str(tmp_path / "edge"), class_weights_exp=cw, height=H, frozen=True)
    l1 = edge.predict_input(frames[:1])
    l2, cm, iou, miou, loss = edge.predict_with_metric(frames[:1], labels[:1])
    assert np.array_equal(l1, l2) and cm.sum() > 0 and np.isfinite(loss)
    with pytest.raises(AssertionError):
        edge.train_with_deque(fm, lm, 1)
    cmx, ioux, mioux = net.calc_cross_miou(np.stack([labels[0], labels[1]]))
    assert cmx.shape == (6, 6) and 0 <= mioux <= 1
    col = net.colorize(label=l1[0])
    assert col.shape == (H, 2 * H, 3)
    net.close_model()
    edge.close_model()


# ---------------------------------------------------------------------------------------------------------
# data-parallel fine-tune step on the real engine: 2 ranks (gloo, both on cuda:0) x 2 frames == 1 rank x 4 frames
# ---------------------------------------------------------------------------------------------------------
def _dp_worker(rank, world, port, tmp):
    import os
    import torch.distributed as dist
    from ams_amd.dist import ArenaAllReduce, init_from_env, shard_bounds
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_from_env("gloo")
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames, labels = synth.SyntheticVideo(64, 4, CI, seed=9).clip()
    b, e = shard_bounds(4, rank, world)
    eng = StudentEngine(CI, 64, 128, max_batch=2, trainable=True)
    eng.load_variables(W0)
    red = ArenaAllReduce(eng.arena)
    ls = eng.train_step(frames[b:e], labels[b:e], 1e-3, allreduce=red, global_batch=4).cpu().numpy()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(tmp, "dp_grads.npy"), eng.grads.cpu().numpy())
        np.save(os.path.join(tmp, "dp_params.npy"), eng.params.cpu().numpy())
        np.save(os.path.join(tmp, "dp_stats.npy"), eng.stats.cpu().numpy())
        np.save(os.path.join(tmp, "dp_loss.npy"), ls)
        np.save(os.path.join(tmp, "dp_calls.npy"), np.array([red.calls, red.bytes]))
    eng.close()
    dist.destroy_process_group()


def test_data_parallel_step_equals_single_process(W0, tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    frames, labels = synth.SyntheticVideo(64, 4, CI, seed=9).clip()
    eng = StudentEngine(CI, 64, 128, max_batch=4, trainable=True)
    eng.load_variables(W0)
    ls = eng.train_step(frames, labels, 1e-3).cpu().numpy()
    dp_ls = np.load(tmp_path / "dp_loss.npy")
    # global CE sum and valid count: the count is exact; the sum carries the f32 summation order of the ranks' BN statistics (two ranks
    # add their halves, a single process adds block partials): measured 1.3e-6 with the early blocks' statistics taken from the block
    # input (k_xdw_train.hip), 2e-7 layer by layer
    assert dp_ls[1] == ls[1] and dp_ls[0] == pytest.approx(ls[0], rel=1e-5)
    g, dg = eng.grads.cpu().numpy().astype(np.float64), np.load(tmp_path / "dp_grads.npy").astype(np.float64)
    cos = float(g @ dg / (np.linalg.norm(g) * np.linalg.norm(dg)))
    assert cos > 0.99999, cos
    assert rel(np.load(tmp_path / "dp_stats.npy"), eng.stats.cpu().numpy()) < 1e-5    # SyncBN: same moving averages
    calls, nbytes = np.load(tmp_path / "dp_calls.npy")
    assert calls == 54 * 2 + 2                     # BN fwd + BN bwd per layer, loss, gradients
    assert nbytes > 4 * eng.spec.n_trainable
    eng.close()


def test_delta_payload_matches_reference_host_loop(W0):
    """Downlink delta (run.py:316-336): mask bits + masked parameters as fp16.  The device gather/cast must produce the
    bytes of the reference's per-variable host loop, for a coordinate-descent mask and for the full model."""
    H = 64
    frames, labels = synth.SyntheticVideo(H, 4, CI, seed=2).clip()
    for strategy in ("coord_desc_rand", "full_model"):
        net = SemanticNetwork("unused", class_weights_exp=exp_configs.class_weights(25), height=H, scale=[1], mini_batch_size=2, lr=1e-3,
                              coord_frac=0.1, masked_gradients=strategy != "full_model", initial_variables=W0)
        np.random.seed(3)
        random.seed(3)
        net.train_with_deque(deque(frames), deque(labels), 2, strategy)
        want = bytearray()
        for val in net.curr_mask:
            want += np.packbits(val.flatten()).tobytes()
        for p_, m_ in zip(net.train_params, net.curr_mask):
            want += p_[m_].astype(np.float16).tobytes()
        got = net.delta_payload()
        assert got == bytes(want), strategy
        net.close_model()


@pytest.mark.parametrize("H,ci", [(64, CI), (48, [2, 8, 9, 10, 11, 13]), (32, list(range(19)))])
def test_cross_confusion_matches_oracle(H, ci):
    """phi-score confusion matrix (SemanticNetwork.py:124-139, :184-194): two teacher label maps, pixels valid in BOTH, rows = the
    earlier map.  Labels carry ids outside the subset, ids >= 19 and 255; the matrix must equal the oracle's exactly, and the
    SemanticNetwork wrapper's mIoU must be the oracle matrix's mIoU."""
    from oracle.student_torch import cross_miou_confusion
    from ams_amd.utils import calculate_miou
    rng = np.random.default_rng(H)
    a = rng.integers(0, 19, (H, 2 * H)).astype(np.uint8)
    b = a.copy()
    flip = rng.random(a.shape) < 0.2
    b[flip] = rng.integers(0, 19, int(flip.sum()))
    for lab in (a, b):
        lab[rng.random(a.shape) < 0.05] = 255
        lab[rng.random(a.shape) < 0.02] = 19
        lab[rng.random(a.shape) < 0.02] = 200
    eng = StudentEngine(ci, H, 2 * H, max_batch=1, trainable=False)
    got = eng.cross_confusion(np.stack([a, b])).cpu().numpy()
    want = cross_miou_confusion(a, b, ci)
    assert got.dtype == np.int64 and np.array_equal(got, want.astype(np.int64))
    assert got.sum() == (np.isin(a, ci) & np.isin(b, ci)).sum() and not np.array_equal(got, got.T)
    eng.close()
    if ci == CI:
        cw = np.zeros((19, 1))
        cw[ci] = 1
        net = SemanticNetwork("unused", class_weights_exp=cw, height=H, frozen=True, cross_miou_compat=True,
                              frozen_graph=_frozen_graph_for(ci, H))
        cm, iou, miou = net.calc_cross_miou(np.stack([a, b]))
        assert cm.dtype == np.float64 and np.array_equal(cm, want)
        assert miou == pytest.approx(np.nanmean(calculate_miou(want, nan=True)), rel=1e-12)
        net.close_model()


def _frozen_graph_for(ci, H):
    from ams_amd.semantic_network import FrozenGraph
    return FrozenGraph(Wt.synthetic_weights(S.build_spec(), seed=0), ci, H, 19)


def test_rccl_communicator_inside_the_engine(W0):
    """The library's own RCCL communicator (comm.hip: librccl resolved with dlopen, PyTorch's copy reused) as a genuine ONE-rank
    communicator: ncclCommInitRank, ncclAllReduce on the launch stream, and the fine-tune step with its 110 collectives issued by the
    engine — which must equal the plain step bit for bit (a sum over one rank).  Multi-rank RCCL needs one GPU per rank (bench.py)."""
    from ams_amd.dist import RcclComm
    comm = RcclComm(0, 1, torch.device("cuda:0"), real_single_rank=True)
    t = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    want = t.clone()
    comm.all_reduce(t)
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    frames, labels = synth.SyntheticVideo(64, 2, CI, seed=4).clip()
    outs = []
    for use_comm in (False, True):
        eng = StudentEngine(CI, 64, 128, max_batch=2, trainable=True)
        eng.load_variables(W0)
        ls = eng.train_step(frames, labels, 1e-3, comm=comm if use_comm else None, global_batch=2).cpu().numpy()
        torch.cuda.synchronize()
        outs.append((ls, eng.params.cpu().numpy().copy(), eng.stats.cpu().numpy().copy()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    calls, nbytes = comm.stats()
    assert calls == 1 + 54 * 2 + 2 and nbytes > 4 * 2113043           # the probe above, BN forward + backward per layer, loss, gradients
    # diagnostic timing (bench.py collective_ms_per_step): a HIP event pair around every collective of one step
    eng = StudentEngine(CI, 64, 128, max_batch=2, trainable=True)
    eng.load_variables(W0)
    comm.set_timing(True)
    ls_t = eng.train_step(frames, labels, 1e-3, comm=comm, global_batch=2).cpu().numpy()
    total_ms, longest_ms, spans = comm.timing()
    comm.set_timing(False)
    eng.close()
    assert spans == 54 * 2 + 2 and 0.0 < longest_ms <= total_ms < 1e3
    assert np.array_equal(ls_t, outs[0][0])                            # timing changes nothing
    comm.close()


def test_weights_beyond_fp16_range_fall_back_to_three_bf16_parts(W0):
    """the default product form splits operands into two fp16 parts; a weight of 65520 or more would become hi = inf, lo = -inf and the
    logits NaN.  Every freeze checks the frozen weights: a layer outside fp16's range runs the three-part bf16 form (f32's range) instead.  One
    project layer and one head layer scaled by 1e6 (their BN folds the scale back, so the network's function is unchanged up to rounding)."""
    H, B = 64, 2
    frames, _ = synth.SyntheticVideo(H, B, CI, seed=3).clip()
    W = {k: np.array(v, copy=True) for k, v in W0.items()}
    scaled = ["MobilenetV2/expanded_conv_14/project", "aspp0"]
    for scope in scaled:
        W[scope + "/weights:0"] = W[scope + "/weights:0"] * np.float32(1e6)
        W[scope + "/BatchNorm/moving_mean:0"] = W[scope + "/BatchNorm/moving_mean:0"] * np.float32(1e6)
        W[scope + "/BatchNorm/moving_variance:0"] = W[scope + "/BatchNorm/moving_variance:0"] * np.float32(1e12)
    assert max(float(np.abs(W[s_ + "/weights:0"]).max()) for s_ in scaled) > 65504
    o = _oracle(W, torch.float64)
    with torch.no_grad():
        low = o.forward_lowres(frames.astype(np.float32), "frozen").numpy()
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W)
    eng.freeze()
    n = C.c_int32()
    hip.check(eng.lib.ams_student_f16_fallback_layers(eng._h, C.byref(n)))
    assert n.value == 2
    eng.predict(frames)
    h, w = eng.lowres
    got = eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].cpu().numpy()
    assert np.isfinite(got).all()
    assert rel(got, low) < 2e-4
    eng.load_variables(W0)                                 # back inside the range: the fp16 form returns
    eng.freeze()
    hip.check(eng.lib.ams_student_f16_fallback_layers(eng._h, C.byref(n)))
    assert n.value == 0
    eng.close()


def test_host_returning_calls_hand_back_uint8_labels(W0):
    """predict_host / predict_with_metric_host (what SemanticNetwork.predict_input / predict_with_metric call): the label maps cross PCIe as
    uint8 and are widened on the host, the metrics come per frame and are summed there — the same int32 maps, confusion matrix and loss sums
    (bit for bit) as the device-side calls."""
    H, B = 64, 3
    frames, labels = synth.SyntheticVideo(H, B, CI, seed=6).clip()
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    lab_dev = eng.predict(frames).cpu().numpy()
    lab_host = eng.predict_host(frames)
    assert lab_host.dtype == np.int32 and lab_host.shape == (B, H, 2 * H) and np.array_equal(lab_host, lab_dev)
    l_d, c_d, s_d = eng.predict_with_metric(frames, labels)
    l_h, c_h, s_h = eng.predict_with_metric_host(frames, labels)
    assert l_h.dtype == np.int32 and np.array_equal(l_h, l_d.cpu().numpy())
    assert c_h.dtype == np.int64 and np.array_equal(c_h, c_d.cpu().numpy())
    assert s_h.dtype == np.float64 and np.array_equal(s_h, s_d.cpu().numpy())
    one = eng.predict_host(frames[:1])                      # a smaller call after a larger one: the output block is re-laid out
    assert np.array_equal(one, lab_dev[:1])
    eng.close()
