"""BASELINE.json full size (512 x 1024): direct parity against the CPU oracle on two frames, and size-independent properties
of the path on a larger batch (determinism, input-dtype and batch-composition invariance, consistency of the metric
outputs, the fine-tune step against the f64 oracle with the split-bf16 kernels engaged)."""
import numpy as np
import pytest
import torch

from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine

pytestmark = pytest.mark.gpu

CI = [0, 1, 2, 10, 11, 13]
H = 512


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)


@pytest.fixture(scope="module")
def W0():
    return Wt.synthetic_weights(S.build_spec(), seed=0)


@pytest.fixture(scope="module")
def clip():
    return synth.SyntheticVideo(H, 8, CI, seed=1).clip()


def _lowres(eng, B):
    h, w = eng.lowres
    return eng.logits_lowres.view(-1, h, w, 32)[:B, :, :, :19].cpu().numpy()


def test_frozen_inference_full_size_matches_oracle(W0, clip):
    """North-star bar at the benchmark's own size: low-res logits within 1e-3 relative of the f32 oracle, label maps equal
    wherever the oracle's top-2 margin exceeds the logit tolerance; default kernel plan (three-part split-bf16 late layers,
    fused first block, fused expand+depthwise) and the faster two-part split."""
    from oracle.student_torch import StudentOracle
    frames, labels = clip
    B = 2
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    o = StudentOracle(W0, CI)
    fr = frames[:B].astype(np.float32)
    with torch.no_grad():
        low = o.forward_lowres(fr, "frozen").numpy()
        full = o.reduced_logits(o.logits_full(fr, "frozen")).numpy()
    lab, conf, loss = eng.predict_with_metric(frames[:B], labels[:B], hip.MODE_FROZEN)
    err = rel(_lowres(eng, B), low)
    assert err < 2e-4, "low-res logits rel err %g at 512x1024 (default plan: f32-level)" % err
    # ... and against the f64 oracle: the default plan (two fp16 parts per operand, 3 MFMAs) must sit at the f32 level — the f32 CPU oracle itself
    # is 4.2e-5 from f64 on these frames, exact-f32 MFMA 3.7e-5, the three-part bf16 form of rounds 1-4 3.7e-5 (VERDICT r4 bar: 5e-5)
    with torch.no_grad():
        low64 = StudentOracle(W0, CI, dtype=torch.float64).forward_lowres(fr, "frozen").numpy()
    err64 = rel(_lowres(eng, B), low64)
    print("512x1024 low-res logits vs the f64 oracle: %.3e (f32 CPU oracle: %.3e)" % (err64, rel(low, low64)))
    assert err64 < 5e-5
    got = lab.cpu().numpy()
    want = np.argmax(full, axis=-1)
    srt = np.sort(full, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    bad = got != want
    assert not np.any(bad & (margin > 2e-3 * np.abs(low).max()))
    # measured: 1 pixel of 1 048 576 (a top-2 tie at f32 summation-order level); the bar allows ten
    print("full-size label mismatches vs the f32 oracle: %d of %d" % (int(bad.sum()), bad.size))
    assert bad.mean() <= 1e-5
    p, cm, l = o.predict_with_metric(fr, labels[:B], "frozen")
    assert conf.sum().item() == cm.sum()
    ls = loss.cpu().numpy()
    assert ls[0] / ls[1] == pytest.approx(l, rel=1e-3)
    # the three-part bf16 form (rounds 1-4's default; what the fine-tune step still uses): the same level
    eng.set_matmul_mode(hip.MATMUL_SPLIT_BF16_X6)
    eng.predict(frames[:B])
    assert rel(_lowres(eng, B), low64) < 5e-5
    # the opt-in two-part bf16 split: inside the north-star tolerance, measurably above the f32 level
    eng.set_matmul_mode(hip.MATMUL_SPLIT_BF16)
    eng.predict(frames[:B])
    err3 = rel(_lowres(eng, B), low)
    assert err < err3 < 1e-3, (err, err3)
    eng.close()


def test_full_size_properties(W0, clip):
    frames, labels = clip
    B = 8
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    lab = eng.predict(frames)
    low = _lowres(eng, B).copy()
    # deterministic: the same call gives the same bits
    assert torch.equal(eng.predict(frames), lab)
    assert np.array_equal(_lowres(eng, B), low)
    # uint8 and float32 frames are the same input
    assert torch.equal(eng.predict(frames.astype(np.float32)), lab)
    # a frame's result does not depend on what else is in the batch (kernel choice may: tolerance at the 1e-5 level)
    single = eng.predict(frames[3:4])
    low1 = _lowres(eng, 1)
    assert rel(low1[0], low[3]) < 1e-4
    assert (single[0] != lab[3]).float().mean().item() < 1e-4
    rev = eng.predict(frames[::-1].copy())
    assert (rev.flip(0) != lab).float().mean().item() < 1e-4
    # metric outputs are consistent with each other and with the label map
    lab_m, conf, loss = eng.predict_with_metric(frames, labels)
    assert torch.equal(lab_m, lab)
    lut = np.full(256, -1)
    lut[CI] = np.arange(len(CI))
    t = lut[labels]
    valid = t >= 0
    assert conf.sum().item() == int(valid.sum()) == int(loss.cpu().numpy()[1])
    cm = np.zeros((len(CI), len(CI)), np.int64)
    np.add.at(cm, (t[valid], lab.cpu().numpy()[valid]), 1)
    assert np.array_equal(conf.cpu().numpy(), cm)
    # the fused first block: by default its stem runs as six bf16 products on table-looked-up parts (f32-level, like the early blocks'
    # expand); with exact-f32 products (block_x6 off) it does the arithmetic of the three kernels it replaces in the same order: identical bits
    eng.set_fuse_first_block(False)
    eng.predict(frames)
    assert rel(_lowres(eng, B), low) < 5e-5
    eng.set_block_x6(False)
    eng.predict(frames)
    low_exact = _lowres(eng, B).copy()
    eng.set_fuse_first_block(True)
    eng.predict(frames)
    assert np.array_equal(_lowres(eng, B), low_exact)
    eng.set_block_x6(True)
    # so does the streaming expand+depthwise kernel of the stride-16 blocks (engaged at this batch: 8 x 2145 rows), in the
    # three-part and the two-part split
    for mode in (0, 2):          # never / every supported block (the 160-channel blocks too)
        eng.set_fuse_expand_dw_stream(mode)
        eng.predict(frames)
        assert np.array_equal(_lowres(eng, B), low)
    eng.set_matmul_mode(hip.MATMUL_SPLIT_BF16)
    eng.predict(frames)
    low2 = _lowres(eng, B).copy()
    eng.set_fuse_expand_dw_stream(0)
    eng.predict(frames)
    assert np.array_equal(_lowres(eng, B), low2)
    eng.set_fuse_expand_dw_stream(1)
    assert not np.array_equal(low2, low)
    eng.set_matmul_mode(hip.MATMUL_SPLIT_BF16_X6)
    # exact-f32 plan vs the default split plan: logits agree at the f32 level
    eng.set_matmul_mode(hip.MATMUL_F32)
    eng.set_fuse_first_block(False)
    eng.set_fuse_expand_dw(0)
    eng.predict(frames[:2])
    assert rel(_lowres(eng, 2), low[:2]) < 2e-4
    eng.close()


@pytest.mark.parametrize("height,batch", [(512, 1), (512, 2), (512, 3), (256, 2), (256, 8)])
def test_whole_block_kernels_off_at_small_batches(W0, height, batch):
    """with the whole-block kernels off, the 24 -> 144 and 32 -> 192 stride-1 blocks stream on the exact-f32 form of the
    streaming kernel; their depthwise result must then NOT be handed over as fp16 pairs (that hand-over belongs to the fp16 form, Cin > 32) —
    at 512x1024 with 1..3 frames and at 256x512 the project GEMM is below its split threshold's other side and the plan used to ask for it:
    predict failed with AMS_E_INVALID.  The plan runs and stays f32-level from the default plan."""
    frames, _ = synth.SyntheticVideo(height, batch, CI, seed=11).clip()
    eng = StudentEngine(CI, height, 2 * height, max_batch=batch, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    lab = eng.predict(frames)
    low = _lowres(eng, batch).copy()
    eng.set_fuse_block(False)
    lab1 = eng.predict(frames)
    assert rel(_lowres(eng, batch), low) < 5e-5
    assert (lab1 != lab).float().mean().item() < 1e-4
    eng.close()


def test_bench_batch_fused_equals_unfused(W0):
    """At the benchmark's batch (32 frames: other segment / chunk plans than at 8) the streaming and weight-register kernels
    give the bits of the kernels they replace, and a frame's logits do not depend on the batch it travels in beyond the
    kernel-choice tolerance."""
    frames, _ = synth.SyntheticVideo(H, 32, CI, seed=3).clip()
    B = 32
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    lab = eng.predict(frames)
    low = _lowres(eng, B).copy()
    eng.set_fuse_expand_dw_stream(0)
    lab0 = eng.predict(frames)
    assert torch.equal(lab0, lab)
    assert np.array_equal(_lowres(eng, B), low)
    eng.set_fuse_expand_dw_stream(1)
    # whole-block kernels off: blocks 1-5 give the same bits either way (tests/test_gpu_kernels.py::test_whole_block_kernel); block 6's
    # project is exact f32 inside the block kernel and a three-part split GEMM outside it: f32-level both, not the same bits
    eng.set_fuse_block(False)
    lab1 = eng.predict(frames)
    assert rel(_lowres(eng, B), low) < 5e-5
    assert (lab1 != lab).float().mean().item() < 1e-4
    eng.set_fuse_block(True)
    for mode in (2, 0):                                   # first block: tile per wave / three kernels — exact-f32 stems: f32-level from the default
        eng.set_fuse_first_block(mode)
        labm = eng.predict(frames)
        assert rel(_lowres(eng, B), low) < 5e-5 and (labm != lab).float().mean().item() < 1e-4, mode
    eng.set_block_x6(False)                               # exact-f32 products everywhere in the early section: the three forms agree bit for bit
    eng.set_fuse_first_block(1)
    lab_e = eng.predict(frames)
    low_e = _lowres(eng, B).copy()
    for mode in (2, 0):
        eng.set_fuse_first_block(mode)
        labm = eng.predict(frames)
        assert torch.equal(labm, lab_e) and np.array_equal(_lowres(eng, B), low_e), mode
    eng.set_block_x6(True)
    eng.set_fuse_first_block(1)
    eng.predict(frames[5:6])
    assert rel(_lowres(eng, 1)[0], low[5]) < 1e-4
    eng.close()


def test_fine_tune_step_full_size_against_f64_oracle(W0, clip):
    """One 2-frame step at 512 x 1024: the late layers run the 3-part bf16 GEMMs and weight gradients here (4290 rows);
    loss within 1e-3 and the whole gradient within the f32 error class of the f64 oracle (cosine)."""
    from oracle.student_torch import StudentOracle
    frames, labels = clip
    B = 2
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W0)
    o = StudentOracle(W0, CI, dtype=torch.float64)
    loss_o, grads_o = o.gradients(frames[:B].astype(np.float32), labels[:B])
    ls = eng.train_step(frames[:B], labels[:B], 1e-3).cpu().numpy()
    assert ls[0] / ls[1] == pytest.approx(loss_o, rel=1e-3)
    g = eng.grads.cpu().numpy().astype(np.float64)
    flat = np.concatenate([grads_o[v.name].numpy().reshape(-1) for v in eng.spec.trainable])
    cos = float(g @ flat / (np.linalg.norm(g) * np.linalg.norm(flat)))
    assert cos > 0.9995, cos
    for name in ("aspp0/weights:0", "MobilenetV2/expanded_conv_16/project/weights:0", "MobilenetV2/expanded_conv_13/expand/weights:0"):
        v = eng.spec.by_name[name]
        want = grads_o[name].numpy().reshape(-1)
        e = np.linalg.norm(g[v.offset:v.offset + v.size] - want) / np.linalg.norm(want)
        assert e < 3e-2, (name, e)
    eng.close()


def test_fine_tune_step_config2_batch_against_f64_oracle(W0, clip):
    """BASELINE.json configs[2] at its own batch: ONE 8-frame step at 512 x 1024 (17160 rows at output stride 16: other GEMM tiles /
    weight-gradient plans than the 2-frame case above) against the f64 oracle — loss to 1e-3, the whole gradient by cosine, three
    late-layer tensors, one early one, the Adam update of the significant entries and the BN moving averages."""
    from oracle.student_torch import StudentOracle
    frames, labels = clip
    B, lr = 8, 1e-3
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W0)
    o = StudentOracle(W0, CI, dtype=torch.float64)
    loss_o, grads_o = o.gradients(frames[:B].astype(np.float32), labels[:B])
    stats_o = {k: (m.numpy().copy(), v.numpy().copy()) for k, (m, v) in o.last_batch_stats.items()}
    before = eng.params.cpu().numpy().astype(np.float64)
    ls = eng.train_step(frames[:B], labels[:B], lr).cpu().numpy()
    assert ls[0] / ls[1] == pytest.approx(loss_o, rel=1e-3)
    assert ls[1] == float(np.isin(labels[:B], CI).sum())
    g = eng.grads.cpu().numpy().astype(np.float64)
    flat = np.concatenate([grads_o[v.name].numpy().reshape(-1) for v in eng.spec.trainable])
    cos = float(g @ flat / (np.linalg.norm(g) * np.linalg.norm(flat)))
    assert cos > 0.9995, cos
    for name in ("aspp0/weights:0", "MobilenetV2/expanded_conv_16/project/weights:0", "MobilenetV2/expanded_conv_13/expand/weights:0",
                 "MobilenetV2/expanded_conv_2/depthwise/depthwise_weights:0"):
        v = eng.spec.by_name[name]
        want = grads_o[name].numpy().reshape(-1)
        e = np.linalg.norm(g[v.offset:v.offset + v.size] - want) / np.linalg.norm(want)
        assert e < 3e-2, (name, e)
    # every one of the 164 gradient tensors by relative L2.  A shift in front of a training-mode BN has no effect, so the beta / bias
    # gradients of the layers that feed one are exactly 0 in real arithmetic: for those the error is taken against the typical tensor
    # norm of the step instead of their own (which is rounding noise in the f64 oracle too).
    norms = np.array([np.linalg.norm(grads_o[v.name].numpy()) for v in eng.spec.trainable])
    floor = 1e-3 * np.median(norms)
    errs, noise = [], []
    for v in eng.spec.trainable:
        want = grads_o[v.name].numpy().reshape(-1)
        d = np.linalg.norm(g[v.offset:v.offset + v.size] - want)
        (errs if np.linalg.norm(want) >= floor else noise).append((d / max(np.linalg.norm(want), floor), v.name))
    errs.sort(reverse=True)
    noise.sort(reverse=True)
    print("8-frame 512x1024 step, relative L2 of %d gradient tensors vs f64: worst five %s; median %.2e; %d zero-gradient tensors, worst %s" %
          (len(errs), ["%s %.2e" % (n, e) for e, n in errs[:5]], float(np.median([e for e, _ in errs])), len(noise),
           ["%s %.2e" % (n, e) for e, n in noise[:2]]))
    assert len(errs) + len(noise) == 164 and len(errs) >= 120
    assert errs[0][0] < 3e-2, errs[:5]                     # f32 error class of this graph (the f32 CPU oracle: 1e-2 .. 3e-2 on the same tensors)
    assert float(np.median([e for e, _ in errs])) < 2e-2
    assert not noise or noise[0][0] < 0.2, noise[:3]       # against the floor: rounding noise stays well below the smallest real gradients
    # first Adam step: -lr * sign(g) wherever |g| is far above eps; entries whose f64 gradient is significant must move the
    # way the oracle's do
    after = eng.params.cpu().numpy().astype(np.float64)
    sig = np.abs(flat) > 1e-2 * np.abs(flat).max()
    step = after - before
    assert sig.sum() > 1000
    agree = np.sign(step[sig]) == -np.sign(flat[sig])
    assert agree.mean() > 0.999, agree.mean()
    np.testing.assert_allclose(np.abs(step[sig]), lr, rtol=2e-3)
    # BN moving averages of a first, a middle and a head layer: mean and UNBIASED variance of the 8-frame batch
    omd = float(np.float32(1.0) - np.float32(S.BN_DECAY))
    got = eng.get_variables()
    for scope in ("MobilenetV2/Conv", "MobilenetV2/expanded_conv_7/depthwise", "aspp0"):
        mu, var_u = stats_o[scope]
        mm0 = W0[scope + "/BatchNorm/moving_mean:0"].astype(np.float64)
        mv0 = W0[scope + "/BatchNorm/moving_variance:0"].astype(np.float64)
        assert rel(got[scope + "/BatchNorm/moving_mean:0"], mm0 - (mm0 - mu) * omd) < 1e-4, scope
        assert rel(got[scope + "/BatchNorm/moving_variance:0"], mv0 - (mv0 - var_u) * omd) < 1e-4, scope
    eng.close()


def test_bf16_variant_is_opt_in_and_close(W0, clip):
    """AMS_MATMUL_BF16 (SURVEY 8 cfg 2 / d6): plain bf16 products in the late layers.  It must change nothing unless selected, stay
    within a few percent of the default plan's logits (bf16 has 8 significand bits; 40 layers deep with random weights) and give
    label maps that differ on a small fraction of pixels only; the fine-tune step ignores it."""
    frames, labels = clip
    B = 2
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    lab = eng.predict(frames[:B])
    low = _lowres(eng, B).copy()
    eng.set_matmul_mode(hip.MATMUL_BF16)
    lab_b = eng.predict(frames[:B])
    dev_b = rel(_lowres(eng, B), low)
    mism = (lab_b != lab).float().mean().item()
    print("bf16 variant at 512x1024: logits max rel deviation %.3e, label mismatch fraction %.3e" % (dev_b, mism))
    assert 1e-4 < dev_b < 0.2 and mism < 0.05
    eng.set_matmul_mode(hip.MATMUL_DEFAULT)
    assert torch.equal(eng.predict(frames[:B]), lab) and np.array_equal(_lowres(eng, B), low)
    eng.close()


def test_two_stream_plan_is_two_half_batches(W0):
    """AMS_OPT_DUAL_STREAM: a 32-frame call run as two halves on two streams gives every frame the bits of a 16-frame call (same kernels in
    half-size launches, second half of every buffer), the automatic mode settles on one plan in its first call and then repeats its bits, and
    metrics computed after the join see the whole batch."""
    frames, labels = synth.SyntheticVideo(H, 32, CI, seed=5).clip()
    B = 32
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    eng.set_dual_stream(0)
    lab_a = eng.predict(frames[:16]).clone()
    low_a = _lowres(eng, 16).copy()
    lab_b = eng.predict(frames[16:]).clone()
    low_b = _lowres(eng, 16).copy()
    eng.set_dual_stream(2)                                # always two streams
    lab2 = eng.predict(frames)
    low2 = _lowres(eng, B).copy()
    assert torch.equal(lab2[:16], lab_a) and torch.equal(lab2[16:], lab_b)
    assert np.array_equal(low2[:16], low_a) and np.array_equal(low2[16:], low_b)
    # the benchmarked plan itself against the oracle: one frame out of each half of the 32-frame two-stream call
    from oracle.student_torch import StudentOracle
    o = StudentOracle(W0, CI)
    pick = [3, 29]
    fr = frames[pick].astype(np.float32)
    with torch.no_grad():
        low_o = o.forward_lowres(fr, "frozen").numpy()
        full_o = o.reduced_logits(o.logits_full(fr, "frozen")).numpy()
    assert rel(low2[pick], low_o) < 2e-4
    bad = lab2[pick].cpu().numpy() != np.argmax(full_o, axis=-1)
    print("two-stream 32-frame call, frames %s vs the f32 oracle: %d of %d labels differ, logits %.2e" % (pick, int(bad.sum()), bad.size,
                                                                                                         rel(low2[pick], low_o)))
    assert bad.mean() <= 1e-5
    lab_m, conf, loss = eng.predict_with_metric(frames, labels)
    assert torch.equal(lab_m, lab2)
    assert conf.sum().item() == int(np.isin(labels, CI).sum())
    eng.set_dual_stream(0)
    lab1 = eng.predict(frames)                            # one stream: the 32-frame plan, f32-level from the half-size plan
    assert rel(_lowres(eng, B), low2) < 1e-4 and (lab1 != lab2).float().mean().item() < 1e-4
    eng.set_dual_stream(1)                                # default: a fixed function of the batch size (32 frames -> two parts), never timed
    first = eng.predict(frames).clone()
    assert torch.equal(first, lab2)
    for _ in range(3):
        assert torch.equal(eng.predict(frames), first)
    eng.set_dual_stream(2, parts=4)                       # the caller's choice: four parts of eight frames = four 8-frame calls
    lab4 = eng.predict(frames).clone()
    eng.set_dual_stream(0)
    for k in range(4):
        assert torch.equal(eng.predict(frames[8 * k:8 * k + 8]), lab4[8 * k:8 * k + 8])
    eng.set_dual_stream(1, parts=2, autotune=True)        # opt-in: timed in the first call (median of three), the same bits from then on
    tuned = eng.predict(frames).clone()
    for _ in range(3):
        assert torch.equal(eng.predict(frames), tuned)
    assert any(torch.equal(tuned, x) for x in (lab1, lab2, lab4)) or (tuned != lab1).float().mean().item() < 1e-4
    eng.close()


def test_frames_per_pass_equal_single_frame_calls(W0):
    """ams_student_predict_frames: several frames labelled in ONE pass with per-frame metrics.  Each frame's label map, confusion matrix
    and loss sums are what its own one-frame call returns, bit for bit (batch-composition invariance incl. the loss: a pixel's loss enters
    the sums as a multiple of 2^-20, so no partial sum ever rounds)."""
    n = 7
    frames, labels = synth.SyntheticVideo(H, n, CI, seed=11).clip()
    ref = StudentEngine(CI, H, 2 * H, max_batch=1, trainable=False)
    ref.load_variables(W0)
    ref.freeze()
    want = [ref.predict_with_metric_host(frames[i:i + 1], labels[i:i + 1]) for i in range(n)]
    again = ref.predict_with_metric_host(frames[0:1], labels[0:1])
    assert np.array_equal(again[2], want[0][2])           # the loss sums are reproducible run to run (order-independent atomics)
    ref.close()
    eng = StudentEngine(CI, H, 2 * H, max_batch=4, trainable=False)
    eng.load_variables(W0)
    eng.freeze()
    for b0, nb in ((0, 2), (2, 3), (3, 4)):
        eng.predict_frames(frames[b0:b0 + nb], labels[b0:b0 + nb])
        labs, confs, losses = eng.fetch_frames()
        for k in range(nb):
            assert np.array_equal(labs[k], want[b0 + k][0][0]) and np.array_equal(confs[k], want[b0 + k][1]) and \
                np.array_equal(losses[k], want[b0 + k][2]), (b0, nb, k)
    labs_only = eng.predict_frames(frames[0:3])[0].cpu().numpy()       # no teacher: labels only
    assert all(np.array_equal(labs_only[k], want[k][0][0]) for k in range(3))
    eng.close()
