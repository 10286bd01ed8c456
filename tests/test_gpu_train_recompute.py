"""Fine-tune step of the early blocks without their expanded tensors (ams_amd/csrc/k_xdw_train.hip, AMS_OPT_TRAIN_RECOMPUTE) and with the
one-kernel depthwise backward of the stride-16 blocks (k_conv.hip dw3x3_dgrad_bn_kernel, AMS_OPT_FUSE_DGRAD_BN) against the
layer-by-layer step of the same engine: same mathematics, different summation orders, so every gradient tensor, the BN moving
statistics and the loss must agree at f32 level.  (Both forms are held to the f64 oracle by tests/test_gpu_network.py and
tests/test_gpu_fullsize.py, which run the default = recompute form.)  Sizes cover odd and even block inputs: the stride-2 blocks pad
differently (SAME: pad before = 0 or 1), which moves the parity classes of the transposed depthwise conv."""
import numpy as np
import pytest
import torch

from ams_amd import spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine

pytestmark = pytest.mark.gpu

CI = [0, 1, 2, 10, 11, 13]


def _step(H, B, recompute, seed=3, steps=1):
    W0 = Wt.synthetic_weights(S.build_spec(), seed)
    fr, lb = synth.SyntheticVideo(H, B, CI, seed=seed).clip()
    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W0)
    eng.set_train_recompute(recompute, fuse_dgrad_bn=recompute)       # both fused forms on, or the layer-by-layer step
    losses = []
    for _ in range(steps):
        losses.append(eng.train_step(fr, lb, 1e-3).cpu().numpy().copy())
    out = {"loss": losses, "grads": eng.grads.cpu().numpy().copy(), "stats": eng.stats.cpu().numpy().copy(),
           "params": eng.params.cpu().numpy().copy(), "spec": eng.spec}
    eng.close()
    return out


@pytest.mark.parametrize("H,B", [(64, 2), (62, 3), (96, 1), (50, 2), (34, 5)])
def test_recompute_step_matches_layerwise_step(H, B):
    a = _step(H, B, True)
    b = _step(H, B, False)
    assert a["loss"][0][1] == b["loss"][0][1]
    # the two forms take their BN statistics differently (f32 partial sums / the Gram matrix of the block input in f64): the loss sums agree
    # to 3e-5, and the fused form is no further from the f64 oracle's loss than the layer-wise one (+ 5e-6)
    assert a["loss"][0][0] == pytest.approx(b["loss"][0][0], rel=3e-5)
    import torch
    from oracle.student_torch import StudentOracle
    W0 = Wt.synthetic_weights(S.build_spec(), 3)
    fr, lb = synth.SyntheticVideo(H, B, CI, seed=3).clip()
    loss64, _ = StudentOracle(W0, CI, dtype=torch.float64).gradients(fr.astype(np.float32), lb)
    ea, eb = (abs(x["loss"][0][0] / x["loss"][0][1] - loss64) / abs(loss64) for x in (a, b))
    print("loss vs the f64 oracle: fused %.2e, layer-wise %.2e" % (ea, eb))
    assert ea <= eb + 5e-6
    spec = a["spec"]
    worst = []
    gmax = float(np.abs(b["grads"]).max())
    for v in spec.trainable:
        ga, gb = a["grads"][v.offset:v.offset + v.size].astype(np.float64), b["grads"][v.offset:v.offset + v.size].astype(np.float64)
        # a shift in front of a training-mode BN has no effect: the beta / bias gradients of such layers are exactly 0 in real
        # arithmetic and pure rounding noise here -> measured against the step's gradient scale, not their own
        den = max(np.abs(gb).max(), 1e-4 * gmax)
        err = np.abs(ga - gb).max() / den
        worst.append((err, v.name))
    worst.sort(reverse=True)
    print("recompute vs layer-wise, worst tensors:", ["%s %.2e" % (n, e) for e, n in worst[:5]])
    # Two f32 evaluations of this graph: the forward statistics differ in the last bits (summation order), ReLU6 masks flip on a few
    # elements and the BN backward chain amplifies that ~1e5 x (the f32 CPU oracle is 1e-2 .. 3e-2 from f64 on the same
    # tensors).  A wrong tap, parity class or coefficient shows up as O(1) here; the f64 oracle arbitrates in tests/test_gpu_network.py.
    assert worst[0][0] < 0.1, worst[:5]
    ga, gb = a["grads"].astype(np.float64), b["grads"].astype(np.float64)
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    print("gradient cosine recompute vs layer-wise: %.7f" % cos)
    assert cos > 0.9999
    early = [e for e, n in worst if any(("expanded_conv_%d/" % k) in n for k in range(1, 7))]
    assert len(early) >= 6 * 9
    # moving statistics: the stride-16 layers see as few as 64 samples at these sizes, so one flipped ReLU6 mask upstream moves a
    # variance by ~1e-3 of itself; everything in front of the first flip agrees to ~1e-6
    np.testing.assert_allclose(a["stats"], b["stats"], rtol=1e-3, atol=3e-5)


def test_recompute_step_is_reproducible():
    a = _step(64, 2, True, steps=2)
    b = _step(64, 2, True, steps=2)
    assert np.array_equal(a["grads"], b["grads"]) and np.array_equal(a["params"], b["params"]) and np.array_equal(a["stats"], b["stats"])


@pytest.mark.parametrize("H,B", [(64, 2), (62, 3), (128, 4)])
@pytest.mark.parametrize("layerwise", [False, True])
def test_operand_transforms_change_no_bit(H, B, layerwise):
    """AMS_OPT_FUSE_OPERAND_BN moves the depthwise layers' BN + activation into the operand loads of their consumers with the same IEEE
    operations in the same order: gradients, parameters after two steps and moving statistics must be BIT-identical with the pass written
    (0) and on load (1) — in the fused step and in the layer-by-layer step (where most consumers take the materialising fallback)."""
    W0 = Wt.synthetic_weights(S.build_spec(), 5)
    fr, lb = synth.SyntheticVideo(H, B, CI, seed=5).clip()
    ref = None
    for bits in (0, 1):
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W0)
        if layerwise:
            eng.set_train_recompute(False, fuse_dgrad_bn=False, fuse_gemm_red=0)
        eng.set_fuse_operand_bn(bits)
        l0 = eng.train_step(fr, lb, 1e-3).cpu().numpy().copy()
        g = eng.grads.cpu().numpy().copy()
        eng.train_step(fr, lb, 1e-3)
        out = (l0, g, eng.params.cpu().numpy().copy(), eng.stats.cpu().numpy().copy())
        eng.close()
        if ref is None:
            ref = out
        else:
            for k, (x, y) in enumerate(zip(ref, out)):
                assert np.array_equal(x, y), "fuse_operand_bn=%d changes %s (max diff %g)" % (bits, ("the loss", "the gradients", "the parameters", "the moving statistics")[k],
                                                                                         np.abs(x.astype(np.float64) - y).max())


def test_weight_gradient_hand_over_batching_changes_no_bit():
    """AMS_OPT_WGRAD_FORK_EVERY only moves the moment a weight gradient is handed to the side stream (nothing it reads is overwritten inside
    the step): loss, gradients, parameters after two steps and moving statistics are BIT-identical for 1 (default), 3 and 64 per hand-over."""
    H, B = 96, 3
    W0 = Wt.synthetic_weights(S.build_spec(), 6)
    fr, lb = synth.SyntheticVideo(H, B, CI, seed=6).clip()
    ref = None
    for n in (1, 3, 64):
        eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=True)
        eng.load_variables(W0)
        eng.set_wgrad_fork_every(n)
        l0 = eng.train_step(fr, lb, 1e-3).cpu().numpy().copy()
        g = eng.grads.cpu().numpy().copy()
        eng.train_step(fr, lb, 1e-3)
        out = (l0, g, eng.params.cpu().numpy().copy(), eng.stats.cpu().numpy().copy())
        eng.close()
        if ref is None:
            ref = out
        else:
            for k, (x, y) in enumerate(zip(ref, out)):
                assert np.array_equal(x, y), "wgrad_fork_every=%d changes %s" % (n, ("the loss", "the gradients", "the parameters", "the moving statistics")[k])
