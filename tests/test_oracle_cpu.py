"""The oracle against itself: two independent restatements (NumPy tap-by-tap, PyTorch functional) must agree,
and the update rules must follow the TF1 forms of SURVEY.md Appendix C.  (Parity with TensorFlow itself is unpinned:
the reference ships no tests, vectors or weights; see oracle/student_torch.py.)"""
import numpy as np
import pytest
import torch

from ams_amd import spec as S, synth, weights as Wt
from oracle import student_np as ON
from oracle import student_torch as OT

CI = [0, 1, 2, 10, 11, 13]


@pytest.fixture(scope="module")
def setup():
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames, labels = synth.SyntheticVideo(32, 4, CI, seed=2).clip()
    return W0, frames.astype(np.float32), labels


def test_numpy_and_torch_restatements_agree(setup):
    W0, frames, labels = setup
    o = OT.StudentOracle(W0, CI)
    for mode, tol in (("frozen", 2e-4), ("train", 1e-3)):
        with torch.no_grad():
            lt = o.forward_lowres(frames, mode).numpy()
        ln = ON.forward_lowres(W0, frames, mode)
        assert lt.shape == ln.shape == (4, 3, 5, 19)
        assert np.abs(lt - ln).max() / np.abs(lt).max() < tol, mode
    p, cm, loss = o.predict_with_metric(frames, labels)
    p2, cm2, loss2 = ON.predict_with_metric(W0, frames, labels, CI)
    assert (p != p2).mean() < 2e-3 and np.abs(cm - cm2).sum() <= 2 * (p != p2).sum()
    assert loss == pytest.approx(loss2, rel=1e-4)
    assert cm.dtype == np.float64 and cm.sum() == (np.isin(labels, CI)).sum()


def test_even_height_uses_asymmetric_same_padding():
    """H not a multiple of 16 gives even feature sizes somewhere: pad_before != pad_after (Appendix C.1)."""
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames = synth.SyntheticVideo(24, 1, CI).clip()[0].astype(np.float32)[:, :, :40]      # 24 x 40
    o = OT.StudentOracle(W0, CI)
    with torch.no_grad():
        lt = o.forward_lowres(frames, "frozen").numpy()
    ln = ON.forward_lowres(W0, frames, "frozen")
    assert lt.shape == ln.shape and np.abs(lt - ln).max() / np.abs(lt).max() < 2e-4


def test_resize_bilinear_align_corners_endpoints_and_identity():
    x = np.random.default_rng(0).standard_normal((1, 3, 5, 2)).astype(np.float32)
    up = ON.resize_bilinear_align_corners(x, 9, 17)
    assert np.array_equal(up[:, ::4, ::4], x)                     # align_corners: source samples are reproduced exactly
    assert np.array_equal(ON.resize_bilinear_align_corners(x, 3, 5), x)
    upt = OT.resize_bilinear_align_corners(torch.as_tensor(x), 9, 17).numpy()
    assert np.array_equal(up, upt)
    one = ON.resize_bilinear_align_corners(x[:, :1, :1], 4, 6)
    assert np.all(one == x[:, :1, :1])                             # 1x1 input broadcasts


def test_label_path_ignores_ids_outside_the_subset(setup):
    W0, frames, labels = setup
    o = OT.StudentOracle(W0, CI)
    lab = np.array([[[0, 3, 13, 255, 18, 10]]])
    tgt, w = o.label_targets(lab)
    assert tgt.tolist() == [[[0, 0, 5, 0, 0, 3]]] and w.tolist() == [[[1, 0, 1, 0, 0, 1]]]
    none_valid = np.full_like(labels, 255)
    assert np.isnan(o.predict_with_metric(frames, none_valid)[2])


def test_adam_and_ema_follow_tf1_forms():
    w, g = np.array([1.0, -2.0]), np.array([0.5, -0.25])
    m = v = np.zeros(2)
    w1, m1, v1 = ON.adam_step(w, g, m, v, lr=1e-3, beta1_power=0.9, beta2_power=0.999)
    assert np.allclose(m1, 0.1 * g) and np.allclose(v1, 0.001 * g * g)
    assert np.allclose(w1, w - 1e-3 * np.sign(g), atol=1e-9)       # first step = lr * sign(g) (eps negligible)
    mv = ON.ema_update(np.float32(1.0), np.float32(2.0))
    assert mv == np.float32(1.0) - (np.float32(1.0) - np.float32(2.0)) * (np.float32(1.0) - np.float32(0.9))


def test_train_step_state_machine(setup):
    W0, frames, labels = setup
    o = OT.StudentOracle(W0, CI)
    l0 = o.train_step(frames, labels, 1e-3)
    l1 = o.train_step(frames, labels, 1e-3)
    l2 = o.train_step(frames, labels, 1e-3)
    assert l2 < l0 and np.isfinite([l0, l1, l2]).all()
    assert o.beta1_power == pytest.approx(0.9 ** 4)
    m_before = o.adam_m["aspp0/weights:0"].clone()
    o.restore(W0)                                                  # restore_initial: weights back, Adam untouched
    assert torch.equal(o.adam_m["aspp0/weights:0"], m_before) and o.beta1_power == pytest.approx(0.9 ** 4)
    assert np.array_equal(o.get_vars()["aspp0/weights:0"], W0["aspp0/weights:0"])
    # moving statistics moved towards the batch statistics by (1 - 0.9) per step
    o2 = OT.StudentOracle(W0, CI)
    o2.train_step(frames, labels, 1e-3)
    name = "MobilenetV2/Conv/BatchNorm/moving_mean:0"
    mu = o2.last_batch_stats["MobilenetV2/Conv"][0].numpy()
    want = W0[name] - (W0[name] - mu) * (np.float32(1) - np.float32(S.BN_DECAY))
    assert np.allclose(o2.get_vars()[name], want, rtol=1e-6, atol=1e-7)
    # masked step: unselected entries keep their bits, moments advance everywhere
    o3 = OT.StudentOracle(W0, CI)
    mask = {v.name: np.zeros(v.shape, bool) for v in o3.spec.trainable}
    mask["logits/semantic/biases:0"][:] = True
    o3.train_step(frames, labels, 1e-3, mask=mask)
    got = o3.get_vars()
    assert np.array_equal(got["aspp0/weights:0"], W0["aspp0/weights:0"])
    assert not np.array_equal(got["logits/semantic/biases:0"], W0["logits/semantic/biases:0"])
    assert float(o3.adam_m["aspp0/weights:0"].abs().max()) > 0


def test_cross_miou_confusion():
    a = np.array([[0, 1, 13, 5, 255, 10]])
    b = np.array([[0, 2, 13, 0, 0, 7]])
    cm = OT.cross_miou_confusion(a, b, CI)
    assert cm.sum() == 3 and cm[0, 0] == 1 and cm[1, 2] == 1 and cm[5, 5] == 1


def test_soft_teacher_loss_follows_the_published_definition(setup):
    """soft_teacher=True (utils/graph_utils.py:375-376, 403-408): labels = softmax(gather(teacher logits)), pixel loss = -sum_k p_k log softmax(z)_k
    (tf.nn.softmax_cross_entropy_with_logits), mask and mean from the HARD labels.  Checked against scalar Python on a 2 x 3 image, against the
    hard loss in the one-hot limit, and its gradient against (softmax(z) - p) / N."""
    import math
    W0, _frames, _labels = setup
    o = OT.StudentOracle(W0, CI, dtype=torch.float64)
    rng = np.random.default_rng(5)
    z19 = rng.standard_normal((1, 2, 3, 19))
    t19 = 2.0 * rng.standard_normal((1, 2, 3, 19))
    labels = np.array([[[0, 5, 13], [255, 1, 10]]], dtype=np.uint8)            # 5 and 255: outside the subset -> weight 0
    z = torch.as_tensor(z19).index_select(3, o.class_indices).requires_grad_(True)
    target, weight = o.label_targets(labels)
    probs = o.soft_targets(t19, 2, 3)
    loss = o.soft_loss_from_reduced(z, probs, weight)
    want, n = 0.0, 0
    for i in range(2):
        for j in range(3):
            if int(labels[0, i, j]) not in CI:
                continue
            zz = [z19[0, i, j, c] for c in CI]
            tt = [t19[0, i, j, c] for c in CI]
            zs = math.log(sum(math.exp(v) for v in zz))
            ts = sum(math.exp(v) for v in tt)
            want += -sum(math.exp(tv) / ts * (zv - zs) for tv, zv in zip(tt, zz))
            n += 1
    assert n == 4 and float(loss) == pytest.approx(want / n, rel=1e-12)
    (g,) = torch.autograd.grad(loss, z)
    sm = torch.softmax(z.detach(), dim=-1)
    want_g = (sm - probs) * (weight > 0).unsqueeze(-1) / n
    assert torch.allclose(g, want_g, atol=1e-14)
    # the one-hot limit: teacher logits that put all mass on the label's class give the hard loss
    peaked = np.full((1, 2, 3, 19), -1e4)
    for i in range(2):
        for j in range(3):
            if int(labels[0, i, j]) in CI:
                peaked[0, i, j, int(labels[0, i, j])] = 1e4
    hard = o.loss_from_reduced(z.detach(), target, weight)
    assert float(o.soft_loss_from_reduced(z.detach(), o.soft_targets(peaked, 2, 3), weight)) == pytest.approx(float(hard), rel=1e-12)
    # low-resolution teacher logits are resized like the student's own (align corners): grid points are reproduced exactly
    low = rng.standard_normal((1, 2, 2, 19))
    up = o.soft_targets(low, 2, 3)
    assert torch.allclose(up[:, :, 0], torch.softmax(torch.as_tensor(low)[:, :, 0].index_select(-1, o.class_indices), -1))
    assert torch.allclose(up[:, :, 2], torch.softmax(torch.as_tensor(low)[:, :, 1].index_select(-1, o.class_indices), -1))


def test_regularizer_follows_the_published_definition(setup):
    """regularize=True (utils/graph_utils.py:451-456): 0.01 * mean over tvars of sum(v^2) / 2; train_biases_only drops every name with 'weight'."""
    W0, frames, labels = setup
    o = OT.StudentOracle(W0, CI, dtype=torch.float64)
    names = [v.name for v in o.spec.trainable]
    assert len(names) == 164
    all_terms = [float((np.asarray(W0[n], np.float64) ** 2).sum() / 2) for n in names]
    assert float(o.regularizer(o.vars, False)) == pytest.approx(0.01 * np.mean(all_terms), rel=1e-12)
    bias_terms = [t for n, t in zip(names, all_terms) if 'weight' not in n]
    assert len(bias_terms) == 109                                               # 54 x (gamma, beta) + the logits biases
    assert float(o.regularizer(o.vars, True)) == pytest.approx(0.01 * np.mean(bias_terms), rel=1e-12)
    l0, g0 = o.gradients(frames[:1], labels[:1])
    l1, g1 = o.gradients(frames[:1], labels[:1], regularize=True)
    assert l1 == pytest.approx(l0 + 0.01 * np.mean(all_terms), rel=1e-10)
    n = "logits/semantic/biases:0" if "logits/semantic/biases:0" in g0 else [k for k in names if k.endswith("biases:0")][0]
    assert torch.allclose(g1[n] - g0[n], 0.01 / 164 * o.vars[n], atol=1e-12)
