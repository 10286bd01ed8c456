"""The HIP kernels, through the C ABI, against the hand-derived TF 1.15 semantics vectors (tests/golden/tf_semantics.json;
generator and the published definition behind each case: tests/golden/make_tf_semantics.py).  The CPU checkers are run
against the same file in tests/test_tf_semantics.py — here no oracle code is involved at all, only the vectors."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from ams_amd import hip, spec as S, weights as Wt
from ams_amd.engine import StudentEngine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def V(golden_dir):
    return json.loads((golden_dir / "tf_semantics.json").read_text())


@pytest.fixture(scope="module")
def lib():
    return hip.lib()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_KEEP = []


def dev(a, dtype=torch.float32):
    """device copy that stays alive: a temporary passed as P(dev(x)) would return its block to the caching allocator before the
    next temporary of the same call is allocated, and the second copy could land on top of the first."""
    t = torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(DEV).contiguous()
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def devu8(a):
    return dev(a, torch.uint8)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_depthwise_same_padding_on_even_sizes(lib, V):
    for c in V["depthwise"]["cases"]:
        x = np.asarray(c["x"], np.float32)
        h, w = x.shape
        Cn = 8                                             # the kernel works on groups of 4 channels: channel k carries (k+1) * x
        gain = np.arange(1, Cn + 1, dtype=np.float32)
        xd = dev(x[None, :, :, None] * gain)
        wd = dev(np.repeat(np.asarray(c["w"], np.float32)[:, :, None, None], Cn, axis=2))
        want = np.asarray(c["y"])
        y = torch.full((1,) + want.shape + (Cn,), np.nan, device=DEV)
        hip.check(lib.ams_k_depthwise3x3(P(xd), 1, h, w, Cn, P(wd), c["stride"], c["rate"], None, None, hip.ACT_NONE, P(y), stream()))
        np.testing.assert_allclose(y.cpu().numpy()[0], want[:, :, None] * gain, rtol=1e-6, atol=1e-5, err_msg=str((c["stride"], c["rate"])))


def test_stem_pad_normalise_conv(lib, V):
    c = V["stem"]
    frame = np.asarray(c["frame_u8"], np.uint8)[None]
    w = np.asarray(c["w_hwio"], np.float32)
    want = np.asarray(c["y"])
    for arr, code in ((frame, hip.DT_U8), (frame.astype(np.float32), hip.DT_F32)):
        fd = dev(arr, torch.as_tensor(arr).dtype)
        y = torch.full((1,) + want.shape, np.nan, device=DEV)
        hip.check(lib.ams_k_stem_conv(P(fd), code, 1, 3, 3, P(dev(w)), 32, None, None, hip.ACT_NONE, float(np.float32(c["pixel_scale"])), P(y), stream()))
        np.testing.assert_allclose(y.cpu().numpy()[0], want, rtol=0, atol=2e-5)


def test_upsample_argmax_ce_confusion(lib, V):
    """align-corners resize (incl. a 1x1 source), gather + first-maximum argmax, out-of-range labels, CE, confusion matrix."""
    h = V["head"]
    ci, nc = h["class_indices"], h["num_classes"]
    K = len(ci)
    cidx = (C.c_int32 * K)(*ci)
    # (a) identity-size "resize": four pixels with the hand-made logits rows, labels from the vector's teacher ids
    logits = np.asarray(h["logits"], np.float32).reshape(1, 2, 2, nc)
    teacher = np.asarray([13, 255, 2, 10], np.uint8).reshape(1, 2, 2)          # subset entries 5, ignored, 2, 3
    labels = torch.empty((1, 2, 2), dtype=torch.int32, device=DEV)
    conf = torch.empty(K * K, dtype=torch.int64, device=DEV)
    loss = torch.empty(2, dtype=torch.float64, device=DEV)
    hip.check(lib.ams_k_upsample_argmax(P(dev(logits)), 1, 2, 2, nc, cidx, K, 2, 2, P(devu8(teacher)), P(labels), P(conf), P(loss),
                                        stream()))
    assert labels.cpu().numpy().reshape(-1).tolist() == h["argmax_in_subset"]      # ties -> first maximum; classes outside the subset never win
    cm = conf.cpu().numpy().reshape(K, K)
    want_cm = np.zeros((K, K), np.int64)
    for t, p in ((5, 2), (2, 2), (3, 3)):                                         # rows = teacher, columns = student
        want_cm[t, p] += 1
    assert np.array_equal(cm, want_cm) and loss.cpu().numpy()[1] == 3
    # (b) teacher-id rule: every id of the vector, one pixel each
    ids = np.asarray(h["teacher_ids"], np.uint8).reshape(1, 1, -1)
    n = ids.size
    flat = np.zeros((1, 1, n, nc), np.float32)
    hip.check(lib.ams_k_upsample_argmax(P(dev(flat)), 1, 1, n, nc, cidx, K, 1, n, P(devu8(ids)),
                                        P(dev(np.zeros((1, 1, n), np.int32), torch.int32)), P(conf), P(loss), stream()))
    cm = conf.cpu().numpy().reshape(K, K)
    assert loss.cpu().numpy()[1] == sum(h["weight"])
    want_rows = np.zeros(K, np.int64)
    for t, wgt in zip(h["target_in_subset"], h["weight"]):
        want_rows[t] += wgt
    assert np.array_equal(cm.sum(axis=1), want_rows) and cm[:, 1:].sum() == 0     # all-equal logits: the student says entry 0 everywhere
    assert loss.cpu().numpy()[0] / loss.cpu().numpy()[1] == pytest.approx(np.log(K), rel=1e-6)
    # (c) CE value
    z = np.zeros((1, 1, 1, nc), np.float32)
    z[0, 0, 0, ci[:3]] = h["ce_logits"]
    z[0, 0, 0, ci[3:]] = -1e4
    t = torch.tensor([[[ci[h["ce_target"]]]]], dtype=torch.uint8, device=DEV)
    hip.check(lib.ams_k_upsample_argmax(P(dev(z)), 1, 1, 1, nc, cidx, K, 1, 1, P(t), P(dev(np.zeros((1, 1, 1), np.int32), torch.int32)), P(conf),
                                        P(loss), stream()))
    assert loss.cpu().numpy()[0] == pytest.approx(h["ce"], rel=1e-6)
    # (d) resize cases: class 0 carries the image, class 1 a threshold plane; the label map is (image < threshold)
    for c in V["resize_bilinear"]["cases"]:
        x = np.asarray(c["x"], np.float32)
        want = np.asarray(c["y"])
        thr = float(np.median(want)) + 1e-3
        lg = np.full((1,) + x.shape + (nc,), -1e4, np.float32)
        lg[0, :, :, 0] = x
        lg[0, :, :, 1] = thr
        lab = torch.empty((1, c["oh"], c["ow"]), dtype=torch.int32, device=DEV)
        tch = torch.zeros((1, c["oh"], c["ow"]), dtype=torch.uint8, device=DEV)          # teacher: class 0 everywhere
        hip.check(lib.ams_k_upsample_argmax(P(dev(lg)), 1, x.shape[0], x.shape[1], nc, cidx, K, c["oh"], c["ow"], P(tch), P(lab), P(conf), P(loss), stream()))
        assert np.array_equal(lab.cpu().numpy()[0], (want < thr).astype(np.int32)), c
        # the loss carries the interpolated VALUES: mean of log(1 + exp(thr - v)) over the output pixels
        ce = np.log1p(np.exp(thr - want)).mean()
        assert loss.cpu().numpy()[0] / loss.cpu().numpy()[1] == pytest.approx(ce, rel=2e-6), c


def test_adam_tf1_eps_outside_sqrt(lib, V):
    a = V["adam_tf1"]
    cs = a["cases"]
    n = len(cs)
    for i, c in enumerate(cs):                    # one launch per case: lr_t depends on t
        pad = lambda key: dev(np.asarray([cs[j][key] if j == i else 0.0 for j in range(n)] + [0.0] * (8 - n), np.float32))  # noqa: E731
        p, g, m, v = pad("w"), pad("g"), pad("m"), pad("v")
        hip.check(lib.ams_k_adam(P(p), P(g), P(m), P(v), None, 8, float(c["lr_t"]), a["beta1"], a["beta2"], a["eps"], stream()))
        assert float(p[i]) == pytest.approx(c["w_after"], rel=2e-7, abs=1e-9)
        assert float(m[i]) == pytest.approx(c["m_after"], rel=1e-6, abs=1e-12)
        assert float(v[i]) == pytest.approx(c["v_after"], rel=1e-6, abs=1e-20)
    tiny = cs[1]                                  # g = 1e-8: eps dominates the denominator only if it sits outside the root
    step_outside = tiny["lr_t"] * 0.0969346
    assert (tiny["w"] - tiny["w_after"]) == pytest.approx(step_outside, rel=1e-4)


def test_moving_averages_take_the_unbiased_batch_variance(lib):
    """FusedBatchNormV3 output 2 (Bessel-corrected) feeds AssignMovingAvg; normalisation uses the biased one (C.3).  Checked on
    the engine's first BN: the raw stem output comes from ams_k_stem_conv, its statistics from NumPy in f64."""
    H, B = 32, 2
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (B, H, 2 * H, 3), dtype=np.uint8)
    labels = rng.integers(0, 3, (B, H, 2 * H)).astype(np.uint8)
    eng = StudentEngine([0, 1, 2, 10, 11, 13], H, 2 * H, max_batch=B, trainable=True)
    eng.load_variables(W0)
    ho, wo = (H + 2) // 2, (2 * H + 2) // 2
    z = torch.empty((B, ho, wo, 32), device=DEV)
    hip.check(lib.ams_k_stem_conv(P(devu8(frames)), hip.DT_U8, B, H, 2 * H, P(dev(W0["MobilenetV2/Conv/weights:0"])), 32, None, None,
                                  hip.ACT_NONE, float(np.float32(S.PIXEL_SCALE)), P(z), stream()))
    zz = z.cpu().numpy().astype(np.float64).reshape(-1, 32)
    n = zz.shape[0]
    mean, var_b = zz.mean(0), zz.var(0)
    var_u = var_b * n / (n - 1)
    eng.train_step(frames, labels, 1e-3)
    got = eng.get_variables()
    omd = float(np.float32(1.0) - np.float32(S.BN_DECAY))
    mm0, mv0 = W0["MobilenetV2/Conv/BatchNorm/moving_mean:0"].astype(np.float64), W0["MobilenetV2/Conv/BatchNorm/moving_variance:0"].astype(np.float64)
    np.testing.assert_allclose(got["MobilenetV2/Conv/BatchNorm/moving_mean:0"], mm0 - (mm0 - mean) * omd, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(got["MobilenetV2/Conv/BatchNorm/moving_variance:0"], mv0 - (mv0 - var_u) * omd, rtol=2e-6)
    # n = 2 * 17 * 33 = 1122: biased and unbiased differ by 9e-4 relative, far above the tolerance — the biased one must NOT match
    wrong = mv0 - (mv0 - var_b) * omd
    assert np.abs(got["MobilenetV2/Conv/BatchNorm/moving_variance:0"] - wrong).max() > 10 * 2e-6 * np.abs(wrong).max()
    eng.close()
