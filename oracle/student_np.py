"""Independent NumPy restatement of the student forward pass, metric path and Adam/EMA update.
TEST INFRASTRUCTURE ONLY (see oracle/student_torch.py for the rules and the "parity unpinned" note).

Written tap-by-tap with explicit slicing (no conv library), so that it shares no code with
oracle/student_torch.py; tests/test_oracle_cpu.py requires the two to agree.  Cites:
model.meta nodes via SURVEY.md Appendix A, TF op semantics via Appendix C, and reference
utils/graph_utils.py:373-408 (class gather / argmax / CE), SemanticNetwork.py:96-115 (frozen metric
path), utils/graph_utils.py:52-76 + :362-369 (frozen BN, eps 1e-3).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from ams_amd import spec as S


def _pad_same(x: np.ndarray, k: int, stride: int, rate: int) -> np.ndarray:
    _, pt, pb = S.same_pad(x.shape[1], k, stride, rate)
    _, pl, pr = S.same_pad(x.shape[2], k, stride, rate)
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))


def conv3x3_dense(x, w, stride):
    """NHWC dense conv, weights HWIO, SAME."""
    b, h, wd, _ = x.shape
    oh = S.same_pad(h, 3, stride, 1)[0]
    ow = S.same_pad(wd, 3, stride, 1)[0]
    xp = _pad_same(x, 3, stride, 1)
    out = np.zeros((b, oh, ow, w.shape[3]), dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            patch = xp[:, i:i + (oh - 1) * stride + 1:stride, j:j + (ow - 1) * stride + 1:stride, :]
            out += patch @ w[i, j]
    return out


def depthwise3x3(x, w, stride, rate):
    """NHWC depthwise conv, weights [3,3,C,1], SAME; rate 2 == SpaceToBatchND/VALID/BatchToSpaceND."""
    b, h, wd, c = x.shape
    oh = S.same_pad(h, 3, stride, rate)[0]
    ow = S.same_pad(wd, 3, stride, rate)[0]
    xp = _pad_same(x, 3, stride, rate)
    out = np.zeros((b, oh, ow, c), dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            r0, c0 = i * rate, j * rate
            patch = xp[:, r0:r0 + (oh - 1) * stride + 1:stride, c0:c0 + (ow - 1) * stride + 1:stride, :]
            out += patch * w[i, j, :, 0]
    return out


def batch_norm(x, gamma, beta, mean, var, eps):
    return (x - mean) * (1.0 / np.sqrt(var + eps)) * gamma + beta


def preprocess(frames):
    """[B,H,W,3] 0..255 -> padded by one row / column of 127.5, times f32(1/127.5), minus 1 (nodes concat, concat_1, mul_4, sub_2)."""
    x = np.pad(frames, ((0, 0), (0, 1), (0, 1), (0, 0)), constant_values=S.PAD_VALUE)
    return x * frames.dtype.type(np.float32(S.PIXEL_SCALE)) - frames.dtype.type(1.0)


def batch_norm_train(x, gamma, beta, eps):
    """FusedBatchNormV3(is_training=True), NHWC: (y, batch mean, unbiased batch variance)."""
    n = x.shape[0] * x.shape[1] * x.shape[2]
    mu = x.mean(axis=(0, 1, 2))
    var = ((x - mu) ** 2).mean(axis=(0, 1, 2))
    return batch_norm(x, gamma, beta, mu, var, eps), mu, var * (n / max(n - 1, 1))


def resize_bilinear_align_corners(x, out_h, out_w):
    b, in_h, in_w, c = x.shape

    def taps(n_in, n_out):
        scale = np.float32((n_in - 1) / (n_out - 1)) if n_out > 1 else np.float32(0)
        src = np.arange(n_out, dtype=np.float32) * scale
        lo = np.floor(src).astype(np.int64)
        return lo, np.minimum(lo + 1, n_in - 1), (src - lo).astype(x.dtype)

    y0, y1, ty = taps(in_h, out_h)
    x0, x1, tx = taps(in_w, out_w)
    tx = tx[None, None, :, None]
    ty = ty[None, :, None, None]
    tl, tr = x[:, y0][:, :, x0], x[:, y0][:, :, x1]
    bl, br = x[:, y1][:, :, x0], x[:, y1][:, :, x1]
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return top + (bot - top) * ty


def forward_lowres(variables: Dict[str, np.ndarray], frames: np.ndarray, mode: str = "frozen",
                   num_classes: int = 19, dtype=np.float32, batch_stats: Optional[dict] = None) -> np.ndarray:
    spec = S.build_spec(num_classes)
    p = {k: np.asarray(v, dtype=dtype) for k, v in variables.items()}
    x = preprocess(np.asarray(frames, dtype=dtype))

    def bn_act(y, l):
        g, bta = p[l.scope + "/BatchNorm/gamma:0"], p[l.scope + "/BatchNorm/beta:0"]
        if mode == "frozen":
            y = batch_norm(y, g, bta, p[l.scope + "/BatchNorm/moving_mean:0"],
                           p[l.scope + "/BatchNorm/moving_variance:0"], dtype(S.BN_EPS_FROZEN))
        else:
            y, mu, var_unbiased = batch_norm_train(y, g, bta, dtype(l.bn_eps))
            if batch_stats is not None:
                batch_stats[l.scope] = (mu, var_unbiased)
        if l.act == "relu6":
            y = np.clip(y, 0, 6)
        elif l.act == "relu":
            y = np.maximum(y, 0)
        return y

    outs = {0: x}
    backbone = [l for l in spec.layers if l.scope.startswith("MobilenetV2")]
    for l in backbone:
        w = p[l.weight_name]
        xin = outs[l.idx - 1]
        if l.kind == "dw":
            y = depthwise3x3(xin, w, l.stride, l.rate)
        elif l.k == 3:
            y = conv3x3_dense(xin, w, l.stride)
        else:
            y = xin @ w[0, 0]
        y = bn_act(y, l)
        if l.residual_from is not None:
            y = y + outs[l.residual_from]
        outs[l.idx] = y
    feat = outs[backbone[-1].idx]
    lp, la, lc, ll = spec.layers[-4:]
    pool = bn_act(feat.mean(axis=(1, 2), keepdims=True) @ p[lp.weight_name][0, 0], lp)
    pool = np.broadcast_to(pool, feat.shape[:3] + (pool.shape[-1],))
    aspp = bn_act(feat @ p[la.weight_name][0, 0], la)
    proj = bn_act(np.concatenate([pool, aspp], axis=-1) @ p[lc.weight_name][0, 0], lc)
    return proj @ p[ll.weight_name][0, 0] + p[ll.scope + "/biases:0"]


def predict_with_metric(variables, frames, labels_teacher, class_indices: Sequence[int], mode="frozen",
                        num_classes: int = 19, dtype=np.float32):
    """(labels int32 [B,H,W], conf_mat f64 [K,K], loss) — frozen metric path of SemanticNetwork.py:96-115."""
    ci = np.asarray(class_indices)
    k = len(ci)
    low = forward_lowres(variables, frames, mode, num_classes, dtype)
    full = resize_bilinear_align_corners(low, frames.shape[1], frames.shape[2])
    z = full[..., ci]
    pred = gather_argmax(full, ci)
    target, weight, sel = label_targets(labels_teacher, ci, num_classes, dtype)
    cm = confusion(target, pred, weight, k)
    pixel = softmax_ce(z, sel)
    valid = weight > 0
    loss = float(pixel[valid].mean()) if valid.any() else float("nan")
    return pred, cm, loss


def gather_argmax(logits, class_indices):
    """tf.gather(axis=-1) then tf.argmax: index of the FIRST maximum among the selected classes, int32."""
    return np.argmax(logits[..., np.asarray(class_indices)], axis=-1).astype(np.int32)


def label_targets(labels_teacher, class_indices, num_classes=19, dtype=np.float32):
    """cast -> one_hot(num_classes) (ids outside 0..num_classes-1: all-zero row) -> gather -> (argmax, row sum, gathered rows)."""
    lab = np.asarray(labels_teacher).astype(np.float32).astype(np.int32)
    onehot = (lab[..., None] == np.arange(num_classes)).astype(dtype)
    sel = onehot[..., np.asarray(class_indices)]
    return np.argmax(sel, axis=-1), sel.sum(axis=-1), sel


def softmax_ce(z, onehot):
    """softmax_cross_entropy_with_logits, max-subtracted: logsumexp(z) - sum(onehot * z)."""
    zmax = z.max(axis=-1, keepdims=True)
    return np.log(np.exp(z - zmax).sum(axis=-1)) + zmax[..., 0] - (z * onehot).sum(axis=-1)


def confusion(target, pred, weight, k):
    """tf.metrics.mean_iou's matrix: [label, prediction] += weight, float64."""
    cm = np.zeros((k, k), dtype=np.float64)
    np.add.at(cm, (np.asarray(target).reshape(-1), np.asarray(pred).reshape(-1)), np.asarray(weight).reshape(-1).astype(np.float64))
    return cm


def adam_step(w, g, m, v, lr, beta1_power, beta2_power):
    """TensorFlow's ApplyAdam functor (core/kernels/training_ops.cc, Appendix C.10) with the graph's f32 hyper-parameter values.
    Returns (w, m, v)."""
    one_minus_b1 = float(np.float32(1.0) - np.float32(0.9))
    one_minus_b2 = float(np.float32(1.0) - np.float32(0.999))
    eps = float(np.float32(1e-8))
    alpha = lr * np.sqrt(1.0 - beta2_power) / (1.0 - beta1_power)
    m = m + (g - m) * one_minus_b1
    v = v + (g * g - v) * one_minus_b2
    return w - (m * alpha) / (np.sqrt(v) + eps), m, v


def ema_update(moving, stat, decay=np.float32(S.BN_DECAY)):
    """AssignMovingAvg: moving -= (moving - stat) * (1 - decay), all in f32."""
    return moving - (moving - stat) * (np.float32(1.0) - np.float32(decay))
