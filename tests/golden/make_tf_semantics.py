#!/usr/bin/env python3
"""Generate tests/golden/tf_semantics.json: small vectors for every TensorFlow 1.15 op rule the oracle restates
(SURVEY.md Appendix C), each derived HERE from the published definition with scalar Python arithmetic — no oracle code,
no NumPy convolution, no product code.  Cases marked "hand" carry values worked out on paper and typed in; the scalar
loops must reproduce them (asserted below), so a slip in the loops cannot silently become the expectation.

These vectors do not replace TensorFlow (absent here and on the GPU box): parity stays "partial".  They pin
the RULES — asymmetric SAME padding on even sizes, unbiased variance in the moving average, align-corners resize incl. a
1x1 source, first-maximum argmax, out-of-range one-hot rows, TF1 Adam with eps outside the square root — and both oracles,
the graph executor and the HIP kernels are run against them (tests/test_tf_semantics.py, tests/test_gpu_tf_semantics.py).

Published definitions used:
  SAME padding     tf.nn.convolution docs ("SAME": out = ceil(in / stride); pad_along = max((out-1)*stride + k_eff - in, 0);
                   pad_before = pad_along // 2, the odd unit goes after), k_eff = (k-1)*rate + 1
  conv / depthwise tf.nn.conv2d / tf.nn.depthwise_conv2d_native docs (cross-correlation, NHWC, HWIO / HWC1 filters)
  FusedBatchNorm   tf.nn.fused_batch_norm docs + core/kernels/fused_batch_norm_op.cc: normalise with the biased batch
                   variance, emit the Bessel-corrected variance (n / (n-1)) as output 2; moving -= (moving - stat)*(1-decay)
  ResizeBilinear   core/kernels/resize_bilinear_op.cc + image_resizer_state.h: align_corners scale = (in-1)/(out-1) (f32),
                   src = dst*scale, lower = floor(src), upper = min(lower+1, in-1), lerp = src - lower;
                   top = tl + (tr-tl)*x_lerp, bottom likewise, out = top + (bottom-top)*y_lerp
  argmax           tf.math.argmax docs: smallest index among ties
  one_hot          tf.one_hot docs: an index outside [0, depth) gives an all-off row
  softmax CE       tf.nn.softmax_cross_entropy_with_logits: -sum(labels * log_softmax(logits))
  mean_iou         tf.metrics.mean_iou: confusion[label, prediction] += weight
  Adam             tf.train.AdamOptimizer docs: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
                   variable -= lr_t * m / (sqrt(v) + epsilon)
"""
import json
import math
import struct
from pathlib import Path

OUT = Path(__file__).resolve().parent / "tf_semantics.json"


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def same_pad(n_in, k, stride, rate):
    out = -(-n_in // stride)
    k_eff = (k - 1) * rate + 1
    along = max((out - 1) * stride + k_eff - n_in, 0)
    return out, along // 2, along - along // 2


def depthwise(x, w, stride, rate):
    """x [H][W], w [3][3] -> SAME output, one channel."""
    h, wd = len(x), len(x[0])
    oh, pt, _ = same_pad(h, 3, stride, rate)
    ow, pl, _ = same_pad(wd, 3, stride, rate)
    out = [[0.0] * ow for _ in range(oh)]
    for y in range(oh):
        for xx in range(ow):
            acc = 0.0
            for i in range(3):
                for j in range(3):
                    sy, sx = y * stride + i * rate - pt, xx * stride + j * rate - pl
                    if 0 <= sy < h and 0 <= sx < wd:
                        acc += x[sy][sx] * w[i][j]
            out[y][xx] = acc
    return out


def resize_1d_taps(n_in, n_out):
    scale = f32((n_in - 1) / (n_out - 1)) if n_out > 1 else 0.0
    taps = []
    for d in range(n_out):
        src = f32(d * scale)
        lo = int(math.floor(src))
        taps.append((lo, min(lo + 1, n_in - 1), f32(src - lo)))
    return taps


def resize(img, oh, ow):
    """img [h][w] -> [oh][ow], align_corners=True."""
    ty, tx = resize_1d_taps(len(img), oh), resize_1d_taps(len(img[0]), ow)
    out = []
    for (y0, y1, fy) in ty:
        row = []
        for (x0, x1, fx) in tx:
            top = img[y0][x0] + (img[y0][x1] - img[y0][x0]) * fx
            bot = img[y1][x0] + (img[y1][x1] - img[y1][x0]) * fx
            row.append(top + (bot - top) * fy)
        out.append(row)
    return out


def main():
    v = {"_about": "hand-derived TF 1.15 op-semantics vectors; generator tests/golden/make_tf_semantics.py (see its docstring for "
                   "the published definition behind each case)"}

    # ---- 1. SAME padding ------------------------------------------------------------------------------------------
    hand = [  # (in, k, stride, rate) -> (out, before, after), worked on paper
        ((4, 3, 2, 1), (2, 0, 1)), ((5, 3, 2, 1), (3, 1, 1)), ((6, 3, 2, 1), (3, 0, 1)), ((7, 3, 2, 1), (4, 1, 1)),
        ((4, 3, 1, 1), (4, 1, 1)), ((4, 3, 1, 2), (4, 2, 2)), ((513, 3, 2, 1), (257, 1, 1)), ((300, 3, 2, 1), (150, 0, 1)),
        ((33, 3, 1, 2), (33, 2, 2)), ((2, 3, 2, 1), (1, 0, 1)), ((1, 3, 1, 2), (1, 2, 2))]
    for args, want in hand:
        assert same_pad(*args) == want, (args, same_pad(*args), want)
    v["same_pad"] = {"rule": "C.1", "cases": [{"in": a[0], "k": a[1], "stride": a[2], "rate": a[3], "out": w[0], "before": w[1], "after": w[2]}
                                              for a, w in hand]}

    # ---- 2. depthwise 3x3 on an even size: the extra padding is at the bottom / right ------------------------------
    x44 = [[float(4 * r + c) for c in range(4)] for r in range(4)]
    ones = [[1.0] * 3 for _ in range(3)]
    ramp = [[float(1 + 3 * i + j) for j in range(3)] for i in range(3)]       # 1..9, asymmetric: catches flipped taps
    assert depthwise(x44, ones, 2, 1) == [[45.0, 39.0], [66.0, 50.0]]        # hand: rows 0-2 / 2-3, cols 0-2 / 2-3
    x56 = [[float((7 * r + 3 * c) % 11) - 4.0 for c in range(6)] for r in range(5)]
    v["depthwise"] = {"rule": "C.1, C.2", "cases": [
        {"x": x44, "w": ones, "stride": 2, "rate": 1, "y": depthwise(x44, ones, 2, 1), "hand": True},
        {"x": x44, "w": ramp, "stride": 2, "rate": 1, "y": depthwise(x44, ramp, 2, 1)},
        {"x": x44, "w": ramp, "stride": 1, "rate": 1, "y": depthwise(x44, ramp, 1, 1)},
        {"x": x44, "w": ramp, "stride": 1, "rate": 2, "y": depthwise(x44, ramp, 1, 2)},
        {"x": x56, "w": ramp, "stride": 2, "rate": 1, "y": depthwise(x56, ramp, 2, 1)},
        {"x": x56, "w": ramp, "stride": 1, "rate": 2, "y": depthwise(x56, ramp, 1, 2)}]}

    # ---- 3. the stem: pad one row / column of 127.5, x * f32(1/127.5) - 1, dense 3x3 stride 2 SAME ------------------
    # frame 3x3x3 -> padded 4x4 (even) -> SAME pads (0, 1): the zero padding sits BELOW / RIGHT of the 127.5 border
    scale = 0.007843137718737125
    frame = [[[(37 * r + 11 * c + 5 * ch) % 256 for ch in range(3)] for c in range(3)] for r in range(3)]
    cout = 32                                       # the real layer-1 width: the reference graph prefix runs on this vector as is
    wst = [[[[((i * 3 + j) * 3 + ci) * 0.03125 - 0.4 + 0.015625 * co for co in range(cout)] for ci in range(3)] for j in range(3)] for i in range(3)]
    padded = [[[float(frame[r][c][ch]) if r < 3 and c < 3 else 127.5 for ch in range(3)] for c in range(4)] for r in range(4)]
    norm = [[[f32(f32(padded[r][c][ch] * scale) - 1.0) for ch in range(3)] for c in range(4)] for r in range(4)]
    oh, pt, _ = same_pad(4, 3, 2, 1)
    stem = [[[0.0] * cout for _ in range(oh)] for _ in range(oh)]
    for y in range(oh):
        for xx in range(oh):
            for co in range(cout):
                acc = 0.0
                for i in range(3):
                    for j in range(3):
                        sy, sx = 2 * y + i - pt, 2 * xx + j - pt
                        if 0 <= sy < 4 and 0 <= sx < 4:
                            for ci in range(3):
                                acc += norm[sy][sx][ci] * wst[i][j][ci][co]
                stem[y][xx][co] = acc
    assert (oh, pt) == (2, 0)
    v["stem"] = {"rule": "C.1 + nodes concat, concat_1, mul_4, sub_2", "frame_u8": frame, "w_hwio": wst, "pixel_scale": scale, "y": stem}

    # ---- 4. FusedBatchNormV3 in training mode and its moving averages ----------------------------------------------
    xs = [1.0, 2.0, 3.0, 4.0]
    gamma, beta, eps, decay = 2.0, 0.5, 0.0010000000474974513, 0.8999999761581421
    mean = sum(xs) / 4
    var_b = sum((t - mean) ** 2 for t in xs) / 4
    var_u = var_b * 4 / 3
    assert (mean, var_b) == (2.5, 1.25) and abs(var_u - 5.0 / 3.0) < 1e-15           # hand
    one_minus = f32(1.0 - f32(decay))
    v["batch_norm_train"] = {"rule": "C.3", "x": xs, "gamma": gamma, "beta": beta, "eps": eps, "decay": decay,
                             "y": [(t - mean) / math.sqrt(var_b + eps) * gamma + beta for t in xs],
                             "batch_mean": mean, "batch_var_biased": var_b, "batch_var_unbiased": var_u,
                             "moving_mean_before": 0.3, "moving_var_before": 1.0,
                             "moving_mean_after": 0.3 - (0.3 - mean) * one_minus, "moving_var_after": 1.0 - (1.0 - var_u) * one_minus}
    v["batch_norm_frozen"] = {"rule": "C.3 + graph_utils.py:52-76 (eps 1e-3 whatever the layer trained with)", "x": xs, "gamma": gamma,
                              "beta": beta, "moving_mean": 0.3, "moving_var": 1.7, "eps": 1e-3,
                              "y": [(t - 0.3) / math.sqrt(1.7 + 1e-3) * gamma + beta for t in xs]}

    # ---- 5. ResizeBilinear(align_corners=True) ---------------------------------------------------------------------
    assert resize([[0.0, 1.0], [2.0, 3.0]], 3, 3) == [[0.0, 0.5, 1.0], [1.0, 1.5, 2.0], [2.0, 2.5, 3.0]]       # hand
    assert resize([[0.0, 10.0, 20.0]], 1, 5) == [[0.0, 5.0, 10.0, 15.0, 20.0]]                               # hand
    assert resize([[7.0]], 2, 3) == [[7.0] * 3] * 2                                                          # 1x1 source broadcasts
    img = [[1.0, -2.0, 4.0], [0.5, 3.0, -1.0]]
    v["resize_bilinear"] = {"rule": "C.5", "cases": [
        {"x": [[0.0, 1.0], [2.0, 3.0]], "oh": 3, "ow": 3, "y": resize([[0.0, 1.0], [2.0, 3.0]], 3, 3), "hand": True},
        {"x": [[0.0, 10.0, 20.0]], "oh": 1, "ow": 5, "y": resize([[0.0, 10.0, 20.0]], 1, 5), "hand": True},
        {"x": [[7.0]], "oh": 2, "ow": 3, "y": resize([[7.0]], 2, 3), "hand": True},
        {"x": [[0.0, 10.0]], "oh": 1, "ow": 4, "y": resize([[0.0, 10.0]], 1, 4)},
        {"x": img, "oh": 4, "ow": 7, "y": resize(img, 4, 7)},
        {"x": img, "oh": 2, "ow": 3, "y": resize(img, 2, 3)}]}

    # ---- 6-9. gather + first-maximum argmax, label path, softmax CE, confusion matrix --------------------------------
    ci = [0, 1, 2, 10, 11, 13]
    logits19 = [[0.0] * 19 for _ in range(4)]
    logits19[0][2] = logits19[0][11] = 3.0            # tie between subset entries 2 and 4 -> 2
    logits19[1][5] = 9.0                              # the largest logit is NOT in the subset: ignored by the gather
    logits19[1][13] = 1.0                             # -> subset entry 5
    logits19[2][0] = logits19[2][1] = logits19[2][13] = -1.0    # all subset entries: -1, -1, 0, 0, 0, -1 -> first 0 is entry 2
    logits19[3][10] = 2.5                             # -> entry 3
    preds = [2, 5, 2, 3]                              # hand
    labels = [2, 5, 255, 13, 19, 200, 0, 10]          # class ids as the teacher PNG holds them
    targets = [2, 0, 0, 5, 0, 0, 0, 3]                # hand: argmax of the gathered one-hot row (all-zero row -> 0)
    weights = [1, 0, 0, 1, 0, 0, 1, 1]                # hand: 5 is a valid id outside the subset, 19 / 200 / 255 are out of range
    z = [1.0, 2.0, 3.0]
    lse = math.log(sum(math.exp(t) for t in z))
    assert abs((lse - z[2]) - 0.40760596444438) < 1e-12                                            # hand
    v["head"] = {"rule": "C.6, C.7, C.8, C.9", "class_indices": ci, "num_classes": 19, "logits": logits19, "argmax_in_subset": preds,
                 "teacher_ids": labels, "target_in_subset": targets, "weight": weights,
                 "ce_logits": z, "ce_target": 2, "ce": lse - z[2],
                 "confusion": {"teacher_in_subset": [0, 0, 2, 5, 5], "pred_in_subset": [0, 1, 2, 5, 0], "weight": [1, 1, 0, 1, 1], "k": 6,
                               "nonzero": [[0, 0, 1], [0, 1, 1], [5, 5, 1], [5, 0, 1]]}}

    # ---- 10. Adam, TF1 form -------------------------------------------------------------------------------------------
    # hyper-parameters are f32 tensors in the graph; ApplyAdam (core/kernels/training_ops.cc) forms (1 - beta) in T = float:
    #   alpha = lr*sqrt(1-b2^t)/(1-b1^t);  m += (g-m)*(1-b1);  v += (g*g-v)*(1-b2);  var -= (m*alpha)/(sqrt(v)+eps)
    b1, b2, eps = f32(0.9), f32(0.999), f32(1e-8)
    omb1, omb2 = f32(1.0 - b1), f32(1.0 - b2)
    assert abs(omb2 - 0.00099998713) < 1e-11           # hand: f32(0.999) = 0.99900001287..., so 1 - beta2 is NOT 0.001

    def adam(w, g, m, vv, t, lr):
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m2 = m + (g - m) * omb1
        v2 = vv + (g * g - vv) * omb2
        return w - (m2 * lr_t) / (math.sqrt(v2) + eps), m2, v2, lr_t

    cases = []
    for (w, g, m, vv, t, lr) in [(1.0, 0.5, 0.0, 0.0, 1, 1e-3), (1.0, 1e-8, 0.0, 0.0, 1, 1e-3), (-0.25, -0.125, 0.02, 3e-4, 2, 1e-3),
                                 (2.0, 0.0, 0.0, 0.0, 1, 1e-3), (0.75, 0.3, -0.01, 1e-5, 7, 5e-4)]:
        w2, m2, v2, lr_t = adam(w, g, m, vv, t, lr)
        cases.append({"w": w, "g": g, "m": m, "v": vv, "t": t, "lr": lr, "lr_t": lr_t, "w_after": w2, "m_after": m2, "v_after": v2})
    # hand: first step with a sizeable gradient moves by ~lr (m/sqrt(v) = 0.05/0.0158114 = 3.1623, lr_t = lr*0.31623)
    assert abs((1.0 - cases[0]["w_after"]) - 1e-3) < 1e-8
    # hand: with g = 1e-8 eps dominates the denominator: step = lr_t * 1e-9 / (3.1623e-10 + 1e-8) = lr_t * 0.096935
    assert abs((1.0 - cases[1]["w_after"]) / cases[1]["lr_t"] - 0.0969346) < 2e-6
    v["adam_tf1"] = {"rule": "C.10", "beta1": b1, "beta2": b2, "eps": eps, "one_minus_beta1": omb1, "one_minus_beta2": omb2, "cases": cases,
                     "note": "case 2 separates eps OUTSIDE the square root (step 0.0969*lr_t) from eps inside it (1e-5*lr_t)"}

    # ---- 4b. activations ---------------------------------------------------------------------------------------------
    v["relu6"] = {"rule": "C.4", "x": [-1.0, 0.0, 0.5, 6.0, 7.5], "y": [0.0, 0.0, 0.5, 6.0, 6.0]}

    OUT.write_text(json.dumps(v, indent=1, sort_keys=True))
    print(OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
