"""Executes the reference's own forward graph, node by node, in NumPy.  TEST INFRASTRUCTURE ONLY.

Input: ``tests/golden/student_program_<ckpt>.json`` — the ancestor nodes of ``student_logits`` decoded from
``/root/reference/checkpoints/<ckpt>/model.meta`` (generator: tests/golden/make_graph_program.py; the graph
``create_student_v3`` imports at reference utils/graph_utils.py:350 and binds at :353-358).  Every op that appears there
is restated below from the published TensorFlow 1.15 op definitions: 24 op types, shape
arithmetic included (``Shape`` / ``StridedSlice`` / ``Pack`` / ``FloorMod`` compute the SpaceToBatchND paddings and the
resize targets exactly as the graph does, for any frame size).

What this buys: the WIRING the oracle follows is the reference's file, not ``ams_amd/spec.py``.  The two hand-written
oracles (student_torch.py, student_np.py), which share the product's layer table, are checked against this executor
(tests/test_graph_interp.py), and so is the HIP engine (tests/test_gpu_graph_parity.py).

PARITY UNPINNED still applies to the op SEMANTICS (TensorFlow is not installable here): see student_torch.py's header
and tests/golden/tf_semantics.json for the hand-derived vectors that pin each rule.

Modes: ``train`` runs FusedBatchNormV3 as the file says (is_training=True: batch statistics, output 1 = mean, output
2 = unbiased variance); ``frozen`` applies what ``convert_batchnorms`` rewires before a model is shipped (reference
utils/graph_utils.py:52-76, :362-369): moving statistics and eps = 1e-3 for every BN.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Optional

import numpy as np

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"
_TF_DTYPES = {1: np.float32, 3: np.int32, 9: np.int64}
BN_EPS_FROZEN = 1e-3        # tf.layers.batch_normalization default taken by the "_patch" twins (graph_utils.py:362-369)


def load_program(tag: str = "cityscapes") -> dict:
    return json.loads((GOLDEN / ("student_program_%s.json" % tag)).read_text())


def _same_pads(size: int, k_eff: int, stride: int):
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    return out, total // 2, total - total // 2          # the odd unit goes after (bottom / right)


def _conv_windows(x, kh, kw, strides, dilations, padding):
    """Yield (i, j, window) for every filter tap of an NHWC conv; window is [B, out_h, out_w, C]."""
    sh, sw = strides[1], strides[2]
    dh, dw = dilations[1], dilations[2]
    h, w = x.shape[1], x.shape[2]
    keh, kew = (kh - 1) * dh + 1, (kw - 1) * dw + 1
    if padding == "SAME":
        oh, pt, pb = _same_pads(h, keh, sh)
        ow, pl, pr = _same_pads(w, kew, sw)
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    elif padding == "VALID":
        oh, ow = (h - keh) // sh + 1, (w - kew) // sw + 1
    else:
        raise ValueError(padding)
    for i in range(kh):
        for j in range(kw):
            yield i, j, x[:, i * dh: i * dh + (oh - 1) * sh + 1: sh, j * dw: j * dw + (ow - 1) * sw + 1: sw, :]


def _strided_slice(x, begin, end, strides, n):
    x = np.asarray(x)
    idx = []
    for d in range(len(begin)):
        if n["ellipsis_mask"] or n["new_axis_mask"]:
            raise NotImplementedError("ellipsis / new-axis masks do not occur in the student graph")
        if (n["shrink_axis_mask"] >> d) & 1:
            idx.append(int(begin[d]))
            continue
        b = None if (n["begin_mask"] >> d) & 1 else int(begin[d])
        e = None if (n["end_mask"] >> d) & 1 else int(end[d])
        idx.append(slice(b, e, int(strides[d])))
    return x[tuple(idx)]


def _resize_bilinear(x, size, align_corners, half_pixel_centers):
    assert align_corners and not half_pixel_centers, "the student graph only resizes with align_corners=True"
    oh, ow = int(size[0]), int(size[1])
    ih, iw = x.shape[1], x.shape[2]

    def taps(n_in, n_out):
        scale = np.float32((n_in - 1) / (n_out - 1)) if n_out > 1 else np.float32(0)
        src = np.arange(n_out, dtype=np.float32) * scale
        lo = np.floor(src).astype(np.int64)
        return lo, np.minimum(lo + 1, n_in - 1), (src - lo.astype(np.float32)).astype(x.dtype)

    y0, y1, fy = taps(ih, oh)
    x0, x1, fx = taps(iw, ow)
    fx = fx.reshape(1, 1, -1, 1)
    fy = fy.reshape(1, -1, 1, 1)
    r0, r1 = x[:, y0], x[:, y1]
    top = r0[:, :, x0] + (r0[:, :, x1] - r0[:, :, x0]) * fx
    bot = r1[:, :, x0] + (r1[:, :, x1] - r1[:, :, x0]) * fx
    return top + (bot - top) * fy


class GraphExecutor:
    def __init__(self, program: dict, variables: Dict[str, np.ndarray], dtype=np.float64):
        self.nodes = program["nodes"]
        self.output = program["output"]
        self.feed = program["feed"]
        self.dtype = dtype
        self.variables = {k: np.asarray(v, dtype=dtype) for k, v in variables.items()}

    def run(self, frames, mode: str = "frozen", fetch: Optional[str] = None, taps: Optional[dict] = None) -> np.ndarray:
        """frames [B,H,W,3] (0..255) -> value of ``fetch`` (default: student_logits [B,H,W,num_classes]).
        ``taps``: optional dict receiving the (mean, unbiased variance) outputs of every FusedBatchNormV3 by node name."""
        assert mode in ("frozen", "train")
        dt = self.dtype
        val: Dict[str, tuple] = {}

        def get(ref):
            return val[ref[0]][ref[1]]

        for n in self.nodes:
            op, name = n["op"], n["name"]
            if name == self.feed:
                val[name] = (np.asarray(frames, dtype=dt),)
                continue
            if op == "QueueDequeueV2":
                val[name] = (None, None)             # only reached through `features`, which is fed
                continue
            if op == "VariableV2":
                v = self.variables[name + ":0"]
                assert list(v.shape) == n["shape"], name
                val[name] = (v,)
                continue
            a = [get(r) for r in n.get("inputs", [])]
            if op == "Const":
                arr = np.asarray(n["value"], dtype=_TF_DTYPES[n["dtype"]])
                shape = tuple(n["shape"])
                if arr.size == 1 and int(np.prod(shape, dtype=np.int64)) != 1:
                    arr = np.full(shape, arr.reshape(-1)[0])
                out = arr.reshape(shape)
                if out.dtype == np.float32:
                    out = out.astype(dt)                 # scalars like 1/127.5 keep their f32 VALUE, widened
            elif op in ("Identity", "StopGradient"):
                out = a[0]
            elif op == "Shape":
                out = np.asarray(a[0].shape, dtype=np.int32)
            elif op == "StridedSlice":
                out = _strided_slice(a[0], np.atleast_1d(a[1]), np.atleast_1d(a[2]), np.atleast_1d(a[3]), n)
            elif op == "Pack":
                out = np.stack([np.asarray(t) for t in a], axis=n.get("axis", 0))
            elif op == "Fill":
                out = np.full(tuple(int(d) for d in a[0]), a[1], dtype=np.asarray(a[1]).dtype)
            elif op == "Mul":
                out = a[0] * a[1]
            elif op == "Sub":
                out = a[0] - a[1]
            elif op == "AddV2":
                out = a[0] + a[1]
            elif op == "FloorMod":
                out = np.mod(a[0], a[1])
            elif op == "Cast":
                tgt = _TF_DTYPES[n["DstT"]]
                out = np.asarray(a[0]).astype(dt if tgt == np.float32 else tgt)      # float -> int truncates toward zero
            elif op == "ConcatV2":
                out = np.concatenate([np.asarray(t) for t in a[:-1]], axis=int(a[-1]))
            elif op == "PadV2":
                out = np.pad(a[0], [tuple(int(v) for v in p) for p in a[1]], constant_values=a[2])
            elif op == "Conv2D":
                w = a[1]
                out = 0
                for i, j, win in _conv_windows(a[0], w.shape[0], w.shape[1], n["strides"], n["dilations"], n["padding"]):
                    out = out + win @ w[i, j]
            elif op == "DepthwiseConv2dNative":
                w = a[1]
                assert w.shape[3] == 1
                out = 0
                for i, j, win in _conv_windows(a[0], w.shape[0], w.shape[1], n["strides"], n["dilations"], n["padding"]):
                    out = out + win * w[i, j, :, 0]
            elif op == "FusedBatchNormV3":
                x, gamma, beta = a[0], a[1], a[2]
                scope = name.rsplit("/FusedBatchNormV3", 1)[0]
                if mode == "train":
                    assert n["is_training"] is True
                    cnt = x.shape[0] * x.shape[1] * x.shape[2]
                    mean = x.mean(axis=(0, 1, 2))
                    var = ((x - mean) ** 2).mean(axis=(0, 1, 2))
                    y = (x - mean) * (1.0 / np.sqrt(var + dt(n["epsilon"]))) * gamma + beta
                    stats = (mean, var * (cnt / max(cnt - 1, 1)))
                    if taps is not None:
                        taps[name] = stats
                    val[name] = (y,) + stats
                    continue
                mean = self.variables[scope + "/moving_mean:0"]
                var = self.variables[scope + "/moving_variance:0"]
                out = (x - mean) * (1.0 / np.sqrt(var + dt(BN_EPS_FROZEN))) * gamma + beta
            elif op == "Relu6":
                out = np.minimum(np.maximum(a[0], 0), 6)
            elif op == "Relu":
                out = np.maximum(a[0], 0)
            elif op == "SpaceToBatchND":
                x, block, pads = a[0], [int(v) for v in a[1]], np.asarray(a[2])
                x = np.pad(x, ((0, 0), tuple(pads[0]), tuple(pads[1]), (0, 0)))
                b, h, w, c = x.shape
                x = x.reshape(b, h // block[0], block[0], w // block[1], block[1], c)
                out = x.transpose(2, 4, 0, 1, 3, 5).reshape(block[0] * block[1] * b, h // block[0], w // block[1], c)
            elif op == "BatchToSpaceND":
                x, block, crops = a[0], [int(v) for v in a[1]], np.asarray(a[2])
                nb, h, w, c = x.shape
                b = nb // (block[0] * block[1])
                x = x.reshape(block[0], block[1], b, h, w, c).transpose(2, 3, 0, 4, 1, 5).reshape(b, h * block[0], w * block[1], c)
                out = x[:, crops[0][0]: x.shape[1] - crops[0][1], crops[1][0]: x.shape[2] - crops[1][1], :]
            elif op == "Mean":
                out = a[0].mean(axis=tuple(int(v) for v in np.atleast_1d(a[1])), keepdims=bool(n["keep_dims"]))
            elif op == "ResizeBilinear":
                out = _resize_bilinear(a[0], a[1], n["align_corners"], n.get("half_pixel_centers", False))
            elif op == "BiasAdd":
                out = a[0] + a[1]
            else:
                raise NotImplementedError("op %s (node %s) is not part of the student forward graph" % (op, name))
            val[name] = (out,)
            if fetch is not None and name == fetch:
                return out
        return val[self.output][0]
