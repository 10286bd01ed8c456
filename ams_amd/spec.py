"""Architecture of the AMS student (DeeplabV3 + MobileNetV2, output stride 16) as data.

The reference has no Python definition of the network: it is the TF1 MetaGraphDef
``checkpoints/deeplabv3_mobilenetv2_cityscapes/model.meta`` imported by
``create_student_v3`` (reference utils/graph_utils.py:350).  This module restates that
graph as a layer table; ``tests/test_spec.py`` pins it node by
node against ``tests/golden/student_graph_*.json`` (decoded from the reference file).

Everything downstream — the HIP engine's launch plan, the flat parameter arena, the
oracle, the ``.npy`` weight dict — is derived from this table, so variable NAMES and
their ORDER (= ``tf.trainable_variables()`` order = order of ``grad_masks_pl``,
``train_params`` and ``curr_mask``; reference SemanticNetwork.py:290-298) are contract.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

BN_EPS_BACKBONE = 0.0010000000474974513   # f32(1e-3), model.meta attr epsilon of MobilenetV2/* BNs
BN_EPS_HEAD = 1.0010000551119447e-05      # f32(1.001e-5), image_pooling / aspp0 / concat_projection BNs
BN_EPS_FROZEN = 1e-3                      # tf.layers.batch_normalization default used by the "_patch" twins
#                                           (reference utils/graph_utils.py:362-369, :52-76)
BN_DECAY = 0.8999999761581421             # f32(0.9), node */BatchNorm/Const_2
PIXEL_SCALE = 0.007843137718737125        # node mul_4/x  (f32(1/127.5))
PAD_VALUE = 127.5                         # nodes mul/x, mul_1/x


@dataclass(frozen=True)
class Layer:
    """One conv (+BN +activation) of the student."""
    idx: int              # 1-based row number of SURVEY.md Appendix A
    scope: str            # TF variable scope, e.g. "MobilenetV2/expanded_conv_3/depthwise"
    kind: str             # "conv" (dense, HWIO) | "dw" (depthwise, HWC1)
    k: int                # kernel size (3 or 1)
    cin: int
    cout: int
    stride: int
    rate: int
    bn_eps: Optional[float]   # None: no BN (logits layer has a bias instead)
    act: str              # "relu6" | "relu" | "none"
    residual_from: Optional[int] = None   # idx of the layer whose *block input* is added after BN (project layers)
    block: Optional[int] = None           # expanded_conv index (0..16) or None

    @property
    def weight_name(self) -> str:
        return self.scope + ("/depthwise_weights:0" if self.kind == "dw" else "/weights:0")

    @property
    def weight_shape(self) -> Tuple[int, ...]:
        if self.kind == "dw":
            return (self.k, self.k, self.cin, 1)
        return (self.k, self.k, self.cin, self.cout)


# (expansion t, output channels, stride, depthwise rate) for expanded_conv .. expanded_conv_16.
# Output stride 16: expanded_conv_13's nominal stride 2 is replaced by stride 1 and every later depthwise
# runs at rate 2 (model.meta nodes expanded_conv_14..16/depthwise/depthwise/SpaceToBatchND).
_BLOCKS = [
    (1, 16, 1, 1),
    (6, 24, 2, 1), (6, 24, 1, 1),
    (6, 32, 2, 1), (6, 32, 1, 1), (6, 32, 1, 1),
    (6, 64, 2, 1), (6, 64, 1, 1), (6, 64, 1, 1), (6, 64, 1, 1),
    (6, 96, 1, 1), (6, 96, 1, 1), (6, 96, 1, 1),
    (6, 160, 1, 1), (6, 160, 1, 2), (6, 160, 1, 2),
    (6, 320, 1, 2),
]

ASPP_DEPTH = 256


def build_layers(num_classes: int = 19) -> List[Layer]:
    layers: List[Layer] = []

    def add(**kw):
        layers.append(Layer(idx=len(layers) + 1, **kw))

    add(scope="MobilenetV2/Conv", kind="conv", k=3, cin=3, cout=32, stride=2, rate=1,
        bn_eps=BN_EPS_BACKBONE, act="relu6")
    c_prev = 32
    for b, (t, c, s, r) in enumerate(_BLOCKS):
        scope = "MobilenetV2/expanded_conv" + ("_%d" % b if b else "")
        c_mid = c_prev * t
        block_in_idx = len(layers)        # idx of the layer producing this block's input
        if t != 1:
            add(scope=scope + "/expand", kind="conv", k=1, cin=c_prev, cout=c_mid, stride=1, rate=1,
                bn_eps=BN_EPS_BACKBONE, act="relu6", block=b)
        add(scope=scope + "/depthwise", kind="dw", k=3, cin=c_mid, cout=c_mid, stride=s, rate=r,
            bn_eps=BN_EPS_BACKBONE, act="relu6", block=b)
        has_res = (s == 1 and c == c_prev)
        add(scope=scope + "/project", kind="conv", k=1, cin=c_mid, cout=c, stride=1, rate=1,
            bn_eps=BN_EPS_BACKBONE, act="none", residual_from=block_in_idx if has_res else None, block=b)
        c_prev = c
    add(scope="image_pooling", kind="conv", k=1, cin=c_prev, cout=ASPP_DEPTH, stride=1, rate=1,
        bn_eps=BN_EPS_HEAD, act="relu")
    add(scope="aspp0", kind="conv", k=1, cin=c_prev, cout=ASPP_DEPTH, stride=1, rate=1,
        bn_eps=BN_EPS_HEAD, act="relu")
    add(scope="concat_projection", kind="conv", k=1, cin=2 * ASPP_DEPTH, cout=ASPP_DEPTH, stride=1, rate=1,
        bn_eps=BN_EPS_HEAD, act="relu")
    add(scope="logits/semantic", kind="conv", k=1, cin=ASPP_DEPTH, cout=num_classes, stride=1, rate=1,
        bn_eps=None, act="none")
    return layers


@dataclass(frozen=True)
class Var:
    name: str
    shape: Tuple[int, ...]
    trainable: bool
    layer: int            # Layer.idx
    role: str             # "weights" | "gamma" | "beta" | "biases" | "moving_mean" | "moving_variance"
    offset: int = 0       # float offset inside its arena (trainable arena or statistics arena)

    @property
    def size(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


@dataclass
class StudentSpec:
    num_classes: int
    layers: List[Layer]
    trainable: List[Var]          # tf.trainable_variables() order
    stats: List[Var]              # moving_mean / moving_variance, layer order
    by_name: Dict[str, Var] = field(default_factory=dict)

    @property
    def n_trainable(self) -> int:
        return sum(v.size for v in self.trainable)

    @property
    def n_stats(self) -> int:
        return sum(v.size for v in self.stats)

    def layer(self, scope: str) -> Layer:
        for l in self.layers:
            if l.scope == scope:
                return l
        raise KeyError(scope)

    def var(self, name: str) -> Var:
        return self.by_name[name]

    def all_variable_names(self) -> List[str]:
        """Names in GraphDef (creation) order: per layer weights, gamma, beta, moving_mean, moving_variance."""
        out = []
        for l in self.layers:
            out.append(l.weight_name)
            if l.bn_eps is not None:
                for r in ("gamma", "beta", "moving_mean", "moving_variance"):
                    out.append("%s/BatchNorm/%s:0" % (l.scope, r))
            else:
                out.append(l.scope + "/biases:0")
        return out


def build_spec(num_classes: int = 19) -> StudentSpec:
    layers = build_layers(num_classes)
    trainable: List[Var] = []
    stats: List[Var] = []
    t_off = 0
    s_off = 0
    for l in layers:
        w = Var(l.weight_name, l.weight_shape, True, l.idx, "weights", t_off)
        trainable.append(w)
        t_off += w.size
        if l.bn_eps is not None:
            for role in ("gamma", "beta"):
                v = Var("%s/BatchNorm/%s:0" % (l.scope, role), (l.cout,), True, l.idx, role, t_off)
                trainable.append(v)
                t_off += v.size
            for role in ("moving_mean", "moving_variance"):
                v = Var("%s/BatchNorm/%s:0" % (l.scope, role), (l.cout,), False, l.idx, role, s_off)
                stats.append(v)
                s_off += v.size
        else:
            v = Var(l.scope + "/biases:0", (l.cout,), True, l.idx, "biases", t_off)
            trainable.append(v)
            t_off += v.size
    spec = StudentSpec(num_classes, layers, trainable, stats)
    spec.by_name = {v.name: v for v in trainable + stats}
    return spec


def same_pad(in_size: int, k: int, stride: int, rate: int) -> Tuple[int, int, int]:
    """TF 'SAME' padding: returns (out_size, pad_before, pad_after)."""
    out = -(-in_size // stride)
    eff = (k - 1) * rate + 1
    total = max((out - 1) * stride + eff - in_size, 0)
    return out, total // 2, total - total // 2


def feature_sizes(height: int, width: int, layers: Optional[List[Layer]] = None) -> List[Tuple[int, int]]:
    """Spatial output size of every layer for an (unpadded) input frame of height x width.

    The graph pads the frame by one row/column of 127.5 first (nodes concat, concat_1), so the
    backbone sees (height+1) x (width+1)."""
    layers = layers or build_layers()
    h, w = height + 1, width + 1
    sizes = []
    for l in layers:
        if l.scope == "image_pooling":
            sizes.append((1, 1))
            continue
        if l.scope in ("aspp0", "concat_projection", "logits/semantic"):
            sizes.append((h, w))
            continue
        h = same_pad(h, l.k, l.stride, l.rate)[0]
        w = same_pad(w, l.k, l.stride, l.rate)[0]
        sizes.append((h, w))
    return sizes


def macs_per_frame(height: int, width: int, num_classes: int = 19) -> int:
    layers = build_layers(num_classes)
    total = 0
    for l, (h, w) in zip(layers, feature_sizes(height, width, layers)):
        per_px = l.k * l.k * (l.cin if l.kind == "conv" else 1) * l.cout
        total += h * w * per_px
    return total


def activation_elements(height: int, width: int, num_classes: int = 19) -> Dict[str, int]:
    """Layer-wise algorithmic element counts (SURVEY.md §8 d4 convention): every conv reads its
    input once and writes its output once, residual operands and the pool input are read once."""
    layers = build_layers(num_classes)
    sizes = feature_sizes(height, width, layers)
    conv_in = conv_out = skip = pool = 0
    prev_hw = (height + 1, width + 1)
    for l, (h, w) in zip(layers, sizes):
        if l.scope == "image_pooling":
            pool = prev_hw[0] * prev_hw[1] * l.cin
            conv_in += l.cin
            conv_out += l.cout
            continue
        if l.idx == 1:
            conv_in += prev_hw[0] * prev_hw[1] * l.cin
        elif l.kind == "dw" and l.stride == 2:
            conv_in += prev_hw[0] * prev_hw[1] * l.cin
        else:
            ih, iw = (prev_hw if l.scope not in ("aspp0",) else prev_hw)
            conv_in += ih * iw * l.cin
        conv_out += h * w * l.cout
        if l.residual_from is not None:
            skip += h * w * l.cout
        prev_hw = (h, w)
    return {"conv_in": conv_in, "conv_out": conv_out, "skip": skip, "pool": pool,
            "total": conv_in + conv_out + skip + pool}
