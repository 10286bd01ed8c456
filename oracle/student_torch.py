"""CPU oracle (PyTorch-CPU, f32 or f64) for the AMS student hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``ams_amd``) never does and fails loudly when its HIP library is missing.

PARITY UNPINNED: the arithmetic of this path lives in TensorFlow 1.15.0 (third party, not under
/root/reference, not installable here), the reference ships no tests, no golden vectors and no weights
(SURVEY.md §8 c1-c3).  This file restates, op by op,
  * the student graph of checkpoints/deeplabv3_mobilenetv2_cityscapes/model.meta (node names in
    SURVEY.md Appendix A; structure pinned by tests/test_spec.py against a decode of that file),
  * the ops ``create_student_v3`` adds on top (reference utils/graph_utils.py:373-408: class gather,
    argmax, one-hot labels, weights, mean_iou, softmax CE, masked mean; :457-496: Adam, BN update
    dependencies, masked update),
  * the frozen/inference variant (utils/graph_utils.py:52-76, :362-369: moving statistics, eps 1e-3),
  * the metric path of the frozen wrapper (SemanticNetwork.py:96-115),
using the TF 1.15 op semantics listed in SURVEY.md Appendix C.  It is cross-checked against an
independent NumPy restatement (oracle/student_np.py); gradients come from torch.autograd over the
restated forward, so the hand-written HIP backward is checked against something it shares no code with.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ams_amd import spec as S


def _same_pad_2d(x: torch.Tensor, k: int, stride: int, rate: int) -> torch.Tensor:
    """TF 'SAME' zero padding on an NCHW tensor (Appendix C.1: surplus pad goes bottom/right)."""
    _, pt, pb = S.same_pad(x.shape[2], k, stride, rate)
    _, pl, pr = S.same_pad(x.shape[3], k, stride, rate)
    if pt or pb or pl or pr:
        x = F.pad(x, (pl, pr, pt, pb))
    return x


def resize_bilinear_align_corners(x: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """tf.image.resize_bilinear(align_corners=True, half_pixel_centers=False) on NHWC (Appendix C.5)."""
    b, in_h, in_w, c = x.shape

    def taps(n_in, n_out):
        scale = np.float32((n_in - 1) / (n_out - 1)) if n_out > 1 else np.float32(0.0)
        src = np.arange(n_out, dtype=np.float32) * scale          # f32, like the TF kernel
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy(src - lo.astype(np.float32))

    y0, y1, ty = taps(in_h, out_h)
    x0, x1, tx = taps(in_w, out_w)
    tx = tx.to(x.dtype).view(1, 1, -1, 1)
    ty = ty.to(x.dtype).view(1, -1, 1, 1)
    rows0 = x.index_select(1, y0)
    rows1 = x.index_select(1, y1)
    tl, tr = rows0.index_select(2, x0), rows0.index_select(2, x1)
    bl, br = rows1.index_select(2, x0), rows1.index_select(2, x1)
    top = tl + (tr - tl) * tx
    bot = bl + (br - bl) * tx
    return top + (bot - top) * ty


def preprocess(frames: torch.Tensor) -> torch.Tensor:
    """nodes concat / concat_1 (one row, one column of 127.5), mul_4, sub_2: NHWC 0..255 -> NCHW network input."""
    x = F.pad(frames.permute(0, 3, 1, 2), (0, 1, 0, 1), value=S.PAD_VALUE)
    return x * torch.tensor(S.PIXEL_SCALE, dtype=torch.float32).to(frames.dtype) - 1.0


def conv_same(x: torch.Tensor, w: torch.Tensor, kind: str, stride: int, rate: int) -> torch.Tensor:
    """Conv2D (kind 'conv', w HWIO) or DepthwiseConv2dNative (kind 'dw', w HWC1) with SAME padding, NCHW activations."""
    xp = _same_pad_2d(x, w.shape[0], stride, rate)
    if kind == "dw":
        return F.conv2d(xp, w.permute(2, 3, 0, 1), stride=stride, dilation=rate, groups=w.shape[2])
    return F.conv2d(xp, w.permute(3, 2, 0, 1), stride=stride)


def batch_norm_train(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """FusedBatchNormV3(is_training=True) on NCHW: (y, batch mean, UNBIASED batch variance) — outputs 0, 1, 2 (Appendix C.3)."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mu = x.mean(dim=(0, 2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(0, 2, 3), keepdim=True)                # biased: what normalises
    y = (x - mu) * torch.rsqrt(var + eps) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    return y, mu.reshape(-1), (var * (n / max(n - 1, 1))).reshape(-1)


def batch_norm_frozen(x, gamma, beta, moving_mean, moving_var, eps: float = S.BN_EPS_FROZEN):
    v = lambda t: t.view(1, -1, 1, 1)  # noqa: E731
    return (x - v(moving_mean)) * torch.rsqrt(v(moving_var) + eps) * v(gamma) + v(beta)


def ema_update(moving: torch.Tensor, stat: torch.Tensor) -> torch.Tensor:
    """AssignMovingAvg: moving -= (moving - stat) * (1 - decay), (1 - decay) formed in f32 as the graph's Sub node does."""
    return moving - (moving - stat) * float(np.float32(1.0) - np.float32(S.BN_DECAY))


ADAM_B1 = float(np.float32(0.9))          # the hyper-parameters are f32 tensors in the TF graph: these are their VALUES
ADAM_B2 = float(np.float32(0.999))
ADAM_EPS = float(np.float32(1e-8))
ADAM_1MB1 = float(np.float32(1.0) - np.float32(0.9))       # (T(1) - beta1()) formed in f32 by the ApplyAdam functor
ADAM_1MB2 = float(np.float32(1.0) - np.float32(0.999))


def adam_update(w, g, m, v, lr: float, beta1_power: float, beta2_power: float):
    """TensorFlow's ApplyAdam functor (core/kernels/training_ops.cc; Appendix C.10), eps OUTSIDE the square root:
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= (m*alpha)/(sqrt(v)+eps).  -> (w, m, v)."""
    alpha = lr * np.sqrt(1.0 - beta2_power) / (1.0 - beta1_power)
    m = m + (g - m) * ADAM_1MB1
    v = v + (g * g - v) * ADAM_1MB2
    return w - (m * alpha) / (torch.sqrt(v) + ADAM_EPS), m, v


class StudentOracle:
    """Stateful restatement of the live (trainable) student plus its frozen twin."""

    def __init__(self, variables: Dict[str, np.ndarray], class_indices: Sequence[int], num_classes: int = 19,
                 dtype: torch.dtype = torch.float32):
        self.spec = S.build_spec(num_classes)
        self.dtype = dtype
        self.class_indices = torch.as_tensor(np.asarray(class_indices, dtype=np.int64))
        self.K = len(class_indices)
        self.vars: Dict[str, torch.Tensor] = {}
        for name in self.spec.all_variable_names():
            self.vars[name] = torch.tensor(np.asarray(variables[name]), dtype=dtype)
        # Adam slots; never reset by restore (SemanticNetwork.py:25, :154-156)
        self.adam_m = {v.name: torch.zeros(v.shape, dtype=dtype) for v in self.spec.trainable}
        self.adam_v = {v.name: torch.zeros(v.shape, dtype=dtype) for v in self.spec.trainable}
        self.beta1_power = ADAM_B1
        self.beta2_power = ADAM_B2
        self.last_batch_stats: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

    # ------------------------------------------------------------------ variables
    def get_vars(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().to(torch.float32).numpy().copy() for k, v in self.vars.items()}

    def restore(self, variables: Dict[str, np.ndarray]) -> None:
        """SaveHelper.restore_vars through the OPT_FILTER: model variables only, Adam state untouched."""
        for k, a in variables.items():
            if k in self.vars:
                self.vars[k] = torch.tensor(np.asarray(a), dtype=self.dtype)

    # ------------------------------------------------------------------ forward
    def _bn(self, x: torch.Tensor, layer: S.Layer, mode: str, p: Dict[str, torch.Tensor]) -> torch.Tensor:
        g = p[layer.scope + "/BatchNorm/gamma:0"]
        b = p[layer.scope + "/BatchNorm/beta:0"]
        if mode == "frozen":
            return batch_norm_frozen(x, g, b, p[layer.scope + "/BatchNorm/moving_mean:0"], p[layer.scope + "/BatchNorm/moving_variance:0"])
        y, mu, var_unbiased = batch_norm_train(x, g, b, layer.bn_eps)
        self.last_batch_stats[layer.scope] = (mu.detach(), var_unbiased.detach())      # what feeds the moving averages
        return y

    @staticmethod
    def _act(x: torch.Tensor, act: str) -> torch.Tensor:
        if act == "relu6":
            return torch.clamp(x, 0.0, 6.0)
        if act == "relu":
            return torch.relu(x)
        return x

    def forward_lowres(self, frames, mode: str = "frozen", params: Optional[Dict[str, torch.Tensor]] = None,
                       taps: Optional[Dict[str, torch.Tensor]] = None, ztaps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """frames [B,H,W,3] (0..255) -> logits [B,h,w,NUM_CLASSES] at output stride 16 (NHWC).

        mode 'frozen': moving statistics, eps 1e-3 everywhere (the graph shipped to the edge);
        mode 'train' : batch statistics with each layer's own eps (the live graph, is_training=True).
        ``taps`` (optional dict) receives every layer's post-activation output as NHWC, for kernel tests; ``ztaps`` every layer's raw
        conv output in the graph's own NCHW layout (the tensors themselves: autograd can differentiate with respect to them).
        """
        p = params if params is not None else self.vars
        x = torch.as_tensor(np.asarray(frames), dtype=self.dtype) if not torch.is_tensor(frames) else frames.to(self.dtype)
        x = preprocess(x)
        outs: Dict[int, torch.Tensor] = {0: x}
        layers = self.spec.layers
        backbone = [l for l in layers if l.scope.startswith("MobilenetV2")]
        for l in backbone:
            w = p[l.weight_name]
            xin = outs[l.idx - 1]
            y = conv_same(xin, w, l.kind, l.stride, l.rate)
            if ztaps is not None:
                ztaps[l.scope] = y
            y = self._act(self._bn(y, l, mode, p), l.act)
            if l.residual_from is not None:
                y = y + outs[l.residual_from]
            outs[l.idx] = y
            if taps is not None:
                taps[l.scope] = y.permute(0, 2, 3, 1)
        feat = outs[backbone[-1].idx]
        lp, la, lc, ll = layers[-4], layers[-3], layers[-2], layers[-1]
        pooled = feat.mean(dim=(2, 3), keepdim=True)                        # node Mean
        pool = F.conv2d(pooled, p[lp.weight_name].permute(3, 2, 0, 1))
        pool = self._act(self._bn(pool, lp, mode, p), lp.act)
        pool = pool.expand(-1, -1, feat.shape[2], feat.shape[3])            # ResizeBilinear of a 1x1 map
        aspp = F.conv2d(feat, p[la.weight_name].permute(3, 2, 0, 1))
        if ztaps is not None:
            ztaps[la.scope] = aspp
        aspp = self._act(self._bn(aspp, la, mode, p), la.act)
        cat = torch.cat([pool, aspp], dim=1)                                # concat_2: pool branch first
        proj = F.conv2d(cat, p[lc.weight_name].permute(3, 2, 0, 1))
        if ztaps is not None:
            ztaps[lc.scope] = proj
        proj = self._act(self._bn(proj, lc, mode, p), lc.act)
        logits = F.conv2d(proj, p[ll.weight_name].permute(3, 2, 0, 1)) + p[ll.scope + "/biases:0"].view(1, -1, 1, 1)
        if taps is not None:
            taps["image_pooling"] = pool[:, :, :1, :1].permute(0, 2, 3, 1)
            taps["aspp0"] = aspp.permute(0, 2, 3, 1)
            taps["concat_projection"] = proj.permute(0, 2, 3, 1)
        return logits.permute(0, 2, 3, 1)

    def logits_full(self, frames, mode: str = "frozen", params=None, taps=None, ztaps=None) -> torch.Tensor:
        """student_logits: ResizeBilinear_1 is an identity resize, ResizeBilinear_2 goes to the UNPADDED H x W."""
        h, w = frames.shape[1], frames.shape[2]
        return resize_bilinear_align_corners(self.forward_lowres(frames, mode, params, taps, ztaps), h, w)

    # ------------------------------------------------------------------ heads added by create_student_v3
    def reduced_logits(self, logits_full: torch.Tensor) -> torch.Tensor:
        return logits_full.index_select(3, self.class_indices)             # graph_utils.py:373

    def predict(self, frames, mode: str = "frozen") -> np.ndarray:
        with torch.no_grad():
            z = self.reduced_logits(self.logits_full(frames, mode))
            return torch.argmax(z, dim=-1).to(torch.int32).numpy()         # first maximum, graph_utils.py:390

    def label_targets(self, labels_teacher) -> Tuple[torch.Tensor, torch.Tensor]:
        """cast -> one_hot(NUM_CLASSES) -> gather K -> argmax; weight = sum of the gathered one-hot row."""
        lab = torch.as_tensor(np.asarray(labels_teacher)).to(torch.float32).to(torch.int64)   # tf.cast(labels, int32)
        nc = self.spec.num_classes
        in_range = (lab >= 0) & (lab < nc)
        onehot = F.one_hot(torch.where(in_range, lab, torch.zeros_like(lab)), nc) * in_range.unsqueeze(-1)
        sel = onehot.index_select(-1, self.class_indices)
        return torch.argmax(sel, dim=-1), sel.sum(dim=-1)

    def loss_from_reduced(self, z: torch.Tensor, target: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """softmax_cross_entropy_with_logits + boolean_mask + reduce_mean (graph_utils.py:403-408)."""
        lse = torch.logsumexp(z, dim=-1)
        picked = torch.gather(z, -1, target.unsqueeze(-1)).squeeze(-1)
        pixel = lse - picked
        valid = weight > 0
        return pixel[valid].mean()                                          # NaN when no pixel is valid

    def soft_targets(self, teacher_logits, out_h: int, out_w: int) -> torch.Tensor:
        """filtered_teacher_labels_probs (graph_utils.py:375-376): gather the K classes of the fed teacher logits, softmax.  The reference feeds
        logits of the label size; a smaller grid is first resized like the student's logits (align corners) — the identity at the full size."""
        t = torch.as_tensor(np.asarray(teacher_logits)).to(self.dtype)
        if t.shape[1] != out_h or t.shape[2] != out_w:
            t = resize_bilinear_align_corners(t, out_h, out_w)
        return torch.softmax(t.index_select(3, self.class_indices), dim=-1)

    def soft_loss_from_reduced(self, z: torch.Tensor, probs: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """soft_teacher=True (graph_utils.py:403-404, 406-408): softmax_cross_entropy_with_logits(labels = teacher probabilities), masked by the
        HARD labels' weights, mean.  (The v1 op does not back-propagate into its labels; they come from a placeholder anyway.)"""
        pixel = -(probs.detach() * torch.log_softmax(z, dim=-1)).sum(dim=-1)
        valid = weight > 0
        return pixel[valid].mean()

    def regularizer(self, params: Dict[str, torch.Tensor], biases_only: bool) -> torch.Tensor:
        """regularize=True (graph_utils.py:451-456): 0.01 * reduce_mean([l2_loss(v) for v in tvars]); tvars loses every name with 'weight'
        in it under train_biases_only."""
        terms = [params[v.name].pow(2).sum() / 2 for v in self.spec.trainable if not (biases_only and 'weight' in v.name)]
        return 0.01 * torch.stack(terms).mean()

    def predict_with_metric(self, frames, labels_teacher, mode: str = "frozen"):
        """(labels, conf_mat f64 [K,K], loss) like SemanticNetwork.predict_with_metric (:196-213)."""
        with torch.no_grad():
            z = self.reduced_logits(self.logits_full(frames, mode))
            pred = torch.argmax(z, dim=-1)
            target, weight = self.label_targets(labels_teacher)
            cm = torch.zeros(self.K * self.K, dtype=torch.float64)
            cm.index_add_(0, (target * self.K + pred).reshape(-1), weight.reshape(-1).to(torch.float64))
            loss = self.loss_from_reduced(z, target, weight)
        return pred.to(torch.int32).numpy(), cm.view(self.K, self.K).numpy(), float(loss)

    # ------------------------------------------------------------------ one optimisation step
    def gradients(self, frames, labels_teacher, teacher_logits=None, regularize: bool = False,
                  train_biases_only: bool = False) -> Tuple[float, Dict[str, torch.Tensor]]:
        """Loss and d(loss)/d(trainable) of the live graph (BN in training mode).  ``teacher_logits``: the soft_teacher graph's feed."""
        params = dict(self.vars)
        leaves = {}
        for v in self.spec.trainable:
            leaf = self.vars[v.name].clone().requires_grad_(True)
            params[v.name] = leaf
            leaves[v.name] = leaf
        z = self.reduced_logits(self.logits_full(frames, "train", params))
        target, weight = self.label_targets(labels_teacher)
        if teacher_logits is not None:
            loss = self.soft_loss_from_reduced(z, self.soft_targets(teacher_logits, z.shape[1], z.shape[2]), weight)
        else:
            loss = self.loss_from_reduced(z, target, weight)
        if regularize:
            loss = loss + self.regularizer(params, train_biases_only)
        grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=False)
        return float(loss.detach()), {k: g for k, g in zip(leaves, grads)}

    def train_step(self, frames, labels_teacher, lr: float, mask: Optional[Dict[str, np.ndarray]] = None,
                   grads_override: Optional[Dict[str, torch.Tensor]] = None, teacher_logits=None, regularize: bool = False,
                   train_biases_only: bool = False) -> float:
        """forward (BN batch stats) -> CE -> backward -> BN EMA (decay 0.9) -> Adam (TF1 form) [-> mask].

        Adam (Appendix C.10): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMAs; w -= lr_t*m/(sqrt(v)+1e-8).
        With ``mask`` (coordinate descent, graph_utils.py:482-493) every variable first takes the full Adam
        step and is then reverted where mask is False; the moments advance for all entries regardless."""
        loss, grads = self.gradients(frames, labels_teacher, teacher_logits, regularize, train_biases_only)
        if grads_override is not None:
            grads = grads_override
        for l in self.spec.layers:
            if l.bn_eps is None:
                continue
            mu, var_unbiased = self.last_batch_stats[l.scope]
            for name, stat in ((l.scope + "/BatchNorm/moving_mean:0", mu), (l.scope + "/BatchNorm/moving_variance:0", var_unbiased)):
                self.vars[name] = ema_update(self.vars[name], stat)
        for v in self.spec.trainable:
            g = grads[v.name].to(self.dtype)
            new, self.adam_m[v.name], self.adam_v[v.name] = adam_update(self.vars[v.name], g, self.adam_m[v.name], self.adam_v[v.name],
                                                                        lr, self.beta1_power, self.beta2_power)
            if mask is not None:
                keep = torch.as_tensor(np.asarray(mask[v.name]).astype(bool))
                new = torch.where(keep, new, self.vars[v.name])
            self.vars[v.name] = new.detach()
        self.beta1_power *= ADAM_B1
        self.beta2_power *= ADAM_B2
        return loss


def cross_miou_confusion(labels_before, labels_after, class_indices, num_classes: int = 19) -> np.ndarray:
    """phi-score confusion matrix of two teacher label maps (SemanticNetwork.py:124-139, :184-194)."""
    ci = np.asarray(class_indices)
    k = len(ci)
    lut = np.full(num_classes + 1, -1, dtype=np.int64)
    lut[ci] = np.arange(k)

    def reduce(lbl):
        lbl = np.asarray(lbl).astype(np.int64)
        ok = (lbl >= 0) & (lbl < num_classes)
        return lut[np.where(ok, lbl, num_classes)]

    a, b = reduce(labels_before), reduce(labels_after)
    valid = (a >= 0) & (b >= 0)
    cm = np.zeros((k, k), dtype=np.float64)
    np.add.at(cm, (a[valid], b[valid]), 1.0)
    return cm
