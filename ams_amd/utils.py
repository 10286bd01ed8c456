"""Host-side helpers of the hot path: replay-memory sampling, mIoU, frame selection, palettes, resizing.

Behavioural mirrors of reference utils/utils.py (``mini_batch`` :129-185, ``calculate_miou`` :80-126,
``choose_frames`` :237-254, ``colormap`` :52-77, ``string_class_iou`` :188-213); values are pinned against
outputs captured from the reference (tests/golden/ref_helpers.json, tests/test_helpers_golden.py).
OpenCV is not available here, so the two resamplers the reference takes from cv2 (``INTER_LINEAR`` for
frames, ``INTER_NEAREST`` for labels; run.py:179-183, utils/utils.py:165-173) are restated in NumPy, the uint8 linear one
in OpenCV's own fixed-point arithmetic.
"""
from __future__ import annotations

import random
from collections import deque

import numpy as np

CITYSCAPES_NAMES = ('road', 'sidewalk', 'building', 'wall', 'fence', 'pole', 'traffic light', 'traffic sign',
                    'vegetation', 'terrain', 'sky', 'person', 'rider', 'car', 'truck', 'bus', 'train',
                    'motorcycle', 'bicycle')

_CITYSCAPES_PALETTE = (
    (128, 64, 128), (244, 35, 232), (70, 70, 70), (102, 102, 156), (190, 153, 153), (153, 153, 153),
    (250, 170, 30), (220, 220, 0), (107, 142, 35), (152, 251, 152), (70, 130, 180), (220, 20, 60),
    (255, 0, 0), (0, 0, 142), (0, 0, 70), (0, 60, 100), (0, 80, 100), (0, 0, 230), (119, 11, 32))


def colormap(name='cityscapes'):
    """uint8 [256, 3] palette; rows >= 19 are black."""
    if name != 'cityscapes':
        raise Exception('Unknown colormap')
    table = np.zeros((256, 3), dtype=np.uint8)
    table[:len(_CITYSCAPES_PALETTE)] = np.asarray(_CITYSCAPES_PALETTE, dtype=np.uint8)
    return table


def calculate_miou(conf_matrix, population=False, detailed=False, nan=False):
    """Per-class IoU list from a confusion matrix (rows = teacher labels, cols = student predictions).

    IoU_i = cm[i,i] / (row_i + col_i - cm[i,i]); a class that is neither present nor predicted yields
    ``nan`` (``nan=True``) or the string ``'Not predicted/present'``.  Optional extras follow the
    reference's return conventions: population shares, false-negative and false-positive rates.
    """
    cm = np.asarray(conf_matrix)
    k = len(cm[0])
    rows = cm.sum(axis=1)
    cols = cm.sum(axis=0)
    ious, fneg, fpos = [], [], []
    for i in range(k):
        union = rows[i] + cols[i] - cm[i][i]
        if union == 0:
            ious.append(np.nan if nan else 'Not predicted/present')
            fneg.append(0)
            fpos.append(0)
            continue
        ious.append(cm[i][i] / max(union, 1))
        fneg.append((rows[i] - cm[i][i]) / union)
        fpos.append((cols[i] - cm[i][i]) / union)
    result = [ious]
    if population:
        result.append(rows / np.sum(rows))
    if detailed:
        result += [fneg, fpos]
    return result[0] if len(result) == 1 else tuple(result)


def choose_frames(frame_label_list, sample_fraction):
    """Pick round(fraction * n) equally spaced (frame, label) pairs, always ending on the newest one."""
    n = len(frame_label_list)
    samples = int(np.round(sample_fraction * n))
    picks = np.round(np.linspace(-1, n - 1, samples + 1, endpoint=True)[1:]).astype(int)
    assert picks.size == samples, f"indices had {picks.size} values but samples is {samples}"
    return ([frame_label_list[i][0] for i in picks], [frame_label_list[i][1] for i in picks])


def _cv_scale(n_in, n_out):
    """OpenCV forms the source step as 1 / (dst / src) in double (cv::resize: inv_scale = dsize / ssize; scale = 1. / inv_scale)."""
    return 1.0 / (float(n_out) / float(n_in))


def resize_nearest(img, out_w, out_h):
    """cv2.resize(..., interpolation=INTER_NEAREST): src = min(floor(dst * scale), size - 1)  (resizeNN)."""
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(out_h) * _cv_scale(h, out_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * _cv_scale(w, out_w)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def _fixed_taps(n_in, n_out, border_zeroes_weight):
    """Tap table of OpenCV's 8-bit INTER_LINEAR: (first tap, second tap, weight of the first, weight of the second) per output
    index; weights are 11-bit integers rounded half-to-even from float32 products.  Columns zero the fractional weight where the
    window leaves the image; rows keep it and clamp the taps instead."""
    first = np.empty(n_out, np.int64)
    second = np.empty(n_out, np.int64)
    w0 = np.empty(n_out, np.int64)
    w1 = np.empty(n_out, np.int64)
    step = _cv_scale(n_in, n_out)
    for d in range(n_out):
        pos = np.float32((d + 0.5) * step - 0.5)
        base = int(np.floor(pos))
        frac = np.float32(pos - np.float32(base))
        if border_zeroes_weight and (base < 0 or base >= n_in - 1):
            base, frac = min(max(base, 0), n_in - 1), np.float32(0.0)
        first[d] = min(max(base, 0), n_in - 1)
        second[d] = min(max(base + 1, 0), n_in - 1)
        w0[d] = int(np.rint((np.float32(1.0) - frac) * np.float32(2048.0)))
        w1[d] = int(np.rint(frac * np.float32(2048.0)))
    return first, second, w0, w1


def resize_linear(img, out_w, out_h):
    """cv2.resize(..., interpolation=INTER_LINEAR).

    uint8 images follow OpenCV's fixed-point path bit for bit (11-bit weights, integer horizontal pass, ``>> 4`` / ``>> 16`` /
    ``(+2) >> 2`` vertical pass; an exact 2x down-scale is the 2x2 box average ``(a+b+c+d+2) >> 2``; equal sizes copy) — pinned
    against hand-derived vectors through oracle/cv_resize.py (tests/test_cv_resize.py).  Other dtypes interpolate in float64
    with half-pixel centres and edge clamp."""
    h, w = img.shape[:2]
    if img.dtype == np.uint8:
        if (h, w) == (out_h, out_w):
            return img.copy()
        wide = img.astype(np.int64)
        if h == 2 * out_h and w == 2 * out_w:
            return ((wide[0::2, 0::2] + wide[0::2, 1::2] + wide[1::2, 0::2] + wide[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        x0, x1, ax0, ax1 = _fixed_taps(w, out_w, True)
        y0, y1, by0, by1 = _fixed_taps(h, out_h, False)
        tail = (1,) * (img.ndim - 2)
        rows = np.take(wide, x0, axis=1) * ax0.reshape((1, -1) + tail) + np.take(wide, x1, axis=1) * ax1.reshape((1, -1) + tail)
        upper = (by0.reshape((-1, 1) + tail) * (np.take(rows, y0, axis=0) >> 4)) >> 16
        lower = (by1.reshape((-1, 1) + tail) * (np.take(rows, y1, axis=0) >> 4)) >> 16
        return ((upper + lower + 2) >> 2).astype(np.uint8)
    src = img.astype(np.float64)

    def taps(n_in, n_out):
        pos = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        lo = np.floor(pos).astype(np.int64)
        frac = pos - lo
        lo_c = np.clip(lo, 0, n_in - 1)
        hi_c = np.clip(lo + 1, 0, n_in - 1)
        return lo_c, hi_c, frac

    y0, y1, fy = taps(h, out_h)
    x0, x1, fx = taps(w, out_w)
    fx = fx.reshape((1, -1) + (1,) * (src.ndim - 2))
    fy = fy.reshape((-1, 1) + (1,) * (src.ndim - 2))
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    if np.issubdtype(img.dtype, np.integer):
        out = np.clip(np.rint(out), np.iinfo(img.dtype).min, np.iinfo(img.dtype).max).astype(img.dtype)
    else:
        out = out.astype(img.dtype)
    return out


def mini_batch(deque_images, deque_labels, crop_size, scale, mini_batch_size, num_of_iterations, flip=False):
    """Sample training batches from the replay memory, with replacement.

    Returns float64 arrays ``[num_of_iterations, mini_batch_size, crop_h, crop_w, C]`` and
    ``[num_of_iterations, mini_batch_size, crop_h, crop_w]``.  RNG consumption per sample is part of the
    contract (it decides which frames a seeded run trains on): one ``np.random.choice`` for the memory
    slot, then ``random.randint`` for the scale choice, the row offset and the column offset, in that
    order; with ``flip`` one ``np.random.random`` more.  Rescaled copies are cached per (scale, slot).
    """
    images = list(deque_images) if isinstance(deque_images, deque) else deque_images
    labels = list(deque_labels) if isinstance(deque_labels, deque) else deque_labels
    crop_h, crop_w = crop_size[0], crop_size[1]
    out_img = np.empty((num_of_iterations, mini_batch_size, crop_h, crop_w, images[0].shape[2]))
    out_lbl = np.empty((num_of_iterations, mini_batch_size, crop_h, crop_w))
    cache = {}
    n_mem = len(images)
    for it in range(num_of_iterations):
        for j in range(mini_batch_size):
            slot = np.random.choice(n_mem)
            src_h, src_w = images[slot].shape[0], images[slot].shape[1]
            s = scale[random.randint(0, len(scale) - 1)]
            factor = s * crop_w / src_w
            slack_h = int(src_h * factor) - crop_h
            slack_w = int(src_w * factor) - crop_w
            assert slack_w >= 0
            assert slack_h >= 0
            top = random.randint(0, slack_h)
            left = random.randint(0, slack_w)
            key = (s, slot)
            if key not in cache:
                if factor == 1 and s == 1:
                    cache[key] = (images[slot], labels[slot])
                else:
                    tw, th = int(src_w * factor), int(src_h * factor)
                    cache[key] = (resize_linear(images[slot], tw, th), resize_nearest(labels[slot], tw, th))
            img, lbl = cache[key]
            img = img[top:top + crop_h, left:left + crop_w, :]
            lbl = lbl[top:top + crop_h, left:left + crop_w]
            if flip and np.random.random() > 0.5:
                img, lbl = img[:, ::-1, :], lbl[:, ::-1]
            out_img[it][j] = img
            out_lbl[it][j] = lbl
    return out_img, out_lbl


def string_class_iou(class_iou_list, population=None, headers=None, class_weights=None):
    """Human-readable per-class IoU table (cosmetic; reference utils/utils.py:188-213)."""
    names = list(CITYSCAPES_NAMES)
    if class_weights is not None:
        names = [names[i] for i in np.where(np.asarray(class_weights).reshape(-1) == 1)[0]]
    columns = class_iou_list if isinstance(class_iou_list[0], list) else [class_iou_list]
    lines = []
    if headers is not None:
        lines.append("%22s\t" % "" + "".join(h + "\t\t" for h in headers))
    for i in range(len(columns[0])):
        tag = names[i] + ("(%.3g):" % (population[i] * 100.0) if population is not None else ":")
        cells = "".join((c[i] + "\t") if isinstance(c[i], str) else ("%.1f" % (c[i] * 100.0) + "\t\t\t")
                        for c in columns)
        lines.append("%-22s\t%s" % (tag, cells))
    return "\n".join(lines) + "\n"


def take_array_for(class_weights_exp):
    """Map full-label-space id -> index inside the selected subset (0 for unselected ids).

    Restates the expression of reference SemanticNetwork.py:58-61."""
    w = np.asarray(class_weights_exp).reshape(-1)
    ranks = np.cumsum(w) * w
    return np.where(ranks != 0, ranks - 1, ranks).astype(int)
