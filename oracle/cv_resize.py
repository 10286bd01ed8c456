"""OpenCV's ``cv2.resize`` for uint8 images, restated exactly.  TEST INFRASTRUCTURE ONLY.

The reference resizes every decoded frame with ``cv2.resize(frame, (2H, H))`` (INTER_LINEAR) and every teacher label map with
``interpolation=cv2.INTER_NEAREST`` (reference run.py:179-183, :415-421; utils/utils.py:165-173).  OpenCV is a third-party
dependency (conda ``opencv 3.4.2`` / ``opencv-python 4.1.2.30``, reference environment.yml:86, :148) that is not installed here
and never travels to the GPU box, so its published algorithm (modules/imgproc/src/resize.cpp, generic C++ path — the SIMD paths
are bit-exact with it) is restated here and pinned by hand-derived vectors (tests/golden/cv_resize_vectors.json, generator
make_cv_resize_vectors.py).  ``ams_amd.utils.resize_linear`` / ``resize_nearest`` and the device kernel ``k_ingest.hip`` must
both equal this file bit for bit (tests/test_cv_resize.py, tests/test_gpu_ingest.py).

INTER_LINEAR, 8-bit (``resizeGeneric_`` with ``HResizeLinear<uchar,int,short,2048>`` / ``VResizeLinear<uchar,int,short,
FixedPtCast<int,uchar,22>>``):
  * scale = 1.0 / (dst / src) in double; for every destination index d: f = float((d + 0.5) * scale - 0.5); s = floor(f); f -= s.
  * columns: s < 0 -> (s, f) = (0, 0); s >= width-1 -> (s, f) = (width-1, 0) and the pixel takes S[s] * 2048 (no right tap).
    rows: the weights stay as computed, the two source rows are s and s+1 each clamped to [0, height-1].
  * weights as 11-bit integers: a0 = cvRound((1.f - f) * 2048), a1 = cvRound(f * 2048) (float products, round half to even, short).
  * horizontal pass in int: D = S[s] * a0 + S[s+1] * a1.
  * vertical pass: dst = uint8(( ((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2 ) >> 2).
  * exact 2x down-scale in both directions switches to INTER_AREA's fast path (cv::resize: "INTER_AREA (fast) also is equal to
    INTER_LINEAR" for scale 2): dst = (S00 + S01 + S10 + S11 + 2) >> 2.  (Cityscapes 1024x2048 -> 512x1024 takes this path.)
  * equal sizes: plain copy.
INTER_NEAREST (``resizeNN``): s = min(floor(d * scale), size-1) with the same double scale.

Build variance: wheels compiled with Intel IPP may route 8-bit linear resize through IPP, whose result can differ by 1 LSB; the
generic path above is what an IPP-less build (the conda package) computes.
"""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _scale(src: int, dst: int) -> float:
    return 1.0 / (float(dst) / float(src))


def _linear_taps(src: int, dst: int, clamp_weights: bool):
    """-> (s [dst] int64 index of the left/upper tap, a0, a1 [dst] int32 11-bit weights, no_right [dst] bool)."""
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * _scale(src, dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    no_right = np.zeros(dst, dtype=bool)
    if clamp_weights:                       # columns
        low = s < 0
        s = np.where(low, 0, s)
        f = np.where(low, np.float32(0), f).astype(np.float32)
        no_right = s + 1 >= src             # dx >= xmax: the pixel is S[s] * 2048
        high = s >= src - 1
        s = np.where(high, src - 1, s)
        f = np.where(high, np.float32(0), f).astype(np.float32)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int32)      # cvRound: half to even
    a1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    return s, a0, a1, no_right


def resize_linear_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 [H,W] or [H,W,C]."""
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    src = img.astype(np.int32)
    if w == 2 * out_w and h == 2 * out_h:
        q = src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2]
        return ((q + 2) >> 2).astype(np.uint8)
    sx, ax0, ax1, no_right = _linear_taps(w, out_w, True)
    sy, by0, by1, _ = _linear_taps(h, out_h, False)
    shape = (1, out_w) + (1,) * (img.ndim - 2)
    right = np.minimum(sx + 1, w - 1)
    hpass = src[:, sx] * ax0.reshape(shape) + src[:, right] * ax1.reshape(shape)
    hpass = np.where(no_right.reshape(shape), src[:, sx] * COEF_SCALE, hpass)          # [h, out_w, ...] int32
    r0 = np.clip(sy, 0, h - 1)
    r1 = np.clip(sy + 1, 0, h - 1)
    vshape = (out_h, 1) + (1,) * (img.ndim - 2)
    t0 = (by0.reshape(vshape).astype(np.int64) * (hpass[r0] >> 4)) >> 16
    t1 = (by1.reshape(vshape).astype(np.int64) * (hpass[r1] >> 4)) >> 16
    out = (t0 + t1 + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_nearest_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_NEAREST)."""
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(out_h, dtype=np.float64) * _scale(h, out_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w, dtype=np.float64) * _scale(w, out_w)).astype(np.int64), w - 1)
    return img[ys][:, xs]
