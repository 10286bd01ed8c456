#!/usr/bin/env python3
"""Generate tests/golden/cv_resize_vectors.json: small uint8 images and what OpenCV's cv2.resize makes of them.

OpenCV is not installable here, so the expected outputs are derived from its published algorithm (modules/imgproc/src/resize.cpp,
INTER_LINEAR 8-bit generic path and resizeNN; summary in oracle/cv_resize.py's header) by the scalar, pixel-at-a-time loop below —
written independently of the vectorised oracle, with `struct` for the float32 roundings and no NumPy.  Cases marked "hand" were
worked out on paper first (the arithmetic is in the comments); the loop must reproduce them.
"""
import json
import math
import struct
from pathlib import Path

OUT = Path(__file__).resolve().parent / "cv_resize_vectors.json"


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def cv_round(x):
    """cvRound: nearest integer, halves to even."""
    fl = math.floor(x)
    d = x - fl
    if d > 0.5 or (d == 0.5 and fl % 2 == 1):
        return int(fl) + 1
    return int(fl)


def taps(src, dst, is_column):
    scale = 1.0 / (dst / src)
    out = []
    for d in range(dst):
        f = f32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = f32(f - s)
        if is_column:
            if s < 0:
                s, f = 0, 0.0
            if s >= src - 1:
                s, f = src - 1, 0.0
        a0 = cv_round(f32(f32(1.0 - f) * 2048.0))
        a1 = cv_round(f32(f * 2048.0))
        out.append((s, a0, a1))
    return out


def linear(img, ow, oh):
    h, w, c = len(img), len(img[0]), len(img[0][0])
    if (h, w) == (oh, ow):
        return [[list(p) for p in row] for row in img]
    if w == 2 * ow and h == 2 * oh:
        return [[[(img[2 * y][2 * x][k] + img[2 * y][2 * x + 1][k] + img[2 * y + 1][2 * x][k] + img[2 * y + 1][2 * x + 1][k] + 2) >> 2
                  for k in range(c)] for x in range(ow)] for y in range(oh)]
    tx, ty = taps(w, ow, True), taps(h, oh, False)
    out = []
    for (sy, b0, b1) in ty:
        r0 = min(max(sy, 0), h - 1)
        r1 = min(max(sy + 1, 0), h - 1)
        row = []
        for (sx, a0, a1) in tx:
            px = []
            for k in range(c):
                right = min(sx + 1, w - 1)
                d0 = img[r0][sx][k] * a0 + img[r0][right][k] * a1
                d1 = img[r1][sx][k] * a0 + img[r1][right][k] * a1
                px.append((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2)
            row.append(px)
        out.append(row)
    return out


def nearest(img, ow, oh):
    h, w = len(img), len(img[0])
    sy, sx = 1.0 / (oh / h), 1.0 / (ow / w)
    return [[img[min(int(math.floor(y * sy)), h - 1)][min(int(math.floor(x * sx)), w - 1)] for x in range(ow)] for y in range(oh)]


def pattern(h, w, c, seed):
    return [[[(seed * 131 + 97 * y + 57 * x + 29 * k + (y * x) % 7 * 13) % 256 for k in range(c)] for x in range(w)] for y in range(h)]


def main():
    cases = []
    # hand 1: one row [0, 100] -> 3 wide.  scale 2/3: d=0: f=-1/6 -> clamped to (0, 0): 0.  d=1: f=0.5: 0*1024 + 100*1024 = 102400;
    #   rows: single row, weights (2048, 0): (2048 * (102400 >> 4)) >> 16 = 200; (200 + 0 + 2) >> 2 = 50.  d=2: right edge: 100.
    h1 = linear([[[0], [100]]], 3, 1)
    assert h1 == [[[0], [50], [100]]]
    cases.append({"kind": "linear", "src": [[[0], [100]]], "w": 3, "h": 1, "dst": h1, "hand": True})
    # hand 2: [0, 255] -> 4 wide.  scale 0.5: d=1: f=0.25: weights (1536, 512): 255*512 = 130560; >>4 = 8160; *2048 >> 16 = 255;
    #   (255 + 2) >> 2 = 64 (63.75 in exact arithmetic).  d=2: weights (512, 1536): 391680 >> 4 = 24480; *2048 >> 16 = 765; 767 >> 2 = 191.
    h2 = linear([[[0], [255]]], 4, 1)
    assert h2 == [[[0], [64], [191], [255]]]
    cases.append({"kind": "linear", "src": [[[0], [255]]], "w": 4, "h": 1, "dst": h2, "hand": True})
    # hand 3: exact 2x down-scale -> INTER_AREA fast path: (10 + 20 + 30 + 43 + 2) >> 2 = 26 (25.75), (1 + 2 + 2 + 1 + 2) >> 2 = 2 (1.5 rounds up)
    src3 = [[[10], [20], [1], [2]], [[30], [43], [2], [1]]]
    h3 = linear(src3, 2, 1)
    assert h3 == [[[26], [2]]]
    cases.append({"kind": "linear", "src": src3, "w": 2, "h": 1, "dst": h3, "hand": True})
    # hand 4: column [0, 100] (2 rows) -> 4 rows.  rows keep their weights at the border: d=0: f=-0.25 -> s=-1, f=0.75, both rows clamp to
    #   row 0: 0.  d=1: f=0.25: (1536*(0>>4)>>16) + (512*((100*2048)>>4)>>16) = 512*12800>>16 = 100; (100+2)>>2 = 25.  d=2: 1536*12800>>16
    #   = 300; 302>>2 = 75.  d=3: f=1.25 -> s=1, f=0.25: rows 1 and clamp(2)=1: (1536*12800>>16) + (512*12800>>16) = 300 + 100; 402>>2 = 100.
    h4 = linear([[[0]], [[100]]], 1, 4)
    assert h4 == [[[0]], [[25]], [[75]], [[100]]]
    cases.append({"kind": "linear", "src": [[[0]], [[100]]], "w": 1, "h": 4, "dst": h4, "hand": True})
    # nearest, hand: 5 -> 3: floor(d * 5/3) = 0, 1, 3;  3 -> 5: floor(d * 0.6) = 0, 0, 1, 1, 2
    n1 = nearest([[1, 2, 3, 4, 5]], 3, 1)
    n2 = nearest([[7, 8, 9]], 5, 1)
    assert n1 == [[1, 2, 4]] and n2 == [[7, 7, 8, 8, 9]]
    cases.append({"kind": "nearest", "src": [[1, 2, 3, 4, 5]], "w": 3, "h": 1, "dst": n1, "hand": True})
    cases.append({"kind": "nearest", "src": [[7, 8, 9]], "w": 5, "h": 1, "dst": n2, "hand": True})
    # generated: up- and down-scales with odd ratios, 3 channels, identity, 2x, 1-pixel sources
    for i, (h, w, oh, ow) in enumerate([(5, 7, 8, 16), (9, 11, 4, 6), (6, 10, 3, 5), (4, 4, 4, 4), (1, 1, 3, 5), (7, 5, 7, 9), (12, 19, 5, 8),
                                        (3, 8, 9, 3), (16, 12, 8, 6), (2, 2, 5, 5)]):
        img = pattern(h, w, 3, i)
        cases.append({"kind": "linear", "src": img, "w": ow, "h": oh, "dst": linear(img, ow, oh)})
        lab = [[p[0] % 20 for p in row] for row in img]
        cases.append({"kind": "nearest", "src": lab, "w": ow, "h": oh, "dst": nearest(lab, ow, oh)})
    OUT.write_text(json.dumps({"about": "cv2.resize on uint8, OpenCV generic path; generator tests/golden/make_cv_resize_vectors.py",
                               "cases": cases}, separators=(",", ":")))
    print(OUT, OUT.stat().st_size, "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
