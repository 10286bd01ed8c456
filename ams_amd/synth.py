"""Procedural synthetic video + teacher labels.

There is no network for datasets, so every benchmark and parity test runs on seeded synthetic clips:
a smooth low-frequency background plus a few moving rectangles/ellipses, so that consecutive frames
are correlated like video.  Teacher label maps are drawn from the same scene: every region owns a class
id from the experiment's subset and about 10 % of the pixels carry ids outside the subset (or 255) to
exercise the ignore path of the loss / confusion matrix (reference utils/graph_utils.py:392-397).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


class SyntheticVideo:
    def __init__(self, height: int, n_frames: int, class_indices: Sequence[int] = (0, 1, 2, 10, 11, 13),
                 num_classes: int = 19, seed: int = 0, n_shapes: int = 5):
        self.h, self.w = height, 2 * height
        self.n = n_frames
        self.classes = list(class_indices)
        self.num_classes = num_classes
        rng = np.random.default_rng(seed)
        self._rng_seed = seed
        yy, xx = np.mgrid[0:self.h, 0:self.w].astype(np.float32)
        self._yy, self._xx = yy / self.h, xx / self.w
        self._bg_phase = rng.uniform(0, 2 * np.pi, size=(3, 2))
        self._bg_freq = rng.uniform(0.5, 2.5, size=(3, 2))
        outside = [c for c in range(num_classes) if c not in self.classes] or [255]
        self.shapes = []
        for s in range(n_shapes):
            self.shapes.append(dict(
                kind=int(rng.integers(0, 2)),
                cx=rng.uniform(0.1, 0.9), cy=rng.uniform(0.1, 0.9),
                vx=rng.uniform(-0.01, 0.01), vy=rng.uniform(-0.006, 0.006),
                rx=rng.uniform(0.05, 0.2), ry=rng.uniform(0.08, 0.25),
                color=rng.integers(30, 255, size=3),
                cls=int(self.classes[(s + 1) % len(self.classes)]) if s < n_shapes - 1
                else int(outside[int(rng.integers(0, len(outside)))])))
        self._ignore_ids = outside + [255]

    def frame(self, t: int) -> Tuple[np.ndarray, np.ndarray]:
        """(frame uint8 RGB [H,2H,3], teacher label uint8 [H,2H]) at time index t."""
        img = np.empty((self.h, self.w, 3), dtype=np.float32)
        for c in range(3):
            img[..., c] = 110 + 60 * np.sin(2 * np.pi * self._bg_freq[c, 0] * self._xx + self._bg_phase[c, 0] + 0.02 * t) \
                * np.cos(2 * np.pi * self._bg_freq[c, 1] * self._yy + self._bg_phase[c, 1])
        lbl = np.full((self.h, self.w), self.classes[0], dtype=np.uint8)
        lbl[self._yy < 0.35 + 0.05 * np.sin(4 * self._xx + 0.01 * t)] = self.classes[min(3, len(self.classes) - 1)]
        for s in self.shapes:
            cx = (s["cx"] + s["vx"] * t) % 1.0
            cy = (s["cy"] + s["vy"] * t) % 1.0
            dx, dy = (self._xx - cx) / s["rx"], (self._yy - cy) / s["ry"]
            m = (np.abs(dx) < 1) & (np.abs(dy) < 1) if s["kind"] == 0 else (dx * dx + dy * dy < 1)
            img[m] = s["color"]
            lbl[m] = s["cls"]
        rng = np.random.default_rng(self._rng_seed * 100003 + t)
        img += rng.normal(0, 3.0, size=img.shape).astype(np.float32)
        # sprinkle ~3 % ignore pixels (ids outside the subset / 255) on top of the "outside" shape
        noise = rng.random((self.h, self.w)) < 0.03
        lbl[noise] = rng.choice(self._ignore_ids, size=int(noise.sum())).astype(np.uint8)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8), lbl

    def clip(self, start: int = 0, count: int | None = None) -> Tuple[np.ndarray, np.ndarray]:
        count = self.n if count is None else count
        fr, lb = zip(*(self.frame(start + i) for i in range(count)))
        return np.stack(fr), np.stack(lb)
