#!/usr/bin/env python3
"""Generate tests/golden/ref_helpers.json by IMPORTING the reference's pure-NumPy helpers.

Build-container only (reads /root/reference).  The reference imports
tensorflow / cv2 / termcolor at module import time; none is installed, and none of the
functions captured here touches them, so empty stub modules are enough.  Output = inputs
and expected outputs only (data), never reference source.

Captured:
  exp_configs.num_classes / class_weights / test_length / is_coco / coco_class_converter
  utils.calculate_miou   (seeded confusion matrices, incl. an absent class -> nan / string)
  utils.choose_frames
  utils.mini_batch       (scale=[1] path: pins RNG consumption order + dtype/shape)
  utils.colormap
  the take_array expression of SemanticNetwork.__init__ (SemanticNetwork.py:58-61)
"""
import importlib.util
import json
import random
import sys
import types
from collections import deque
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "ref_helpers.json"


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _jsonable(x):
    if isinstance(x, np.ndarray):
        return _jsonable(x.tolist())
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (np.floating, float)):
        return None if np.isnan(x) else float(x)
    if isinstance(x, (np.integer,)):
        return int(x)
    return x


def main():
    for name in ("tensorflow", "cv2", "termcolor"):
        _stub(name)
    exp = _load("ref_exp_configs", REF / "exp_configs.py")
    utils = _load("ref_utils", REF / "utils" / "utils.py")

    out = {}

    # ---- exp_configs tables -------------------------------------------------------------
    table = {}
    for n in range(0, 60):
        entry = {}
        try:
            entry["class_weights"] = exp.class_weights(n).reshape(-1).astype(int).tolist()
            entry["class_weights_shape"] = list(exp.class_weights(n).shape)
        except Exception as e:  # noqa: BLE001
            entry["class_weights_error"] = type(e).__name__
        try:
            entry["num_classes"] = int(exp.num_classes(n))
        except Exception as e:  # noqa: BLE001
            entry["num_classes_error"] = type(e).__name__
        try:
            entry["test_length"] = int(exp.test_length(n))
        except Exception as e:  # noqa: BLE001
            entry["test_length_error"] = type(e).__name__
        entry["is_coco"] = bool(exp.is_coco(n))
        table[str(n)] = entry
    out["exp_configs"] = table
    out["coco_class_converter"] = exp.coco_class_converter().tolist()

    # ---- calculate_miou -----------------------------------------------------------------
    rng = np.random.default_rng(1234)
    cases = []
    for k in (2, 6, 19):
        for trial in range(3):
            cm = rng.integers(0, 5000, size=(k, k)).astype(np.float64)
            if trial == 1:       # class never present nor predicted -> nan / 'Not predicted/present'
                cm[k // 2, :] = 0
                cm[:, k // 2] = 0
            if trial == 2:
                cm *= (rng.random((k, k)) > 0.6)
            case = {"cm": cm.tolist()}
            case["nan"] = _jsonable(utils.calculate_miou(cm, nan=True))
            case["plain"] = _jsonable(utils.calculate_miou(cm))
            iou, pop = utils.calculate_miou(cm, population=True, nan=True)
            case["population"] = _jsonable(pop)
            iou, fn, fp = utils.calculate_miou(cm, detailed=True, nan=True)
            case["false_neg"] = _jsonable(fn)
            case["false_pos"] = _jsonable(fp)
            cases.append(case)
    out["calculate_miou"] = cases

    # ---- choose_frames ------------------------------------------------------------------
    cf = []
    for n, frac in ((30, 0.1), (30, 1.0), (30, 0.5), (25, 0.2), (7, 0.34), (1, 1.0), (10, 0.0), (300, 0.033)):
        items = [(i, 1000 + i) for i in range(n)]
        frames, labels = utils.choose_frames(items, frac)
        cf.append({"n": n, "fraction": frac, "frames": frames, "labels": labels})
    out["choose_frames"] = cf

    # ---- mini_batch (scale=[1]) ---------------------------------------------------------
    mb = []
    for (seed, n_mem, h, b, iters, as_deque) in ((0, 5, 4, 3, 2, True), (7, 1, 2, 4, 1, False), (42, 9, 6, 10, 3, True)):
        w = 2 * h
        rs = np.random.RandomState(seed)
        frames = [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(n_mem)]
        labels = [rs.randint(0, 19, size=(h, w)).astype(np.uint8) for _ in range(n_mem)]
        np.random.seed(seed)
        random.seed(seed)
        fr = deque(frames) if as_deque else frames
        lb = deque(labels) if as_deque else labels
        imgs, lbls = utils.mini_batch(fr, lb, [h, w], [1], b, iters, flip=False)
        # which memory slot was drawn for every (iteration, sample)
        picks = [[int(next(i for i in range(n_mem) if np.array_equal(frames[i], imgs[it][j].astype(np.uint8))
                       and np.array_equal(labels[i], lbls[it][j].astype(np.uint8))))
                  for j in range(b)] for it in range(iters)]
        after_np = float(np.random.random())
        after_py = random.random()
        mb.append({"seed": seed, "n_mem": n_mem, "h": h, "w": w, "batch": b, "iters": iters, "deque": as_deque,
                   "img_dtype": str(imgs.dtype), "lbl_dtype": str(lbls.dtype),
                   "img_shape": list(imgs.shape), "lbl_shape": list(lbls.shape),
                   "picks": picks, "img_sum": float(imgs.sum()), "lbl_sum": float(lbls.sum()),
                   "np_random_after": after_np, "py_random_after": after_py})
    out["mini_batch"] = mb

    # ---- colormap + take_array ----------------------------------------------------------
    out["colormap_cityscapes"] = utils.colormap().tolist()
    ta = {}
    for n in (12, 13, 19, 21, 25, 26, 40):
        cw = exp.class_weights(n)
        total = cw.shape[0]
        take = np.cumsum(cw).reshape(total) * cw.reshape(total)
        take = np.where(take != 0, take - 1, take).astype(int)
        ta[str(n)] = {"take_array": take.tolist(), "class_indices": np.where(cw == 1)[0].tolist()}
    out["take_array"] = ta

    OUT.write_text(json.dumps(out, sort_keys=True))
    print(OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
