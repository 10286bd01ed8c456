#!/usr/bin/env python3
"""Generate tests/golden/ref_masks.json by running the reference's SemanticNetwork.get_train_mask.

Build-container only.  The method only needs ``self.coord_frac``, ``self.student['grad_masks_pl']``,
``self.saver.save_vars`` (for shapes) and ``self.train_vars_count``, so it is called on a stand-in object;
tensorflow / cv2 / termcolor / the ``ams`` package path are stubbed (none is used by this method).
Output (data only): per (strategy, fraction), under np.random.seed(123): for every trainable variable the
number of selected entries, plus a CRC of the packed mask bits and the generator state probe afterwards.
"""
import importlib.util
import json
import sys
import types
import zlib
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "ref_masks.json"
GRAPH = Path(__file__).resolve().parent / "student_graph_cityscapes.json"


class _Anything:
    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def main():
    np.bool = bool          # removed in NumPy 1.24+, the reference still uses it
    for name in ("tensorflow", "cv2", "termcolor", "ams", "ams.utils", "ams.utils.graph_utils", "ams.utils.utils"):
        m = types.ModuleType(name)
        m.__getattr__ = lambda attr: _Anything()       # any attribute resolves to a harmless stand-in
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_semnet", REF / "SemanticNetwork.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = mod.SemanticNetwork

    graph = json.loads(GRAPH.read_text())
    shapes = {v["name"]: tuple(v["shape"]) for v in graph["variables"]}
    names = graph["trainable_variables"]

    class Saver:
        def save_vars(self, sess, save_vars, map_fun, save_dir=None):
            return {n: np.zeros(shapes[n], dtype=np.float32) for n in names}

    out = {}
    for strategy in ("coord_desc_first", "coord_desc_last", "coord_desc_both", "coord_desc_rand"):
        for frac in (0.01, 0.02, 0.05, 0.1, 0.2):
            fake = types.SimpleNamespace(coord_frac=frac, student={"grad_masks_pl": {n: n for n in names}},
                                         saver=Saver(), sess=None, save_vars=None, filter=None, mask=None)
            fake.train_vars_count = lambda tm, _f=fake: cls.train_vars_count(_f, tm)
            np.random.seed(123)
            _before, mask = cls.get_train_mask(fake, strategy)
            counts = [int(np.sum(mask[n])) for n in names]
            bits = np.packbits(np.concatenate([mask[n].reshape(-1) for n in names]).astype(np.uint8))
            out["%s@%g" % (strategy, frac)] = {
                "counts": counts, "crc32": zlib.crc32(bits.tobytes()), "total": int(sum(counts)),
                "np_random_after": float(np.random.random())}
    # error behaviour
    errs = {}
    for strategy, frac in (("coord_desc_first", 0.3), ("bogus", 0.1), ("full_model", 0.1)):
        fake = types.SimpleNamespace(coord_frac=frac, student={"grad_masks_pl": {n: n for n in names}},
                                     saver=Saver(), sess=None, save_vars=None, filter=None, mask=None)
        fake.train_vars_count = lambda tm, _f=fake: cls.train_vars_count(_f, tm)
        try:
            r = cls.get_train_mask(fake, strategy)
            errs["%s@%g" % (strategy, frac)] = "ok:" + repr(r)
        except Exception as e:  # noqa: BLE001
            errs["%s@%g" % (strategy, frac)] = type(e).__name__
    out["errors"] = errs
    OUT.write_text(json.dumps(out, sort_keys=True))
    print(OUT, OUT.stat().st_size, errs)


if __name__ == "__main__":
    main()
