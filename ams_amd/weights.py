"""Weight I/O for the student: the reference's ``.npy`` dict format, a seeded synthetic initialiser,
and packing into the flat arenas the HIP engine trains in.

Reference format (utils/utils.py:20-49 ``SaveHelper``): ``np.save`` of a pickled dict
``{"<variable name>:0": float32 ndarray}``; ``restore_vars`` assigns every entry whose name survives the
caller's filter (SemanticNetwork.py:154-156 drops names containing 'Adam'/'Momentum', so optimizer state
is never restored).  The reference checkout ships no weight blob (.MISSING_LARGE_BLOBS), so benchmarks
and tests use ``synthetic_weights``; a real ``model.npy`` drops in through ``load_npy``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .spec import StudentSpec, build_spec


def synthetic_weights(spec: StudentSpec | None = None, seed: int = 0) -> Dict[str, np.ndarray]:
    """He-normal conv weights, gamma~U(0.5,1.5), beta~N(0,0.1), moving_mean~N(0,0.1), moving_var~U(0.5,1.5)."""
    spec = spec or build_spec()
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    stats = {v.name: v for v in spec.stats}
    for name in spec.all_variable_names():
        v = spec.by_name[name]
        if v.role == "weights":
            kh, kw, cin, cout = v.shape
            fan_in = kh * kw * (cin if not name.endswith("depthwise_weights:0") else 1)
            arr = rng.standard_normal(v.shape) * np.sqrt(2.0 / fan_in)
        elif v.role == "gamma":
            arr = rng.uniform(0.5, 1.5, v.shape)
        elif v.role in ("beta", "moving_mean"):
            arr = rng.standard_normal(v.shape) * 0.1
        elif v.role == "moving_variance":
            arr = rng.uniform(0.5, 1.5, v.shape)
        elif v.role == "biases":
            arr = rng.standard_normal(v.shape) * 0.1
        else:  # pragma: no cover
            raise AssertionError(v.role)
        out[name] = arr.astype(np.float32)
    assert set(stats) <= set(out)
    return out


def load_npy(path: str) -> Dict[str, np.ndarray]:
    """Read a reference-format checkpoint (pickled dict inside a .npy)."""
    if not path.endswith(".npy"):
        path = path + ".npy"
    data = np.load(path, allow_pickle=True).item()
    if not isinstance(data, dict):
        raise ValueError("%s does not hold a {name: ndarray} dict" % path)
    return data


def save_npy(path: str, variables: Dict[str, np.ndarray]) -> None:
    np.save(path, dict(variables))


def pack_trainable(spec: StudentSpec, variables: Dict[str, np.ndarray]) -> np.ndarray:
    flat = np.empty(spec.n_trainable, dtype=np.float32)
    for v in spec.trainable:
        a = np.asarray(variables[v.name], dtype=np.float32)
        if a.shape != v.shape:
            raise ValueError("%s: shape %s, expected %s" % (v.name, a.shape, v.shape))
        flat[v.offset:v.offset + v.size] = a.reshape(-1)
    return flat


def pack_stats(spec: StudentSpec, variables: Dict[str, np.ndarray]) -> np.ndarray:
    flat = np.empty(spec.n_stats, dtype=np.float32)
    for v in spec.stats:
        a = np.asarray(variables[v.name], dtype=np.float32)
        if a.shape != v.shape:
            raise ValueError("%s: shape %s, expected %s" % (v.name, a.shape, v.shape))
        flat[v.offset:v.offset + v.size] = a.reshape(-1)
    return flat


def unpack(spec: StudentSpec, trainable_flat: np.ndarray, stats_flat: np.ndarray) -> Dict[str, np.ndarray]:
    """Inverse of pack_*: dict in GraphDef variable order (the order ``get_vars`` reports)."""
    out: Dict[str, np.ndarray] = {}
    for name in spec.all_variable_names():
        v = spec.by_name[name]
        src = trainable_flat if v.trainable else stats_flat
        out[name] = np.array(src[v.offset:v.offset + v.size], dtype=np.float32).reshape(v.shape)
    return out
