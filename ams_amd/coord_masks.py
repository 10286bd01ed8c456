"""Coordinate-descent parameter masks ("train only a fraction of the model", ICCV'21 AMS §model updates).

Data restatement of the strategy tables hard-coded in the reference (SemanticNetwork.py:302-669): for each
(strategy, fraction) a variable is fully trained when its name contains one of ``substrings`` or equals one
of ``names``; a few boundary tensors get a Bernoulli mask with a fixed probability so that the total hits
the requested fraction; everything else is frozen.  Draws come from the global NumPy generator in
``tf.trainable_variables()`` order with the same ``np.random.choice([True, False], size, p=[p, q])`` call the
reference makes, so a seeded run selects the same coordinates (tests/golden/ref_masks.json pins this).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

_M = "MobilenetV2/expanded_conv"


def _blocks(*idx):
    """'/Conv/' (stem) + '/expanded_conv/' + '/expanded_conv_<i>/' substrings."""
    out = ["/Conv/", "/expanded_conv/"]
    out += ["/expanded_conv_%d/" % i for i in idx]
    return out


def _wgb(scope):
    return [scope + "/weights:0", scope + "/BatchNorm/gamma:0", scope + "/BatchNorm/beta:0"]


_CP_BN = ["concat_projection/BatchNorm/gamma:0", "concat_projection/BatchNorm/beta:0"]
_CPW = "concat_projection/weights:0"

# (strategy suffix, fraction) -> (substrings, exact names, {name: (p_true, p_false)})
TABLES: Dict[Tuple[str, float], Tuple[list, list, dict]] = {
    ("last", 0.1): ([], ["aspp0/BatchNorm/gamma:0", "aspp0/BatchNorm/beta:0", _CPW] + _CP_BN +
                    ["logits/semantic/weights:0", "logits/semantic/biases:0"],
                    {"aspp0/weights:0": (0.90728, 0.09272)}),
    ("first", 0.1): (_blocks(1, 2, 3, 4, 5, 6, 7, 8), _wgb(_M + "_9/expand"),
                     {_M + "_9/depthwise/depthwise_weights:0": (0.25231, 0.74769)}),
    ("both", 0.1): (_blocks(1, 2, 3, 4, 5, 6) + ["logits/semantic/"],
                    _wgb(_M + "_7/expand") + [_M + "_7/depthwise/depthwise_weights:0"] + _CP_BN,
                    {_M + "_7/depthwise/BatchNorm/gamma:0": (0.80208, 0.19792), _CPW: (0.76490, 0.23510)}),
    ("last", 0.05): (["logits/semantic/"], list(_CP_BN), {_CPW: (0.76490, 0.23510)}),
    ("first", 0.05): (_blocks(1, 2, 3, 4, 5, 6), _wgb(_M + "_7/expand") + [_M + "_7/depthwise/depthwise_weights:0"],
                      {_M + "_7/depthwise/BatchNorm/gamma:0": (0.80208, 0.19792)}),
    ("both", 0.05): (_blocks(1, 2, 3, 4) + ["/expanded_conv_5/expand/", "/expanded_conv_5/depthwise/", "logits/semantic/"],
                     list(_CP_BN),
                     {_M + "_5/project/weights:0": (0.42285, 0.57715), _CPW: (0.36187, 0.63813)}),
    ("last", 0.01): (["logits/semantic/", "concat_projection/BatchNorm/"], [], {_CPW: (0.12005, 0.87995)}),
    ("first", 0.01): (_blocks(1, 2) + ["/expanded_conv_3/depthwise/", "/expanded_conv_3/expand/"], [],
                      {_M + "_3/project/weights:0": (0.00217, 0.99783)}),
    ("both", 0.01): (_blocks(1) + ["logits/semantic/", "concat_projection/BatchNorm/"],
                     [_M + "_2/expand/weights:0", _M + "_2/expand/BatchNorm/gamma:0"],
                     {_M + "_2/expand/BatchNorm/beta:0": (0.03472, 0.96528), _CPW: (0.03944, 0.96056)}),
    ("last", 0.2): (["logits/semantic/", "concat_projection/", "aspp0/", "image_pooling/",
                     _M + "_16/project/BatchNorm"], [], {_M + "_16/project/weights:0": (0.39270, 0.60730)}),
    ("first", 0.2): (_blocks(1, 2, 3, 4, 5, 6, 7, 8, 9, 10) + ["/expanded_conv_11/expand/", "/expanded_conv_11/depthwise/"],
                     [], {_M + "_11/project/weights:0": (0.97367, 0.02633)}),
    ("both", 0.2): (_blocks(1, 2, 3, 4, 5, 6, 7, 8) + ["concat_projection/", "aspp0/BatchNorm/", "logits/semantic/"],
                    _wgb(_M + "_9/expand"),
                    {_M + "_9/depthwise/depthwise_weights:0": (0.25231, 0.74769), "aspp0/weights:0": (0.90728, 0.09272)}),
    ("last", 0.02): (["logits/semantic/", "concat_projection/BatchNorm/"], [], {_CPW: (0.7187, 0.2813)}),
    ("first", 0.02): (_blocks(1, 2, 3, 4), [], {_M + "_5/expand/weights:0": (0.7367, 0.2633)}),
    ("both", 0.02): (_blocks(1, 2) + ["/expanded_conv_3/depthwise/", "/expanded_conv_3/expand/", "logits/semantic/",
                                      "concat_projection/BatchNorm/"], [],
                     {_M + "_3/project/weights:0": (0.00217, 0.99783), _CPW: (0.12005, 0.87995)}),
}


def build_mask(train_strategy: str, coord_frac: float, shapes: Dict[str, tuple]) -> Dict[str, np.ndarray]:
    """``shapes``: ordered dict variable name -> shape in trainable order.  Returns name -> bool ndarray."""
    if train_strategy == "coord_desc_rand":
        return {k: np.random.choice([True, False], size=s, p=[coord_frac, 1 - coord_frac]).astype(bool)
                for k, s in shapes.items()}
    if train_strategy not in ("coord_desc_first", "coord_desc_last", "coord_desc_both"):
        raise NameError('train_strategy %s is not implemented.' % train_strategy)
    key = (train_strategy[len("coord_desc_"):], coord_frac)
    if key not in TABLES:
        # the reference selects its tables by float equality; any other fraction ends in its final `else`
        raise NameError('train_strategy %s is not implemented.' % train_strategy)
    substrings, names, bern = TABLES[key]
    out = {}
    for k, s in shapes.items():
        if any(sub in k for sub in substrings) or k in names:
            out[k] = np.ones(s, dtype=bool)
        elif k in bern:
            out[k] = np.random.choice([True, False], size=s, p=list(bern[k])).astype(bool)
        else:
            out[k] = np.zeros(s, dtype=bool)
    return out
