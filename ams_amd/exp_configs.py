"""Per-video experiment tables: class subset, label-space size, test length.

Same query surface as the reference's exp_configs.py (``num_classes`` :8, ``class_weights`` :19,
``test_length`` :83 relative to its own numbering, ``coco_class_converter``, ``is_coco``) but kept
as one table instead of if/elif chains.  ``tests/test_helpers_golden.py`` checks every entry
(including which experiment numbers raise ``ValueError``) against values captured from the
reference (tests/golden/ref_helpers.json).
"""
import numpy as np

# experiment -> (label-space size, selected class ids, test length in seconds)
# 12-21 Outdoor Scenes, 22-24 A2D2, 25 Cityscapes (19 Cityscapes train-ids);
# 26-54 LVS (21 PASCAL-VOC ids, teacher labels arrive in COCO numbering).
_TABLE = {
    12: (19, (0, 1, 2, 8, 10, 11, 13), 900),
    13: (19, (2, 8, 9, 10, 11, 13), 420),
    14: (19, (0, 1, 2, 8, 10, 11), 810),
    15: (19, (0, 2, 8, 10, 11, 13), 900),
    17: (19, (0, 2, 8, 10, 11, 13), 900),
    19: (19, (1, 2, 8, 10, 11), 900),
    21: (19, (0, 8, 9, 10, 11), 800),
    22: (19, (0, 1, 2, 10, 11, 13), 520),
    23: (19, (0, 1, 2, 10, 11, 13), 900),
    24: (19, (0, 1, 2, 10, 11, 13), 740),
    25: (19, (0, 1, 2, 10, 11, 13), 2790),
    26: (21, (0, 15), 1000),
    27: (21, (0, 15), 1000),
    28: (21, (0, 15), 1200),
    29: (21, (0, 15), 1000),
    30: (21, (0, 15), 1000),
    31: (21, (0, 15), 1000),
    32: (21, (0, 15), 500),
    33: (21, (0, 15), 1000),
    34: (21, (0, 15), 1000),
    35: (21, (0, 15), 1000),
    36: (21, (0, 15), 1190),
    37: (21, (0, 15), 1000),
    39: (21, (0, 3), 600),
    40: (21, (0, 7, 12, 15), 1000),
    41: (21, (0, 13, 15), 1250),
    42: (21, (0, 15), 1000),
    43: (21, (0, 7, 15), 500),
    44: (21, (0, 15), 1000),
    45: (21, (0, 15), 500),
    46: (21, (0, 2, 15), 500),
    47: (21, (0, 7, 15), 1780),
    48: (21, (0, 7, 15), 1200),
    49: (21, (0, 7, 15), 1000),
    50: (21, (0, 2, 7, 15), 1000),
    51: (21, (0, 2, 7, 15), 1000),
    52: (21, (0, 7, 15), 1000),
    53: (21, (0, 2, 7, 15), 1000),
    54: (21, (0, 2, 7, 15), 1000),
}

# COCO (81 ids) -> PASCAL-VOC id for the classes LVS uses; everything else maps to background.
_COCO_TO_VOC = {1: 15, 2: 2, 3: 7, 15: 3, 17: 12, 18: 13}


def _row(experiment_number):
    try:
        return _TABLE[experiment_number]
    except KeyError:
        raise ValueError('Experiment %d not configured' % experiment_number) from None


def num_classes(experiment_number):
    return _row(experiment_number)[0]


def class_weights(experiment_number):
    """0/1 float32 column vector [num_classes, 1]; 1 marks a class the student is trained/scored on."""
    total, selected, _ = _row(experiment_number)
    w = np.zeros((total, 1), dtype=np.float32)
    w[list(selected), 0] = 1.0
    return w


def class_indices(experiment_number):
    return np.asarray(_row(experiment_number)[1], dtype=np.int64)


def test_length(experiment_number):
    return _row(experiment_number)[2]


def is_coco(experiment_number):
    return experiment_number in _TABLE and _TABLE[experiment_number][0] == 21


def coco_class_converter():
    conv = np.zeros(81, dtype=np.int32)
    for coco_id, voc_id in _COCO_TO_VOC.items():
        conv[coco_id] = voc_id
    return conv
