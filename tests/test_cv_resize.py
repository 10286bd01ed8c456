"""cv2.resize restated (OpenCV is a third-party dependency of the reference that is absent here): the oracle restatement
(oracle/cv_resize.py) against hand-derived vectors, and the product's host resamplers (ams_amd/utils.py) against the oracle on
the sizes the scheduler meets.  tests/test_gpu_ingest.py holds the device kernel to the same oracle."""
import json

import numpy as np
import pytest

from ams_amd import utils as U
from oracle import cv_resize as CV


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.loads((golden_dir / "cv_resize_vectors.json").read_text())["cases"]


def test_oracle_and_product_reproduce_the_vectors(cases):
    assert sum(1 for c in cases if c.get("hand")) >= 6
    for c in cases:
        src = np.asarray(c["src"], np.uint8)
        want = np.asarray(c["dst"], np.uint8)
        if c["kind"] == "linear":
            assert np.array_equal(CV.resize_linear_u8(src, c["w"], c["h"]), want), (src.shape, c["w"], c["h"])
            assert np.array_equal(U.resize_linear(src, c["w"], c["h"]), want), (src.shape, c["w"], c["h"])
        else:
            assert np.array_equal(CV.resize_nearest_u8(src, c["w"], c["h"]), want)
            assert np.array_equal(U.resize_nearest(src, c["w"], c["h"]), want)


@pytest.mark.parametrize("src,dst", [((1208, 1920), (512, 1024)), ((1080, 1920), (256, 512)), ((1024, 2048), (512, 1024)), ((100, 200), (256, 512)),
                                     ((513, 1025), (512, 1024)), ((37, 53), (37, 53)), ((7, 5), (64, 128))])
def test_product_host_resize_equals_the_oracle(src, dst):
    rng = np.random.default_rng(src[0] * 7 + dst[1])
    img = rng.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    assert np.array_equal(U.resize_linear(img, dst[1], dst[0]), CV.resize_linear_u8(img, dst[1], dst[0]))
    lab = rng.integers(0, 20, src, dtype=np.uint8)
    assert np.array_equal(U.resize_nearest(lab, dst[1], dst[0]), CV.resize_nearest_u8(lab, dst[1], dst[0]))


def test_fixed_point_result_stays_within_one_level_of_exact_interpolation():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (60, 90, 3), dtype=np.uint8)
    fixed = CV.resize_linear_u8(img, 200, 130).astype(np.int64)
    exact = U.resize_linear(img.astype(np.float64), 200, 130)
    assert np.abs(fixed - exact).max() <= 1.0 + 1e-9
    # the 2x case is the box average, rounded half up
    half = CV.resize_linear_u8(img, 45, 30).astype(np.int64)
    box = img.astype(np.int64).reshape(30, 2, 45, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(half, (box + 2) >> 2)
