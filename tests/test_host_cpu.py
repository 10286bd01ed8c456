"""Host-side logic of the scheduler / ingest boundary that needs no GPU."""
import numpy as np
import pytest
import torch

from ams_amd import run as R
from ams_amd import utils as U


def test_to_size_host_path_matches_resize_helpers():
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (96, 192, 3), dtype=np.uint8)
    label = rng.integers(0, 19, (96, 192), dtype=np.uint8)
    f, l = R._to_size(frame, label, [64, 128])
    assert f.shape == (64, 128, 3) and l.shape == (64, 128)
    assert np.array_equal(f, U.resize_linear(frame, 128, 64))
    assert np.array_equal(l, U.resize_nearest(label, 128, 64))
    # frames that already have the network's size pass through untouched (same objects)
    f2, l2 = R._to_size(f, l, [64, 128])
    assert f2 is f and l2 is l


def test_batch_and_host_helpers_accept_arrays_and_tensors():
    a = np.zeros((4, 5, 3), np.uint8)
    t = torch.zeros((4, 5, 3), dtype=torch.uint8)
    assert R._batch1(a).shape == (1, 4, 5, 3) and tuple(R._batch1(t).shape) == (1, 4, 5, 3)
    assert isinstance(R._host(t), np.ndarray) and R._host(a) is a


def test_resize_edge_cases():
    img = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    assert np.array_equal(U.resize_linear(img, 2, 2), img)                    # identity
    up = U.resize_linear(img, 4, 4)
    assert up.shape == (4, 4, 3) and up.min() >= img.min() and up.max() <= img.max()
    lab = np.array([[1, 2], [3, 255]], np.uint8)
    assert np.array_equal(U.resize_nearest(lab, 4, 4)[::2, ::2], lab)


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a host without a GPU")
def test_ingest_has_no_cpu_fallback():
    from ams_amd.ingest import FrameIngest
    with pytest.raises(RuntimeError):
        FrameIngest()


def test_gpu_ingest_flag_is_an_extra_flag():
    flags = R.build_parser().parse_args(["--input_video", "synthetic:25-x", "--student_checkpoint", "synthetic", "--output_dir", "o",
                                         "--mode", "simple"])
    assert flags.gpu_ingest is False
    flags = R.build_parser().parse_args(["--input_video", "synthetic:25-x", "--student_checkpoint", "synthetic", "--output_dir", "o",
                                         "--mode", "simple", "--gpu_ingest"])
    assert flags.gpu_ingest is True
