"""Host side of `SemanticNetwork.train_with_deque` without a GPU: the sampler thread, the stager thread and the training loop of the
reference (SemanticNetwork.py:215-300, :679-704) with the device work replaced by stand-ins.  What is under test is the hand-over
machinery: more iterations than staging slots, and — the reference's flaw a drop-in must not inherit — a helper thread that dies must
surface as its exception from `train_with_deque`, promptly, with `process_lock` released and every thread joined."""
import threading
import time
from collections import deque

import numpy as np
import pytest
import torch

from ams_amd.semantic_network import SemanticNetwork

H, MB = 8, 2


class _Spec:
    trainable = []


class _Engine:
    device = "cpu"
    spec = _Spec()

    def __init__(self):
        self.steps = 0
        self.fail_at = None
        self.seen = []

    def train_step(self, frames, labels, lr, mask):
        if self.fail_at is not None and self.steps == self.fail_at:
            raise RuntimeError("engine failure at step %d" % self.steps)
        self.steps += 1
        self.seen.append(int(frames.to(torch.int64).sum()))
        return torch.tensor([float(self.steps), 1.0], dtype=torch.float64)


class _Net(SemanticNetwork):
    """The real train_with_deque / _train / _fill_batch / _fill_queue / _staging_slot over stand-ins for the three device touch points."""

    def __init__(self):               # the constructor builds a StudentEngine (GPU only): set the attributes the training path reads
        self.process_lock = threading.Lock()
        self.height, self.mini_batch_size, self.scale, self.lr = H, MB, [1], 1e-3
        self.frozen, self.mask, self.verbose, self.coord_frac = False, None, False, 0.1
        self.engine = _Engine()
        self.train_params = self.curr_mask = None
        self.last_losses = []
        self.stage_fail_at = None
        self.staged = 0

    def _model_vars(self):
        return {}

    def _make_copy_stream(self):
        return None

    def _stage_batch(self, batch, copy_stream):
        if self.stage_fail_at is not None and self.staged == self.stage_fail_at:
            raise OSError("H2D copy failed")
        self.staged += 1
        f = torch.from_numpy(np.array(batch['frames'], dtype=np.uint8))
        l = torch.from_numpy(np.array(batch['labels'], dtype=np.uint8))
        slot = batch.get('slot')
        if slot is not None:

            class _Done:
                def synchronize(self):
                    pass

            slot[2] = _Done()
        return f, l, None

    def _consume_staged(self, staged):
        return staged[0], staged[1]


def _memory(n=5, bad=None):
    rng = np.random.default_rng(0)
    frames = deque(rng.integers(0, 255, (H, 2 * H, 3), dtype=np.uint8) for _ in range(n))
    labels = deque(rng.integers(0, 19, (H, 2 * H), dtype=np.uint8) for _ in range(n))
    if bad is not None:
        frames[bad] = np.zeros((H - 1, 2 * H, 3), np.uint8)       # smaller than the crop: mini_batch asserts (reference utils/utils.py:156-157)
    return frames, labels


def _helpers_gone(before):
    deadline = time.time() + 2.0
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.01)
    return threading.active_count() <= before


def test_more_iterations_than_staging_slots():
    net = _Net()
    frames, labels = _memory()
    before = threading.active_count()
    np.random.seed(0)
    net.train_with_deque(frames, labels, 11)                      # the pinned ring has four slots
    assert net.engine.steps == 11 and net.last_losses == [float(i + 1) for i in range(11)]
    assert net.process_lock.acquire(False)
    net.process_lock.release()
    assert _helpers_gone(before)
    # every batch is the sampler's: same draws, same frames (the ring never hands a slot out before its copy was taken)
    np.random.seed(0)
    fl = list(frames)
    want = [int(sum(int(fl[np.random.choice(len(fl))].astype(np.int64).sum()) for _ in range(MB))) for _ in range(11)]
    assert net.engine.seen == want


def test_wrong_shaped_frame_raises_instead_of_hanging():
    net = _Net()
    frames, labels = _memory(bad=2)
    before = threading.active_count()
    t0 = time.time()
    with pytest.raises(AssertionError):
        net.train_with_deque(frames, labels, 50)
    assert time.time() - t0 < 1.0
    assert net.process_lock.acquire(False), "process_lock still held after a failed phase"
    net.process_lock.release()
    assert _helpers_gone(before)
    good_f, good_l = _memory()
    net.train_with_deque(good_f, good_l, 6)                        # the instance is usable afterwards (staging ring reset)
    assert len(net.last_losses) == 6


def test_stager_failure_surfaces():
    net = _Net()
    net.stage_fail_at = 3
    frames, labels = _memory()
    before = threading.active_count()
    t0 = time.time()
    with pytest.raises(OSError, match="H2D copy failed"):
        net.train_with_deque(frames, labels, 40)
    assert time.time() - t0 < 1.0 and net.engine.steps <= 3
    assert net.process_lock.acquire(False)
    net.process_lock.release()
    assert _helpers_gone(before)


def test_training_step_failure_stops_the_helpers():
    net = _Net()
    net.engine.fail_at = 2
    frames, labels = _memory()
    before = threading.active_count()
    with pytest.raises(RuntimeError, match="engine failure"):
        net.train_with_deque(frames, labels, 400)                  # the helpers would run 398 more batches into a bounded queue
    assert net.process_lock.acquire(False)
    net.process_lock.release()
    assert _helpers_gone(before)


def test_unknown_strategy_raises_name_error_with_lock_released():
    net = _Net()
    frames, labels = _memory()
    with pytest.raises(NameError):
        net.train_with_deque(frames, labels, 3, train_strategy='no_such_strategy')
    assert net.process_lock.acquire(False)
    net.process_lock.release()
