"""``SemanticNetwork`` surface on top of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Used by tests/golden/make_scheduler_fixture.py to run the ``ams_amd.run`` scheduler once on the CPU restatement (tiny clip, build
container) and commit the per-frame outputs, so that the scheduler LOOP — sampling times, restore-then-train-then-publish order,
per-frame metric outputs (reference run.py:78-461) — is pinned by something other than the HIP path itself.
Only the methods run.py calls are provided.  Arithmetic: oracle/student_torch.py in f32 (what the reference's TF1 CPU path
computes in); sampling: ``ams_amd.utils.mini_batch`` (pinned to the reference by tests/golden/ref_helpers.json), one call per
iteration exactly as SemanticNetwork.py:681-687 does; masks: ``ams_amd.coord_masks`` (pinned by ref_masks.json).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ams_amd import coord_masks
from ams_amd.semantic_network import FrozenGraph
from ams_amd.utils import calculate_miou, mini_batch
from ams_amd.weights import load_npy
from oracle.student_torch import StudentOracle, cross_miou_confusion


class OracleSemanticNetwork:
    TOTAL_CLASSES = 19

    def __init__(self, meta_dir, class_weights_exp=None, height=None, gpu_id='0', frozen=False, scale=None, mini_batch_size=None,
                 lr=None, mem_frac=1, coord_frac=0.1, cross_miou_compat=False, initial_variables=None, frozen_graph=None, **_kw):
        assert height is not None and class_weights_exp is not None
        self.TOTAL_CLASSES = class_weights_exp.shape[0]
        self.class_indices = np.where(np.asarray(class_weights_exp).reshape(-1) == 1)[0]
        self.height, self.frozen, self.lr = height, frozen, lr
        self.mini_batch_size, self.scale, self.coord_frac = mini_batch_size, scale, coord_frac
        if frozen:
            if frozen_graph is None:
                with open(meta_dir + ".pb", "rb") as f:
                    frozen_graph = FrozenGraph.ParseFromString(f.read())
            variables = frozen_graph.variables
        else:
            variables = initial_variables if initial_variables is not None else load_npy(meta_dir + ".npy")
            self._initial = variables
        self.oracle = StudentOracle(variables, self.class_indices, num_classes=self.TOTAL_CLASSES, dtype=torch.float32)
        self.train_params = self.curr_mask = self.mask = None
        self.last_losses = []

    def _mode(self):
        return "frozen" if self.frozen else "train"

    def restore_initial(self):
        self.oracle.restore(self._initial)           # model variables only: Adam slots and step count stay

    def predict_with_metric(self, frames, labels_teacher):
        lab, cm, loss = self.oracle.predict_with_metric(np.asarray(frames, np.float32), labels_teacher, self._mode())
        iou = calculate_miou(cm, nan=True)
        return lab, cm, iou, np.nanmean(iou), np.float32(loss)

    # the asynchronous surface of the product class (run.py --edge_pipeline): computed at once, handed out on collect
    def predict_with_metric_async(self, frames, labels_teacher):
        if not hasattr(self, "_pending"):
            self._pending, self._ticket = {}, 0
        self._ticket += 1
        self._pending[self._ticket] = self.predict_with_metric(frames, labels_teacher)
        return self._ticket

    def collect(self, ticket):
        return self._pending.pop(ticket)

    def predict_input(self, frames):
        return self.oracle.predict(np.asarray(frames, np.float32), self._mode())

    def calc_cross_miou(self, labels):
        cm = cross_miou_confusion(labels[0], labels[1], self.class_indices, self.TOTAL_CLASSES)
        iou = calculate_miou(cm, nan=True)
        return cm, iou, np.nanmean(iou)

    def train_with_deque(self, frame_deque, label_deque, num_of_iterations, train_strategy='full_model', keep_mask=False):
        assert not self.frozen, "Can't train frozen graph!!!"
        spec = self.oracle.spec
        mask: Optional[Dict[str, np.ndarray]] = None
        if train_strategy != 'full_model':
            assert train_strategy != 'coord_desc_auto', "the fixture generator does not cover coord_desc_auto"
            mask = coord_masks.build_mask(train_strategy, self.coord_frac, {v.name: v.shape for v in spec.trainable})
        # the batch-producer thread of the reference draws every batch of the phase through mini_batch, one call per iteration
        batches = [mini_batch(frame_deque, label_deque, [self.height, 2 * self.height], self.scale, self.mini_batch_size, 1, flip=False)
                   for _ in range(num_of_iterations)]
        self.last_losses = [float(self.oracle.train_step(ib[0].astype(np.float32), lb[0], self.lr, mask)) for ib, lb in batches]
        after = self.oracle.get_vars()
        if mask is not None:
            names = [v.name for v in spec.trainable]
            self.curr_mask = [np.asarray(mask[k]) for k in names]
            self.train_params = [after[k] for k in names]
        else:
            self.train_params = [after[k] for k in after]
            self.curr_mask = [np.ones_like(after[k], dtype=bool) for k in after]

    def delta_payload(self) -> bytes:
        payload = bytearray()
        for m in self.curr_mask:
            payload += np.packbits(m.flatten()).tobytes()
        for p_, m_ in zip(self.train_params, self.curr_mask):
            payload += p_[m_].astype(np.float16).tobytes()
        return bytes(payload)

    def get_frozen_graph(self):
        return FrozenGraph(self.oracle.get_vars(), self.class_indices, self.height, self.TOTAL_CLASSES)

    def save_to_frozen_graph(self, save_dir):
        with open(save_dir + ".pb", "wb") as f:
            f.write(self.get_frozen_graph().SerializeToString())

    def close_model(self):
        pass
