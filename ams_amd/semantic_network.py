"""``SemanticNetwork`` — the drop-in boundary of the AMS hot path on MI355X.

Mirrors the public surface of the reference class (SemanticNetwork.py:24-755): constructor arguments,
method names, argument meaning, return types, assertion behaviour and the process-wide lock, so that the
reference's edge/server loop (run.py) can use it unchanged.  Underneath, one ``StudentEngine`` (HIP) per
instance replaces the tf.Session; there is no CPU fallback.

Deliberate differences (all listed in INTEGRATION.md):
  * ``<meta_dir>.pb`` written by ``save_to_frozen_graph`` is an ``AMSF`` container (weights + statistics),
    not a TensorFlow GraphDef: the hand-off semantic (inference-mode BN with eps 1e-3 everywhere, trained
    gamma/beta, moving statistics; reference utils/graph_utils.py:52-126) is the same, the bytes are not.
  * ``infer`` / ``train_step`` aliases are added (BASELINE.json north-star names).
  * per-iteration loss printing is off unless ``verbose=True`` (each print forces a device sync).
"""
from __future__ import annotations

import io
import os
import random
import threading
import time
from collections import deque
from typing import Dict, List, Optional

import numpy as np
import torch

from . import coord_masks, hip
from .engine import StudentEngine
from .utils import calculate_miou, colormap, mini_batch
from .weights import load_npy

FROZEN_MAGIC = b"AMSF\x01"


class FrozenGraph:
    """What the server ships to the edge: every model variable after training (reference: a GraphDef with the
    variables folded to constants and BN rebound to inference mode, utils/graph_utils.py:79-126)."""

    def __init__(self, variables: Dict[str, np.ndarray], class_indices, height: int, num_classes: int):
        self.variables = variables
        self.class_indices = [int(c) for c in class_indices]
        self.height = int(height)
        self.num_classes = int(num_classes)

    def SerializeToString(self) -> bytes:
        buf = io.BytesIO()
        np.savez(buf, __class_indices=np.asarray(self.class_indices, dtype=np.int32),
                 __height=np.asarray(self.height), __num_classes=np.asarray(self.num_classes),
                 **{k.replace("/", "|"): v for k, v in self.variables.items()})
        return FROZEN_MAGIC + buf.getvalue()

    @staticmethod
    def ParseFromString(data: bytes) -> "FrozenGraph":
        if not data.startswith(FROZEN_MAGIC):
            raise ValueError("not an AMSF frozen student (TensorFlow .pb files cannot be loaded by this build)")
        z = np.load(io.BytesIO(data[len(FROZEN_MAGIC):]))
        variables = {k.replace("|", "/"): z[k] for k in z.files if not k.startswith("__")}
        return FrozenGraph(variables, z["__class_indices"].tolist(), int(z["__height"]), int(z["__num_classes"]))


class SemanticNetwork(object):
    OPT_FILTER = ['Adam', 'Momentum']
    OP_FILTER = ['image_cache:0', 'global_step:0']
    THREAD_SLEEP_INTERVAL = 1 / 1000.
    TOTAL_CLASSES = 19
    WHITE = np.array([255, 255, 255], dtype=np.uint8)
    BLACK = np.array([0, 0, 0], dtype=np.uint8)

    def __init__(self, meta_dir, class_weights_exp=None, height=None, gpu_id='0', frozen=False,
                 scale=None, mini_batch_size=None, lr=None, mem_frac=1, coord_frac=0.1, cross_miou_compat=False,
                 filter_out=None, over_ride_total_classes=None, **kwargs):
        assert height is not None, "No height is given"
        assert class_weights_exp is not None, "No class weights specified"
        assert frozen or None not in [scale, mini_batch_size, lr], "Training parameters must be specified for " \
                                                                   "non-frozen graph"
        self.lr = lr
        self.mini_batch_size = mini_batch_size
        self.scale = scale
        if over_ride_total_classes is not None:
            self.TOTAL_CLASSES = over_ride_total_classes
        self.coord_frac = coord_frac

        self.class_weights_graph = class_weights_exp
        self.class_indices_graph = np.where(self.class_weights_graph == 1)[0]
        assert self.class_weights_graph.shape == (self.TOTAL_CLASSES, 1)
        self.class_count = len(self.class_indices_graph)
        assert self.class_indices_graph.shape == (self.class_count,)
        assert self.class_count > 0
        self.cross_miou_compat = cross_miou_compat

        self.color_map_reduced_ = np.take(colormap(), self.class_indices_graph, axis=0)
        ranks = np.cumsum(self.class_weights_graph).reshape(self.TOTAL_CLASSES) * \
            self.class_weights_graph.reshape(self.TOTAL_CLASSES)
        self.take_array = np.where(ranks != 0, ranks - 1, ranks).astype(int)
        assert self.take_array.shape == (self.TOTAL_CLASSES,)

        self.frozen = frozen
        self.height = height
        assert self.height > 0
        self.meta_dir = meta_dir
        self.process_lock = threading.Lock()
        self.verbose = bool(kwargs.pop("verbose", False))
        # kwargs the reference forwards to create_student_v3 (graph_utils.py:338-339); only the ones run.py can
        # switch on are meaningful here
        self.masked_gradients = bool(kwargs.pop("masked_gradients", False))
        for dead in ("threshold", "map_misc", "test_mode"):
            kwargs.pop(dead, None)
        # create_student_v3's remaining kwargs (utils/graph_utils.py:338-339, 403-404, 451-456); run.py:150 leaves them off
        self.train_biases_only = bool(kwargs.pop("train_biases_only", False))
        self.regularize = bool(kwargs.pop("regularize", False))
        self.soft_teacher = bool(kwargs.pop("soft_teacher", False))
        if frozen:                 # the reference's frozen branch never calls create_student_v3: the kwargs are accepted and unused there too
            self.soft_teacher = self.regularize = self.train_biases_only = False
        initial_variables = kwargs.pop("initial_variables", None)
        frozen_graph = kwargs.pop("frozen_graph", None)
        max_batch = kwargs.pop("max_batch", None)
        # frozen only: depth of the asynchronous single-call pipeline (predict_with_metric_async / collect); 1 = off
        self.pipeline_depth = int(kwargs.pop("pipeline_depth", 1))
        assert 1 <= self.pipeline_depth <= 4, "pipeline_depth must be 1 .. 4"
        assert not kwargs, "unknown arguments: %s" % sorted(kwargs)

        # gpu_id is the reference's visible_device_list string (SemanticNetwork.py:74): an ordinal among the devices this process
        # can see.  An ordinal that does not exist is an error, as it is for tf.ConfigProto (no silent fallback to device 0);
        # mem_frac (per_process_gpu_memory_fraction, :73) has no counterpart: the engine allocates exactly its arena.
        device = "cuda:%d" % int(str(gpu_id).split(",")[0]) if not str(gpu_id).startswith("cuda") else str(gpu_id)
        if torch.cuda.is_available() and int(device.split(":")[1]) >= torch.cuda.device_count():
            raise ValueError("gpu_id %r: this process sees %d GPU(s) (ordinals are relative to the visible devices, "
                             "cf. HIP_VISIBLE_DEVICES)" % (gpu_id, torch.cuda.device_count()))
        assert 0 < float(mem_frac) <= 1, "mem_frac must be in (0, 1]"
        self._mem_frac = float(mem_frac)
        if self.frozen:
            if frozen_graph is None:
                with open(meta_dir + ".pb", 'rb') as pb_file:
                    frozen_graph = FrozenGraph.ParseFromString(pb_file.read())
            self._call_batch = int(max_batch or 1)
            self.engine = StudentEngine(self.class_indices_graph, self.height, 2 * self.height,
                                        max_batch=self._call_batch * self.pipeline_depth, trainable=False,
                                        num_classes=self.TOTAL_CLASSES, device=device)
            self.engine.load_variables(frozen_graph.variables)
            self.engine.freeze()
            self._queued = []              # (ticket, frame, label) not yet launched
            self._pending = []             # tickets of the pass that is running on the GPU (its results are still on the device)
            self._ready = {}               # ticket -> result, after a pass was fetched
            self._tickets = 0
        else:
            self.engine = StudentEngine(self.class_indices_graph, self.height, 2 * self.height,
                                        max_batch=int(max_batch or max(int(mini_batch_size), 1)), trainable=True,
                                        num_classes=self.TOTAL_CLASSES, device=device)
            if filter_out is not None:
                self.OPT_FILTER = list(self.OPT_FILTER) + list(filter_out)
            self.filter = lambda elem: elem if all(
                keyword not in elem for keyword in self.OPT_FILTER) and elem not in self.OP_FILTER else None
            self._initial = initial_variables if initial_variables is not None else load_npy("%s.npy" % self.meta_dir)
            self._restore_dict(self._initial)
            if self.soft_teacher:
                self.engine.set_soft_teacher(True)
            if self.regularize:
                self.engine.set_regularizer(True, biases_only=self.train_biases_only)
            self.mask = None
            self.train_params = None
            self.curr_mask = None
            self.last_losses: List[float] = []
        self._last_train_ms = 0.0
        # mem_frac (tf.ConfigProto per_process_gpu_memory_fraction, SemanticNetwork.py:73): TensorFlow refuses allocations past that share of
        # the device; the engine allocates exactly one arena, so the cap is checked once, against it
        if torch.cuda.is_available():
            total = torch.cuda.get_device_properties(self.engine.device).total_memory
            if self.engine.arena_bytes > self._mem_frac * total:
                need = self.engine.arena_bytes
                self.engine.close()
                raise MemoryError("the student's arena (%.2f GB) exceeds mem_frac = %g of the device's %.1f GB" % (need / 1e9, self._mem_frac, total / 1e9))

    # ------------------------------------------------------------------ variables (SaveHelper semantics)
    def _restore_dict(self, variables: Dict[str, np.ndarray]) -> None:
        kept = {k: v for k, v in variables.items() if self.filter(k) is not None}
        self.engine.load_variables(kept)

    def restore_initial(self):
        """Reload ``<meta_dir>.npy``; optimizer state (Adam moments, step count) is NOT reset."""
        self._restore_dict(self._initial)

    def restore(self, chk):
        if isinstance(chk, str):
            chk = load_npy(chk)
        elif not isinstance(chk, dict):
            raise SystemExit(1)
        self._restore_dict(chk)

    def get_vars(self):
        out = self.engine.get_variables()
        if not self.frozen:
            m, v = self.engine.adam_m.cpu().numpy(), self.engine.adam_v.cpu().numpy()
            for var in self.engine.spec.trainable:
                stem = var.name[:-2]
                out[stem + "/Adam:0"] = m[var.offset:var.offset + var.size].reshape(var.shape).copy()
                out[stem + "/Adam_1:0"] = v[var.offset:var.offset + var.size].reshape(var.shape).copy()
            t = self.engine.adam_step
            out["beta1_power:0"] = np.float32(0.9 ** (t + 1))
            out["beta2_power:0"] = np.float32(0.999 ** (t + 1))
        return out

    def _model_vars(self) -> Dict[str, np.ndarray]:
        return self.engine.get_variables()

    # ------------------------------------------------------------------ inference
    def _mode(self) -> int:
        return hip.MODE_FROZEN if self.frozen else hip.MODE_LIVE

    def _drain_async(self):
        """A synchronous call shares the engine's one output block with the asynchronous edge pipeline: a pass that is still on the GPU is
        fetched first, so that ``collect`` later returns ITS metrics and not this call's."""
        if self.frozen and self._pending:
            self._fetch_pending()

    def predict_input(self, frames):
        self.process_lock.acquire()
        try:
            self._drain_async()
            labels_ = self.engine.predict_host(frames, self._mode())
            assert labels_.shape == tuple(frames.shape[:-1] if hasattr(frames, 'shape') else np.shape(frames)[:-1])
        finally:
            self.process_lock.release()
        return labels_

    infer = predict_input

    def calc_cross_miou(self, labels):
        assert not self.frozen or self.cross_miou_compat
        assert labels.shape == (2, self.height, 2 * self.height)
        self.process_lock.acquire()
        try:
            conf_mat_ = self.engine.cross_confusion(labels).cpu().numpy().astype(np.float64)
            iou_ = calculate_miou(conf_mat_, nan=True)
            miou_ = np.nanmean(iou_)
        finally:
            self.process_lock.release()
        return conf_mat_, iou_, miou_

    def predict_with_metric(self, frames, labels_teacher):
        self.process_lock.acquire()
        try:
            self._drain_async()
            # one device -> host copy for labels + confusion matrix + loss (they share one preallocated output block)
            labels_student, conf_i64, ls = self.engine.predict_with_metric_host(frames, labels_teacher, self._mode())
            conf_mat_ = conf_i64.astype(np.float64)
            loss_ = np.float32(ls[0] / ls[1]) if ls[1] > 0 else np.float32(np.nan)
            assert labels_student.shape == tuple(frames.shape[:-1] if hasattr(frames, 'shape') else np.shape(frames)[:-1])
            iou_ = calculate_miou(conf_mat_, nan=True)
            miou_ = np.nanmean(iou_)
        finally:
            self.process_lock.release()
        return labels_student, conf_mat_, iou_, miou_, loss_

    # The edge's per-frame call, pipeline_depth frames at a time (an addition: the reference's call is synchronous).  A one-frame forward is
    # ~45 dependent launches that leave most of the chip idle (0.49 ms); two frames in one pass take 0.59 ms, three 0.68 ms.  Submitted
    # frames wait until pipeline_depth of them are there (or until one of them is collected), then run as ONE pass with per-frame metrics
    # (ams_student_predict_frames).  Each frame's result is what predict_with_metric returns for it, bit for bit.
    def predict_with_metric_async(self, frames, labels_teacher):
        assert self.frozen and self.pipeline_depth > 1, "construct the frozen network with pipeline_depth >= 2"
        assert np.shape(frames)[0] == 1 and np.shape(labels_teacher)[0] == 1, "one frame per call"
        with self.process_lock:
            self._tickets += 1
            self._queued.append((self._tickets, frames, labels_teacher))
            if len(self._queued) >= self.pipeline_depth:
                self._launch_queued()
            return self._tickets

    def _fetch_pending(self):
        if not self._pending:
            return
        labs, confs, losses = self.engine.fetch_frames()          # one device -> host copy, one synchronisation for the whole pass
        for k, t in enumerate(self._pending):
            conf_mat_ = confs[k].astype(np.float64)
            ls = losses[k]
            loss_ = np.float32(ls[0] / ls[1]) if ls[1] > 0 else np.float32(np.nan)
            iou_ = calculate_miou(conf_mat_, nan=True)
            self._ready[t] = (labs[k:k + 1], conf_mat_, iou_, np.nanmean(iou_), loss_)
        self._pending = []

    def _launch_queued(self):
        self._fetch_pending()             # the engine has one output block: the previous pass leaves it before the next one writes it
        cat = (lambda xs: torch.cat(list(xs))) if hasattr(self._queued[0][1], "unsqueeze") else (lambda xs: np.concatenate([np.asarray(x) for x in xs]))
        frames = cat(q[1] for q in self._queued)
        labels = cat(q[2] for q in self._queued)
        self._pending = [q[0] for q in self._queued]
        self._queued = []
        self.engine.predict_frames(frames, labels, self._mode(), u8=True)     # returns at once: the pass runs while the caller goes on (labels leave as uint8)

    def collect(self, ticket):
        with self.process_lock:
            if ticket not in self._ready:
                if ticket not in self._pending:
                    assert any(q[0] == ticket for q in self._queued), "unknown ticket"
                    self._launch_queued()
                self._fetch_pending()
            return self._ready.pop(ticket)

    # ------------------------------------------------------------------ training
    def train_with_deque(self, frame_deque, label_deque, num_of_iterations, train_strategy='full_model',
                         keep_mask=False, teacher_logits_deque=None):
        """``teacher_logits_deque`` (soft_teacher=True only; the reference's _train never feeds teacher_labels_logits_pl, so its soft graph cannot
        run through this method at all): the cached teacher logits of the replay memory, one f32 [th, tw, TOTAL_CLASSES] array per frame of
        ``frame_deque``; a mini-batch takes the logits of the frames it drew."""
        assert not self.frozen, "Can't train frozen graph!!!"
        assert (teacher_logits_deque is not None) == bool(getattr(self, "soft_teacher", False)), \
            "teacher_logits_deque goes with soft_teacher=True (and is required then)"
        if teacher_logits_deque is not None:
            assert len(teacher_logits_deque) == len(frame_deque), "one teacher-logit array per frame of the replay memory"
        if not keep_mask:
            self.mask = None
        self.process_lock.acquire()
        # The reference's helper threads (SemanticNetwork.py:222-231, :679-704) have no failure path: if one dies — a frame of the wrong shape
        # trips the assert in _fill_batch — _train polls its deque forever with process_lock held.  Here every helper hands its exception
        # over through `ctx`, every wait loop watches ctx['abort'], and the caller gets the helper's exception with the lock released.
        ctx = self._helpers = {"abort": threading.Event(), "error": None}
        batch_thr = None
        try:
            batch_deque = deque()
            batch_thr = threading.Thread(target=self._guarded, args=(ctx, self._fill_batch, batch_deque, frame_deque, label_deque,
                                                                     num_of_iterations, teacher_logits_deque))
            batch_thr.start()
            self._train(batch_deque, num_of_iterations, train_strategy)
        finally:
            ctx["abort"].set()                      # after a clean phase the helpers have returned already; after an error this stops them
            if batch_thr is not None:
                batch_thr.join()
            fill_thr = ctx.pop("fill_thr", None)
            if fill_thr is not None:
                fill_thr.join()
            self._helpers = None
            self._release_staging()
            self.process_lock.release()
        if ctx["error"] is not None:
            raise ctx["error"]

    @staticmethod
    def _guarded(ctx, fn, *args):
        """Body of a helper thread: the first exception of any helper is kept for the caller and stops the others."""
        try:
            fn(*args)
        except BaseException as e:  # noqa: BLE001  (handed to the calling thread, which re-raises it)
            if ctx["error"] is None:
                ctx["error"] = e
            ctx["abort"].set()

    def _aborted(self):
        h = getattr(self, "_helpers", None)
        return h is not None and h["abort"].is_set()

    class _Aborted(Exception):
        """a wait was cut short because another helper of the same call failed"""

    def train_step(self, frames, labels_teacher, train_strategy='full_model', teacher_logits=None):
        """North-star alias: ONE optimisation step on an explicit batch; returns the loss (float).  With ``soft_teacher=True`` the cached teacher
        logits of the batch are fed as ``teacher_logits`` f32 [B, th, tw, TOTAL_CLASSES] — what a caller of the reference puts into
        ``feed_dict[student['teacher_labels_logits_pl']]`` (th x tw = the label size, or a smaller cached grid: include/ams_hip.h)."""
        assert not self.frozen, "Can't train frozen graph!!!"
        assert (teacher_logits is not None) == self.soft_teacher, "teacher_logits go with soft_teacher=True (and are required then)"
        with self.process_lock:
            mask_dev = None
            if 'coord_desc_' in train_strategy:
                _before, train_mask_ = self.get_train_mask(train_strategy)
                mask_dev = self._mask_to_device(train_mask_)
            ls = self.engine.train_step(frames, labels_teacher, self.lr, mask_dev, teacher_logits=teacher_logits).cpu().numpy()
        return float(ls[0] / ls[1]) if ls[1] > 0 else float("nan")

    def _mask_to_device(self, train_mask_: Dict[str, np.ndarray]) -> torch.Tensor:
        flat = np.empty(self.engine.spec.n_trainable, dtype=np.uint8)
        for v in self.engine.spec.trainable:
            flat[v.offset:v.offset + v.size] = np.asarray(train_mask_[v.name]).reshape(-1)
        return torch.from_numpy(flat).to(self.engine.device)

    def _train(self, batch_deque, num_of_iterations, train_strategy):
        signal_deque = deque()
        ctx = getattr(self, "_helpers", None) or {"abort": threading.Event(), "error": None}
        fill_thr = threading.Thread(target=self._guarded, args=(ctx, self._fill_queue, batch_deque, num_of_iterations, signal_deque))
        ctx["fill_thr"] = fill_thr                  # joined by train_with_deque's finally, whatever happens below
        fill_thr.start()

        _before, train_mask_ = self.get_train_mask(train_strategy)
        mask_dev = self._mask_to_device(train_mask_) if train_mask_ is not None else None
        losses = []
        t_phase = time.time()
        for it in range(num_of_iterations):
            staged = None
            while staged is None:
                try:
                    staged = signal_deque.popleft()
                except IndexError:
                    if ctx["abort"].is_set():       # a helper died: its exception is raised by train_with_deque
                        return
                    time.sleep(self.THREAD_SLEEP_INTERVAL)
            t1 = time.time()
            frames_dev, labels_dev = self._consume_staged(staged)
            logits_dev = staged[3] if len(staged) > 3 else None
            if logits_dev is not None:
                logits_dev.record_stream(torch.cuda.current_stream(self.engine.device))
            if logits_dev is not None:
                loss_dev = self.engine.train_step(frames_dev, labels_dev, self.lr, mask_dev, teacher_logits=logits_dev)
            else:
                loss_dev = self.engine.train_step(frames_dev, labels_dev, self.lr, mask_dev)
            losses.append(loss_dev)
            if self.verbose:
                ls = loss_dev.cpu().numpy()
                print('Loss is %.3f at iteration %d and took %.1f ms' % (ls[0] / max(ls[1], 1), it,
                                                                         (time.time() - t1) * 1000.0))
            if train_strategy == 'coord_desc_auto':
                if it == 0 and self.mask is None:
                    # derive the mask from the first step's |delta w|: keep the top coord_frac, roll back the rest
                    _after = self._model_vars()
                    names = [v.name for v in self.engine.spec.trainable]
                    changes = np.concatenate([np.abs(_after[k] - _before[k]).reshape(-1) for k in names], axis=0)
                    cut_threshold = np.percentile(changes, 100 * (1 - self.coord_frac))
                    _combine = {}
                    kept = total = 0
                    for k in names:
                        train_mask_[k] = np.abs(_after[k] - _before[k]) > cut_threshold
                        kept += int(np.sum(train_mask_[k]))
                        total += train_mask_[k].size
                        _combine[k] = np.where(train_mask_[k], _after[k], _before[k])
                    if self.verbose:
                        print("Using auto mode, Training %.3f%% of variables" % (100 * kept / total))
                    self._restore_dict(_combine)
                    self.mask = train_mask_
                    mask_dev = self._mask_to_device(train_mask_)
        fill_thr.join()
        if ctx["error"] is not None:
            return
        stacked = torch.stack(losses).cpu().numpy() if losses else np.zeros((0, 2))
        self.last_losses = [float(s / c) if c > 0 else float("nan") for s, c in stacked]
        self._last_train_ms = (time.time() - t_phase) * 1000.0

        _after_train = self._model_vars()
        if 'coord_desc_' in train_strategy:
            names = [v.name for v in self.engine.spec.trainable]
            self.curr_mask = [np.asarray(train_mask_[k]) for k in names]
            self.train_params = [_after_train[k] for k in names]
        else:
            self.train_params = [_after_train[k] for k in _after_train.keys()]
            self.curr_mask = [np.ones_like(_after_train[k], dtype=bool) for k in _after_train.keys()]

    def _consume_staged(self, staged):
        """Make the compute stream wait for a staged batch's copy; returns its device tensors."""
        frames_dev, labels_dev, ready = staged[:3]
        compute = torch.cuda.current_stream(self.engine.device)
        compute.wait_event(ready)
        # the buffers were allocated on the stager's copy stream: tell the caching allocator that the compute stream uses
        # them too, or dropping the references one iteration later hands the block back to the copy stream's pool while
        # this step's kernels (the stem weight gradient re-reads the frames in backward) are still queued
        frames_dev.record_stream(compute)
        labels_dev.record_stream(compute)
        return frames_dev, labels_dev

    def delta_payload(self) -> bytes:
        """The downlink model delta of reference run.py:316-336 as bytes: per variable ``np.packbits(mask.flatten())``, then
        per variable the masked parameters as fp16.  Under the coordinate-descent strategies (``train_params`` = the
        trainable variables, in arena order) the value part is gathered and cast on the device by ``ams_pack_masked_fp16``;
        otherwise (``full_model``: every model variable incl. BN statistics) it is the reference's host loop."""
        assert self.curr_mask is not None and self.train_params is not None, "no training phase has run yet"
        payload = bytearray()
        for val in self.curr_mask:
            payload += np.packbits(val.flatten()).tobytes()
        trainable = self.engine.spec.trainable
        on_device = len(self.curr_mask) == len(trainable) and all(m.size == v.size for m, v in zip(self.curr_mask, trainable))
        if on_device:
            flat = np.concatenate([m.reshape(-1) for m in self.curr_mask]).astype(np.uint8)
            self.process_lock.acquire()
            try:
                halves = self.engine.pack_masked_fp16(torch.from_numpy(flat).to(self.engine.device))
                payload += halves.cpu().numpy().tobytes()
            finally:
                self.process_lock.release()
        else:
            for p_, m_ in zip(self.train_params, self.curr_mask):
                assert p_.shape == m_.shape
                payload += p_[m_].astype(np.float16).tobytes()
        return bytes(payload)

    def get_train_mask(self, train_strategy):
        """Coordinate-descent masks (SemanticNetwork.py:302-669): dict variable name -> bool array, or None."""
        if train_strategy == 'full_model':
            return None, None
        _before = {v.name: None for v in self.engine.spec.trainable}
        shapes = {v.name: v.shape for v in self.engine.spec.trainable}
        if train_strategy == 'coord_desc_auto':
            _before = {k: v for k, v in self._model_vars().items() if k in shapes}
            if self.mask is None:
                train_mask_ = {k: np.ones(shapes[k], dtype=bool) for k in shapes}
            else:
                train_mask_ = self.mask
            return _before, train_mask_
        train_mask_ = coord_masks.build_mask(train_strategy, self.coord_frac, shapes)   # raises NameError if unknown
        if self.verbose:
            all_vars, train_vars_len = self.train_vars_count(train_mask_)
            print("Using %s mode, Training %.3f%% of variables" % (train_strategy, 100 * train_vars_len / all_vars))
        return _before, train_mask_

    def train_vars_count(self, train_mask_):
        all_vars = sum(m.size for m in train_mask_.values())
        train_vars_len = sum(int(np.sum(m)) for m in train_mask_.values())
        return all_vars, train_vars_len

    def _fill_batch(self, batch_deque, frame_deque, label_deque, number_of_batches, teacher_logits_deque=None):
        """Producer thread: sample mini-batches from the replay memory (utils.mini_batch contract).

        Fast path (the only one run.py exercises: scale == [1], frames already at network size): draws the same
        random numbers in the same order as ``mini_batch`` but gathers the uint8 frames directly instead of
        materialising float64 copies (B x 12.6 MB per batch at 512x1024)."""
        frames = list(frame_deque) if isinstance(frame_deque, deque) else frame_deque
        labels = list(label_deque) if isinstance(label_deque, deque) else label_deque
        crop = [self.height, self.height * 2]
        fast = (list(self.scale) == [1] and all(f.shape[:2] == tuple(crop) for f in frames))
        fast = fast and all(f.dtype == np.uint8 for f in frames) and all(l.dtype == np.uint8 and l.shape == tuple(crop) for l in labels)
        soft = list(teacher_logits_deque) if teacher_logits_deque is not None else None
        # soft targets follow the frames a batch drew: only where frames are taken as they are (no rescale / crop of the logits is defined)
        assert soft is None or fast, "teacher_logits_deque needs uint8 frames and labels at the network size and scale == [1]"
        for _ in range(number_of_batches):
            if self._aborted():
                return
            slot = None
            if fast:
                picks = []
                for _j in range(self.mini_batch_size):
                    picks.append(np.random.choice(len(frames)))
                    random.randint(0, 0)      # scale choice
                    random.randint(0, 0)      # row offset  (slack is 0 when the frame already has the crop size)
                    random.randint(0, 0)      # column offset
                # gathered straight into a pinned staging slot: one host copy per frame, none per batch (a fresh pin_memory() per batch
                # costs a page-lock of 12-16 MB each time)
                try:
                    slot = self._staging_slot()
                except self._Aborted:
                    return
                image_batch, label_batch = slot[0].numpy(), slot[1].numpy()
                for j, p in enumerate(picks):
                    image_batch[j] = frames[p]
                    label_batch[j] = labels[p]
            else:
                ib, lb = mini_batch(frames, labels, crop, self.scale, self.mini_batch_size, 1, flip=False)
                image_batch, label_batch = ib[0], lb[0]
            assert np.shape(label_batch) == (self.mini_batch_size, self.height, self.height * 2)
            assert np.shape(image_batch) == (self.mini_batch_size, self.height, self.height * 2, 3)
            batch = {'frames': image_batch, 'labels': label_batch, 'slot': slot}
            if soft is not None:
                batch['teacher_logits'] = np.stack([np.asarray(soft[p], dtype=np.float32) for p in picks])
            batch_deque.append(batch)

    def _staging_slot(self):
        """Next slot of a small ring of pinned host buffers [mini_batch, H, 2H, 3] / [mini_batch, H, 2H] uint8 (created on first use).  A slot
        is handed out again only after the stager has issued the H2D copy that reads it AND that copy has finished (its event): the sampler runs
        at most four batches ahead of the copies."""
        ring = getattr(self, "_staging", None)
        shape = (self.mini_batch_size, self.height, 2 * self.height)
        if ring is None or ring["shape"] != shape:              # first use, or the instance's batch geometry changed since
            ring = self._staging = {"next": 0, "shape": shape, "slots": [
                [self._pinned(shape + (3,)), self._pinned(shape), None, False] for _ in range(4)]}
        slot = ring["slots"][ring["next"] % len(ring["slots"])]
        ring["next"] += 1
        while slot[3] and slot[2] is None:        # handed out earlier and still waiting in the batch deque for the stager
            if self._aborted():                   # ... which has died: do not wait for it
                raise self._Aborted()
            time.sleep(self.THREAD_SLEEP_INTERVAL)
        if slot[2] is not None:
            slot[2].synchronize()
            slot[2] = None
        slot[3] = True
        return slot

    @staticmethod
    def _pinned(shape):
        t = torch.empty(shape, dtype=torch.uint8)
        return t.pin_memory() if torch.cuda.is_available() else t

    def _release_staging(self):
        """After a phase (clean or failed) no slot is owed to a stager any more."""
        ring = getattr(self, "_staging", None)
        if ring is not None:
            for slot in ring["slots"]:
                if slot[2] is not None:
                    slot[2].synchronize()
                slot[2], slot[3] = None, False

    def _fill_queue(self, batch_deque, number_of_batches, signal_deque):
        """Stager thread (the FIFO queue of the reference graph, capacity 200): H2D on a side stream."""
        copy_stream = self._make_copy_stream()
        max_staged = None
        for _ in range(number_of_batches):
            batch = None
            while batch is None:
                try:
                    batch = batch_deque.popleft()
                except IndexError:
                    if self._aborted():
                        return
                    time.sleep(self.THREAD_SLEEP_INTERVAL)
            staged = self._stage_batch(batch, copy_stream)
            if max_staged is None:
                # the reference's FIFO queue holds 200 batches whatever their size (3 GB of device memory at batch 10 of 512x1024): here the
                # staged-ahead set is bounded by BYTES — 1 GiB, at least two batches, at most the reference's 200 entries
                nbytes = sum(int(t.numel()) * t.element_size() for t in staged[:2])
                max_staged = max(2, min(200, (1 << 30) // max(nbytes, 1)))
            while len(signal_deque) >= max_staged:
                if self._aborted():
                    return
                time.sleep(self.THREAD_SLEEP_INTERVAL)
            signal_deque.append(staged)

    def _make_copy_stream(self):
        return torch.cuda.Stream(device=self.engine.device)

    def _stage_batch(self, batch, copy_stream):
        """Host batch -> (frames on the device, labels on the device, event of the copies) on the copy stream."""
        dev = self.engine.device
        slot = batch.get('slot')
        if slot is not None:              # already in pinned memory (_fill_batch's fast path)
            with torch.cuda.stream(copy_stream):
                f_dev = slot[0].to(dev, non_blocking=True)
                l_dev = slot[1].to(dev, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            slot[2] = ready
        else:
            fr = batch['frames']
            fr = fr if fr.dtype == np.uint8 else fr.astype(np.float32)
            lb = batch['labels']
            if lb.dtype != np.uint8:
                li = lb.astype(np.float32).astype(np.int64)
                lb = np.where((li >= 0) & (li < 255), li, 255).astype(np.uint8)
            with torch.cuda.stream(copy_stream):
                f_dev = torch.from_numpy(np.ascontiguousarray(fr)).pin_memory().to(dev, non_blocking=True)
                l_dev = torch.from_numpy(np.ascontiguousarray(lb)).pin_memory().to(dev, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
        if batch.get('teacher_logits') is not None:       # soft_teacher: the batch's cached teacher logits travel with it
            with torch.cuda.stream(copy_stream):
                t_dev = torch.from_numpy(batch['teacher_logits']).pin_memory().to(dev, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            if slot is not None:
                slot[2] = ready
            return f_dev, l_dev, ready, t_dev
        return f_dev, l_dev, ready

    # ------------------------------------------------------------------ freeze / export
    def get_frozen_graph(self):
        return FrozenGraph(self._model_vars(), self.class_indices_graph, self.height, self.TOTAL_CLASSES)

    def save_to_frozen_graph(self, save_dir):
        graph_def = self.get_frozen_graph()
        with open(save_dir + ".pb", 'wb') as pb_file:
            pb_file.write(graph_def.SerializeToString())

    def close_model(self):
        self.engine.close()

    # ------------------------------------------------------------------ visualisation helpers (NumPy only)
    # Same signatures and return conventions as the reference (SemanticNetwork.py:719-755); bodies are this build's own:
    # palettes are looked up through one helper, the 50/50 overlay is integer arithmetic (cv2.addWeighted rounds half
    # to even on the float sum; so does _overlay), and the disagreement picture is built from boolean planes.
    def _check_hw(self, arr, channels=None, what="array"):
        want = (self.height, 2 * self.height) + (() if channels is None else (channels,))
        assert arr.shape == want, "%s must be %s, got %s" % (what, want, arr.shape)

    @staticmethod
    def _overlay(frame, colours):
        total = frame.astype(np.uint16) + colours.astype(np.uint16)           # 0..510
        half = total >> 1
        half += (total & 1) & (half & 1)                                      # x.5 -> nearest even, like rint
        return half.astype(np.uint8)

    def _paint(self, palette, label, frame):
        colours = np.asarray(palette)[np.asarray(label)]
        return colours if frame is None else (colours, self._overlay(frame, colours))

    def colorize(self, frame=None, label=None):
        assert frame is not None or label is not None, "At least a label or frame must be given"
        if frame is not None:
            self._check_hw(frame, 3, "frame")
        if label is None:
            label = self.predict_input(frame[None])[0]
        self._check_hw(label, None, "label")
        return self._paint(self.color_map_reduced_, label, frame)

    def colorize_teacher(self, label, frame=None):
        if frame is not None:
            self._check_hw(frame, 3, "frame")
        self._check_hw(label, None, "label")
        return self._paint(colormap(), label, frame)

    def cross_ignore(self, label_teacher, label_student=None, frame_student=None):
        """(cross_mask, ignore_mask): where teacher and student disagree (teacher colour, black elsewhere) and which
        pixels the metric skips (white).  As in the reference, subset index 0 doubles as 'ignored'."""
        assert label_student is not None or frame_student is not None, \
            "At least a label or frame from student must be given"
        self._check_hw(label_teacher, None, "label_teacher")
        if label_student is None:
            label_student = self.predict_input(frame_student[None])[0]
        self._check_hw(label_student, None, "label_student")
        teacher_k = self.take_array[label_teacher]
        skipped = teacher_k == 0
        ignore_mask = np.zeros(teacher_k.shape + (3,), dtype=np.uint8)
        ignore_mask[skipped] = self.WHITE
        differs = ~skipped & (teacher_k != label_student)
        cross_mask = np.zeros_like(ignore_mask)
        cross_mask[differs] = self.color_map_reduced_[teacher_k[differs]]
        return cross_mask, ignore_mask
