"""Python handle over the C++/HIP student engine (ams_amd/csrc/engine_*.hip, api.hip).

PyTorch-ROCm is plumbing here: it owns the device arena (one ``torch.uint8`` tensor the engine carves up),
the H2D/D2H copies and the stream; all arithmetic happens in libams_hip.so.  There is no CPU path: creating
an engine without a GPU or without the built library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import hip
from .spec import (BN_DECAY, BN_EPS_FROZEN, PIXEL_SCALE, Layer, StudentSpec, build_spec)
from . import weights as W

_ACT = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "relu6": hip.ACT_RELU6}

# Tuning knobs of tools/*.sh and the profiling recipes: environment variable -> ams_student_set_option id.  They are read HERE, once per
# engine, and applied through the one documented option entry point; the library itself reads no per-student environment.
_ENV_OPTIONS = {
    "AMS_DUAL_STREAM": hip.OPT_DUAL_STREAM, "AMS_DUAL_PARTS": hip.OPT_DUAL_PARTS, "AMS_DUAL_AUTOTUNE": hip.OPT_DUAL_AUTOTUNE,
    "AMS_BLOCK_X6": hip.OPT_BLOCK_X6, "AMS_LATE_SUB": hip.OPT_LATE_SUBBATCH, "AMS_STREAM_MIN_ROWS": hip.OPT_STREAM_MIN_ROWS,
    "AMS_OVERLAP_HEAD": hip.OPT_OVERLAP_HEAD, "AMS_FUSE_BLOCK": hip.OPT_FUSE_BLOCK, "AMS_FUSE_XDS": hip.OPT_FUSE_EXPAND_DW_STREAM,
    "AMS_OVERLAP_WGRAD": hip.OPT_OVERLAP_WGRAD, "AMS_FUSE_DGRAD_BN": hip.OPT_FUSE_DGRAD_BN, "AMS_FUSE_GEMM_RED": hip.OPT_FUSE_GEMM_RED,
    "AMS_TRAIN_RECOMPUTE": hip.OPT_TRAIN_RECOMPUTE, "AMS_NAN_GRADS": hip.OPT_NAN_GRADS, "AMS_FUSE_OPERAND_BN": hip.OPT_FUSE_OPERAND_BN,
    "AMS_WGRAD_FORK_EVERY": hip.OPT_WGRAD_FORK_EVERY, "AMS_TRAIN_FWD_F16": hip.OPT_TRAIN_FWD_F16,
}


def _role(layer: Layer) -> int:
    s = layer.scope
    if s == "MobilenetV2/Conv":
        return hip.ROLE_STEM
    if s.endswith("/expand"):
        return hip.ROLE_EXPAND
    if s.endswith("/depthwise"):
        return hip.ROLE_DEPTHWISE
    if s.endswith("/project"):
        return hip.ROLE_PROJECT
    return {"image_pooling": hip.ROLE_POOL_CONV, "aspp0": hip.ROLE_ASPP, "concat_projection": hip.ROLE_CONCAT_PROJ,
            "logits/semantic": hip.ROLE_LOGITS}[s]


def layer_table(spec: StudentSpec):
    """spec.py layer table -> C array of ams_layer_desc."""
    arr = (hip.LayerDesc * len(spec.layers))()
    for i, l in enumerate(spec.layers):
        d = arr[i]
        d.role = _role(l)
        d.cin, d.cout, d.stride, d.rate = l.cin, l.cout, l.stride, l.rate
        d.act = _ACT[l.act]
        d.residual_from = l.residual_from or 0
        d.w_off = spec.by_name[l.weight_name].offset
        if l.bn_eps is not None:
            d.bn_eps = l.bn_eps
            d.gamma_off = spec.by_name[l.scope + "/BatchNorm/gamma:0"].offset
            d.beta_off = spec.by_name[l.scope + "/BatchNorm/beta:0"].offset
            d.mean_off = spec.by_name[l.scope + "/BatchNorm/moving_mean:0"].offset
            d.var_off = spec.by_name[l.scope + "/BatchNorm/moving_variance:0"].offset
        else:
            d.bn_eps = -1.0
            d.gamma_off = spec.by_name[l.scope + "/biases:0"].offset
            d.beta_off = d.mean_off = d.var_off = 0
    return arr


class StudentEngine:
    """One student network resident on one GPU."""

    def __init__(self, class_indices: Sequence[int], height: int, width: Optional[int] = None, max_batch: int = 1,
                 trainable: bool = False, num_classes: int = 19, device: str | torch.device = "cuda:0"):
        if not torch.cuda.is_available():
            raise hip.AmsHipError("no GPU visible: the AMS student runs only on MI355X (no CPU fallback)")
        self.lib = hip.lib()
        self.device = torch.device(device)
        self.spec = build_spec(num_classes)
        self.height, self.width = int(height), int(width if width is not None else 2 * height)
        self.max_batch = int(max_batch)
        self.trainable = bool(trainable)
        self.num_classes = int(num_classes)
        self.soft_teacher = False
        self._reg_mask = self._teacher_logits_dev = None
        self._frames_b, self._frames_u8, self._frames_metric = 0, False, False
        self.class_indices = [int(c) for c in class_indices]
        self.K = len(self.class_indices)
        cfg = hip.StudentConfig()
        cfg.abi_version = hip.ABI_VERSION
        cfg.height, cfg.width, cfg.max_batch = self.height, self.width, self.max_batch
        cfg.num_classes, cfg.n_selected = num_classes, self.K
        for i, c in enumerate(self.class_indices):
            cfg.class_indices[i] = c
        cfg.n_layers = len(self.spec.layers)
        cfg.trainable = 1 if trainable else 0
        cfg.act_dtype = hip.DT_F32
        cfg.n_trainable, cfg.n_stats = self.spec.n_trainable, self.spec.n_stats
        cfg.bn_decay, cfg.bn_eps_frozen, cfg.pixel_scale = BN_DECAY, BN_EPS_FROZEN, PIXEL_SCALE
        self._cfg = cfg
        self._layers = layer_table(self.spec)
        need = C.c_size_t(0)
        hip.check(self.lib.ams_student_arena_bytes(C.byref(cfg), self._layers, C.byref(need)), "ams_student_arena_bytes")
        self.arena_bytes = int(need.value)
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(self.arena_bytes, dtype=torch.uint8, device=self.device)
            handle = C.c_void_p()
            hip.check(self.lib.ams_student_create(C.byref(cfg), self._layers, C.c_void_p(self.arena.data_ptr()),
                                                  self.arena_bytes, C.byref(handle)), "ams_student_create")
        self._h = handle
        self.params = self._view(hip.REGION_PARAMS, torch.float32)
        self.stats = self._view(hip.REGION_STATS, torch.float32)
        self.frozen_params = self._view(hip.REGION_FROZEN, torch.float32)
        self.logits_lowres = self._view(hip.REGION_LOGITS, torch.float32)
        self.bn_sync = self._view(hip.REGION_BN_SYNC, torch.float64)
        if trainable:
            self.grads = self._view(hip.REGION_GRADS, torch.float32)
            self.adam_m = self._view(hip.REGION_ADAM_M, torch.float32)
            self.adam_v = self._view(hip.REGION_ADAM_V, torch.float32)
        h, w = C.c_int32(), C.c_int32()
        hip.check(self.lib.ams_student_lowres_size(self._h, C.byref(h), C.byref(w)))
        self.lowres = (h.value, w.value)
        # one output block for the host-returning calls: [conf int64 K*K | loss f64[2] | labels int32 B*H*W], mirrored in pinned
        # host memory, so a call costs ONE device -> host copy and no allocation (SemanticNetwork.predict_with_metric)
        self._out_meta = (self.K * self.K + 2) * 8
        nbytes = self._out_meta * self.max_batch + self.max_batch * self.height * self.width * 4      # room for per-frame metrics too
        self._out_dev = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self._out_host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        # ... and one pinned input block [frames uint8 B*H*W*3 | labels uint8 B*H*W]: host uint8 arrays are gathered there and leave with an
        # asynchronous DMA (a pageable source makes the driver stage the copy itself, synchronously, chunk by chunk); `_in_free` is the event of
        # the last DMA that read the block
        self._in_frames = torch.zeros(self.max_batch * self.height * self.width * 3, dtype=torch.uint8).pin_memory()
        self._in_labels = torch.zeros(self.max_batch * self.height * self.width, dtype=torch.uint8).pin_memory()
        self._in_free = {}
        self._keepalive = []
        for name, opt in _ENV_OPTIONS.items():
            if name in os.environ:
                hip.check(self.lib.ams_student_set_option(self._h, opt, int(os.environ[name])), "ams_student_set_option(%s)" % name)

    # ------------------------------------------------------------------ plumbing
    def _view(self, region: int, dtype: torch.dtype) -> torch.Tensor:
        off, n = C.c_size_t(), C.c_size_t()
        hip.check(self.lib.ams_student_region(self._h, region, C.byref(off), C.byref(n)), "ams_student_region")
        item = torch.empty((), dtype=dtype).element_size()
        return self.arena[off.value:off.value + n.value * item].view(dtype)

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self) -> None:
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self.lib.ams_student_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _frames_to_device(self, frames) -> tuple:
        """[B,H,W,3] uint8 or float array/tensor -> (device tensor, dtype code, B)."""
        if isinstance(frames, torch.Tensor):
            t = frames
        else:
            a = np.asarray(frames)
            if a.dtype == np.uint8 and a.ndim == 4 and tuple(a.shape[1:]) == (self.height, self.width, 3) and 0 < a.shape[0] <= self.max_batch:
                return self._staged(a, self._in_frames), hip.DT_U8, int(a.shape[0])
            a = np.ascontiguousarray(a)
            if a.dtype != np.uint8:
                a = a.astype(np.float32, copy=False)
            t = torch.from_numpy(a)
        if t.dtype not in (torch.uint8, torch.float32):
            t = t.to(torch.float32)
        assert t.dim() == 4 and tuple(t.shape[1:]) == (self.height, self.width, 3), \
            "frames must be [B,%d,%d,3], got %s" % (self.height, self.width, tuple(t.shape))
        assert 0 < t.shape[0] <= self.max_batch, "batch %d outside 1..%d" % (t.shape[0], self.max_batch)
        t = t.to(self.device, non_blocking=True).contiguous()
        return t, (hip.DT_U8 if t.dtype == torch.uint8 else hip.DT_F32), int(t.shape[0])

    def _staged(self, a: np.ndarray, block: torch.Tensor) -> torch.Tensor:
        """host uint8 array -> device tensor through the engine's pinned input block (one host copy, one asynchronous DMA)."""
        prev = self._in_free.get(id(block))
        if prev is not None:
            prev.synchronize()                          # the previous DMA out of THIS block (long done in a synchronous call loop)
        view = block[:a.size].view(a.shape)
        np.copyto(view.numpy(), a)
        t = view.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._in_free[id(block)] = ev
        return t

    def _labels_to_device(self, labels, batch: int) -> torch.Tensor:
        if isinstance(labels, torch.Tensor):
            t = labels
        else:
            a = np.asarray(labels)
            if a.dtype == np.uint8 and tuple(a.shape) == (batch, self.height, self.width) and a.size <= self._in_labels.numel():
                return self._staged(a, self._in_labels)
            if a.dtype != np.uint8:
                # tf.cast(labels, int32) truncates toward zero; ids outside 0..255 can never be selected -> 255
                ai = a.astype(np.float32).astype(np.int64)
                a = np.where((ai >= 0) & (ai < 255), ai, 255).astype(np.uint8)
            t = torch.from_numpy(np.ascontiguousarray(a))
        if t.dtype != torch.uint8:
            ti = t.to(torch.int64)
            t = torch.where((ti >= 0) & (ti < 255), ti, torch.full_like(ti, 255)).to(torch.uint8)
        assert tuple(t.shape) == (batch, self.height, self.width), "labels must be [B,H,W], got %s" % (tuple(t.shape),)
        return t.to(self.device, non_blocking=True).contiguous()

    # ------------------------------------------------------------------ variables
    def load_variables(self, variables: Dict[str, np.ndarray]) -> None:
        """Assign every model variable present in ``variables`` (SaveHelper.restore_vars); Adam state untouched."""
        tr = self.params.cpu().numpy()
        st = self.stats.cpu().numpy()
        for v in self.spec.trainable:
            if v.name in variables:
                tr[v.offset:v.offset + v.size] = np.asarray(variables[v.name], dtype=np.float32).reshape(-1)
        for v in self.spec.stats:
            if v.name in variables:
                st[v.offset:v.offset + v.size] = np.asarray(variables[v.name], dtype=np.float32).reshape(-1)
        self.params.copy_(torch.from_numpy(tr))
        self.stats.copy_(torch.from_numpy(st))

    def get_variables(self) -> Dict[str, np.ndarray]:
        return W.unpack(self.spec, self.params.cpu().numpy(), self.stats.cpu().numpy())

    def set_matmul_mode(self, mode: int) -> None:
        """hip.MATMUL_SPLIT_BF16_X6 (default: late-layer products via 6 bf16 MFMAs on three-part splits, f32-level),
        hip.MATMUL_SPLIT_BF16 (3 MFMAs on two-part splits in frozen inference: +5 % frames/s, logits 2e-4..5e-4),
        hip.MATMUL_F32 (exact f32 MFMA everywhere) or hip.MATMUL_BF16 (the opt-in bf16 variant: one part, plain bf16 products in the
        late layers of frozen inference; outside the f32 tolerance — bench.py reports its mismatch beside its speed)."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_MATMUL, int(mode)), "ams_student_set_option")

    def pack_masked_fp16(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Masked trainable parameters as fp16, compacted in trainable order, on the device (the value part of the downlink
        delta, reference run.py:330-332).  mask: uint8 [n_trainable] on the device, None = all.  -> int16 view of the halves."""
        n = self.spec.n_trainable
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.numel() == n and mask.device == self.arena.device
        out = torch.empty(n, dtype=torch.int16, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        scratch = torch.empty(int(self.lib.ams_pack_masked_fp16_scratch(n)), dtype=torch.int64, device=self.device)
        hip.check(self.lib.ams_pack_masked_fp16(C.c_void_p(self.params.data_ptr()), C.c_void_p(mask.data_ptr()) if mask is not None else None,
                                                n, C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                                C.c_void_p(scratch.data_ptr()), scratch.numel(), self._stream()),
                  "ams_pack_masked_fp16")
        return out[:int(cnt.item())]

    def set_fuse_operand_bn(self, on: bool) -> None:
        """Fine-tune step: BN + activation of the depthwise layers applied by the project GEMM / project weight gradient on their operand loads
        (default on; the depthwise activation is never written); bit-identical to off = the pass written."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_OPERAND_BN, int(bool(on))), "ams_student_set_option")

    def set_wgrad_fork_every(self, n: int) -> None:
        """Fine-tune step: weight gradients handed to the side stream ``n`` at a time (default 1: each as soon as its operands exist; larger
        n = fewer events on the main stream, measured slower on MI355X).  Bit-identical for every n."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_WGRAD_FORK_EVERY, int(n)), "ams_student_set_option")

    def set_train_recompute(self, on: bool, fuse_dgrad_bn: Optional[bool] = None, fuse_gemm_red: Optional[int] = None) -> None:
        """Fine-tune step of the early blocks without their 6x-expanded tensors (default on); off = every tensor materialised.
        ``fuse_dgrad_bn``: the one-kernel depthwise backward of the stride-16 blocks (default on).
        ``fuse_gemm_red``: BN column reductions in the 1x1 GEMM epilogues, bit 0 forward statistics, bit 1 backward sums (default 3)."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_TRAIN_RECOMPUTE, 2 if on is True else int(on)), "ams_student_set_option")
        if fuse_gemm_red is not None:
            hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_GEMM_RED, int(fuse_gemm_red)), "ams_student_set_option")
        if fuse_dgrad_bn is not None:
            hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_DGRAD_BN, 2 if fuse_dgrad_bn is True else int(fuse_dgrad_bn)), "ams_student_set_option")

    def set_soft_teacher(self, on: bool) -> None:
        """create_student_v3(soft_teacher=True) (utils/graph_utils.py:403-404): the fine-tune loss targets softmax(gather(teacher logits)); every
        ``train_step`` then needs ``teacher_logits``."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_SOFT_TEACHER, int(bool(on))), "ams_student_set_option")
        self.soft_teacher = bool(on)

    def set_regularizer(self, on: bool, biases_only: bool = False, coef: float = 0.01) -> None:
        """create_student_v3(regularize=True[, train_biases_only=True]) (utils/graph_utils.py:451-456): loss += coef * mean over tvars of
        l2_loss(v); tvars = every trainable variable, or those without 'weight' in their name."""
        if not on:
            hip.check(self.lib.ams_student_set_regularizer(self._h, None, 0, 0.0), "ams_student_set_regularizer")
            self._reg_mask = None
            return
        flat = np.zeros(self.spec.n_trainable, dtype=np.uint8)
        n_vars = 0
        for v in self.spec.trainable:
            if biases_only and 'weight' in v.name:
                continue
            flat[v.offset:v.offset + v.size] = 1
            n_vars += 1
        self._reg_mask = torch.from_numpy(flat).to(self.device)          # referenced by the handle: kept alive here
        hip.check(self.lib.ams_student_set_regularizer(self._h, C.c_void_p(self._reg_mask.data_ptr()), n_vars, float(coef)),
                  "ams_student_set_regularizer")

    def _feed_teacher_logits(self, teacher_logits, b: int):
        """feed_dict[student['teacher_labels_logits_pl']]: f32 [b, th, tw, num_classes] (host array or device tensor); returns the device tensor
        (the caller keeps it alive until the step has been enqueued on the same stream order)."""
        if teacher_logits is None:
            assert not getattr(self, "soft_teacher", False), "soft_teacher is on: teacher_logits must be fed (teacher_labels_logits_pl)"
            return None
        t = teacher_logits if isinstance(teacher_logits, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(teacher_logits, dtype=np.float32))
        assert t.dim() == 4 and t.shape[0] == b and t.shape[3] == self.num_classes, "teacher logits must be [batch, th, tw, %d]" % self.num_classes
        assert t.shape[1] <= self.height and t.shape[2] <= self.width
        t = t.to(self.device, dtype=torch.float32, non_blocking=True).contiguous()
        hip.check(self.lib.ams_student_feed_teacher_logits(self._h, C.c_void_p(t.data_ptr()), int(t.shape[1]), int(t.shape[2])),
                  "ams_student_feed_teacher_logits")
        return t

    def set_nan_grads(self, on: bool) -> None:
        """A fine-tune batch without one valid pixel: False (default) NaN loss and ZERO gradients, TensorFlow's result for
        utils/graph_utils.py:408 (reduce_mean over the empty boolean_mask: the weights survive); True NaN gradients instead."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_NAN_GRADS, int(bool(on))), "ams_student_set_option")

    def set_fuse_first_block(self, on: int) -> None:
        """Frozen inference, stem + depthwise + project of the first block: 0 three kernels, 1 one kernel with a tile per block
        (k_first_block.hip; default), 2 one kernel with a tile per wave (k_block.hip; measured slower).  Forms 0 and 2 keep an exact-f32 stem;
        form 1 runs the stem as six bf16 products on three-part splits while set_block_x6 is on (default; f32-level), else the same bits."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_FIRST_BLOCK, int(on)), "ams_student_set_option")

    def set_fuse_dw_project(self, on: bool) -> None:
        """Frozen inference: depthwise + project of the stride-16 blocks as one kernel (default off: measured no faster than
        the two kernels; split-bf16 mode only)."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_DW_PROJECT, int(bool(on))), "ams_student_set_option")

    def set_block_x6(self, on: bool) -> None:
        """Whole-block kernels: expand products of the K = 24 / 32 blocks as six bf16 MFMAs on three-part splits (default on, f32-level)
        or as exact f32 MFMAs (bit-identical to the layer-by-layer plan).  Also selects the stem's products in the one-kernel first block."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_BLOCK_X6, int(bool(on))), "ams_student_set_option")

    def set_dual_stream(self, mode: int, parts: Optional[int] = None, autotune: Optional[bool] = None) -> None:
        """Frozen inference as two to four part-batches on as many streams: 0 never, 1 (default) a fixed function of the batch size
        (never synchronises), n >= 2 always ``parts`` parts from n frames on.  ``autotune=True`` (with mode 1): time the plans inside the
        first call per batch size instead of the static rule (that call synchronises)."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_DUAL_STREAM, int(mode)), "ams_student_set_option")
        if parts is not None:
            hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_DUAL_PARTS, int(parts)), "ams_student_set_option")
        if autotune is not None:
            hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_DUAL_AUTOTUNE, int(bool(autotune))), "ams_student_set_option")

    def set_late_subbatch(self, frames: int) -> None:
        """Frozen inference: frames per pass of the output-stride-16 section (0 = the whole batch); same bits either way."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_LATE_SUBBATCH, int(frames)), "ams_student_set_option")

    def set_fuse_block(self, on: bool) -> None:
        """Frozen inference: every early block with Cin <= 32 as ONE kernel (expand + depthwise + project [+ input]; default on)."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_BLOCK, int(bool(on))), "ams_student_set_option")

    def set_fuse_expand_dw(self, on: int) -> None:
        """Frozen inference, expand + depthwise of a block as one kernel: 0 never, 1 (default) the blocks where it is
        measured faster, 2 every supported block."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_EXPAND_DW, int(on)), "ams_student_set_option")

    def set_fuse_expand_dw_stream(self, on: int) -> None:
        """Frozen inference, split-bf16 modes: expand + depthwise of the stride-16 blocks as one streaming kernel, bit-identical
        to the two kernels it replaces: 0 never, 1 (default) where measured faster, 2 every supported block."""
        hip.check(self.lib.ams_student_set_option(self._h, hip.OPT_FUSE_EXPAND_DW_STREAM, int(on)), "ams_student_set_option")

    def freeze(self) -> None:
        """Device-side server->edge hand-off (replaces save_to_frozen_graph + reload)."""
        hip.check(self.lib.ams_student_freeze(self._h, self._stream()), "ams_student_freeze")

    # ------------------------------------------------------------------ compute
    def predict(self, frames, mode: int = hip.MODE_FROZEN) -> torch.Tensor:
        t, dt, b = self._frames_to_device(frames)
        out = torch.empty((b, self.height, self.width), dtype=torch.int32, device=self.device)
        hip.check(self.lib.ams_student_predict(self._h, C.c_void_p(t.data_ptr()), dt, b, mode, C.c_void_p(out.data_ptr()),
                                               self._stream()), "ams_student_predict")
        return out

    def graphed_predict(self, batch: int, mode: int = hip.MODE_FROZEN) -> "GraphedPredict":
        """Capture one inference step for ``batch`` frames into a hipGraph (see GraphedPredict)."""
        return GraphedPredict(self, batch, mode)

    def predict_with_metric(self, frames, labels_teacher, mode: int = hip.MODE_FROZEN):
        """-> (labels int32 [B,H,W] (device), conf_mat int64 [K,K] (device), loss_sum_count f64[2] (device))."""
        t, dt, b = self._frames_to_device(frames)
        lab = self._labels_to_device(labels_teacher, b)
        out = torch.empty((b, self.height, self.width), dtype=torch.int32, device=self.device)
        conf = torch.empty(self.K * self.K, dtype=torch.int64, device=self.device)
        loss = torch.empty(2, dtype=torch.float64, device=self.device)
        hip.check(self.lib.ams_student_predict_with_metric(
            self._h, C.c_void_p(t.data_ptr()), dt, b, mode, C.c_void_p(lab.data_ptr()), C.c_void_p(out.data_ptr()),
            C.c_void_p(conf.data_ptr()), C.c_void_p(loss.data_ptr()), self._stream()), "ams_student_predict_with_metric")
        return out, conf.view(self.K, self.K), loss

    # Host-returning calls (what SemanticNetwork.predict_input / predict_with_metric make): the label maps leave the device as uint8 (an index
    # inside the subset of K <= 32 classes) and are widened to the reference's int32 on the host: 0.5 MB instead of 2 MB per 512 x 1024 frame over
    # PCIe, and the widening writes the fresh array the caller gets anyway (copying 2 MB of int32 out of the pinned mirror took longer than the
    # transfer).  Metrics come back per frame (ams_student_predict_frames_u8) and are summed here: integers and exact fixed-point sums.
    def predict_host(self, frames, mode: int = hip.MODE_FROZEN) -> np.ndarray:
        """predict() with the label maps returned as a fresh int32 ndarray (one device -> host copy, preallocated buffers)."""
        self.predict_frames(frames, None, mode, u8=True)
        return self.fetch_frames(labels_only=True)[0]

    def predict_with_metric_host(self, frames, labels_teacher, mode: int = hip.MODE_FROZEN):
        """predict_with_metric() -> (labels int32 [B,H,W], conf_mat int64 [K,K], loss_sum_count f64[2]) as fresh ndarrays."""
        self.predict_frames(frames, labels_teacher, mode, u8=True)
        labels, confs, losses = self.fetch_frames()
        return labels, confs.sum(axis=0), losses.sum(axis=0)

    def predict_frames(self, frames, labels_teacher=None, mode: int = hip.MODE_FROZEN, u8: bool = False):
        """Per-FRAME results of one batched pass (ams_student_predict_frames): device tensors labels int32 [B,H,W] (``u8``: uint8, for results
        that go to the host), conf int64 [B,K,K], loss f64 [B,2] (views of the engine's output block, valid until the next host-returning /
        predict_frames call).  Nothing is synchronised; ``fetch_frames`` brings them to the host."""
        t, dt, b = self._frames_to_device(frames)
        lab = self._labels_to_device(labels_teacher, b) if labels_teacher is not None else None
        kk = self.K * self.K
        base = self._out_dev.data_ptr()
        meta = self._out_meta * b
        fn = self.lib.ams_student_predict_frames_u8 if u8 else self.lib.ams_student_predict_frames
        hip.check(fn(self._h, C.c_void_p(t.data_ptr()), dt, b, mode, C.c_void_p(lab.data_ptr()) if lab is not None else None, C.c_void_p(base + meta),
                     C.c_void_p(base), C.c_void_p(base + kk * 8 * b), self._stream()), "ams_student_predict_frames")
        self._keepalive = [t, lab]
        self._frames_b = b
        self._frames_u8 = bool(u8)
        self._frames_metric = lab is not None
        conf = self._out_dev[:kk * 8 * b].view(torch.int64).view(b, self.K, self.K)
        loss = self._out_dev[kk * 8 * b:meta].view(torch.float64).view(b, 2)
        px = b * self.height * self.width
        labels = (self._out_dev[meta:meta + px].view(b, self.height, self.width) if u8 else
                  self._out_dev[meta:meta + px * 4].view(torch.int32).view(b, self.height, self.width))
        return labels, conf, loss

    def fetch_frames(self, labels_only: bool = False):
        """(labels [B,H,W] int32, conf [B,K,K] int64, loss [B,2] f64) of the last ``predict_frames`` as fresh ndarrays: one copy, one sync.
        (``labels_only``, or a pass without teacher labels: conf and loss are None and only the label bytes cross PCIe.)"""
        b = self._frames_b
        kk = self.K * self.K
        meta = self._out_meta * b
        px = b * self.height * self.width
        n = meta + (px if self._frames_u8 else px * 4)
        first = meta if (labels_only or not self._frames_metric) else 0
        self._out_host[first:n].copy_(self._out_dev[first:n], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        host = self._out_host.numpy()
        conf = loss = None
        if first == 0:
            conf = host[:kk * 8 * b].view(np.int64).reshape(b, self.K, self.K).copy()
            loss = host[kk * 8 * b:meta].view(np.float64).reshape(b, 2).copy()
        if self._frames_u8:
            labels = host[meta:n].reshape(b, self.height, self.width).astype(np.int32)       # widened into the fresh array the caller gets
        else:
            labels = host[meta:n].view(np.int32).reshape(b, self.height, self.width).copy()
        return labels, conf, loss

    def cross_confusion(self, labels_pair) -> torch.Tensor:
        a = np.asarray(labels_pair)
        assert a.shape[0] == 2
        lab = self._labels_to_device(a.reshape(2, *a.shape[1:]), 2) if a.shape[1:] == (self.height, self.width) else None
        assert lab is not None, "labels must be [2,H,W]"
        conf = torch.empty(self.K * self.K, dtype=torch.int64, device=self.device)
        hip.check(self.lib.ams_cross_confusion(self._h, C.c_void_p(lab.data_ptr()), self.height * self.width,
                                               C.c_void_p(conf.data_ptr()), self._stream()), "ams_cross_confusion")
        return conf.view(self.K, self.K)

    def train_step(self, frames, labels_teacher, lr: float, mask: Optional[torch.Tensor] = None,
                   allreduce=None, global_batch: Optional[int] = None, comm=None, teacher_logits=None) -> torch.Tensor:
        """One Adam iteration; returns the device tensor f64[2] = (CE sum over valid pixels, valid pixel count).

        Data-parallel step, ``global_batch`` = frames over all ranks: ``comm`` = an ``ams_amd.dist.RcclComm``
        (production: the engine issues its all-reduces on the launch stream through RCCL), or ``allreduce`` = an
        ``ams_amd.dist.ArenaAllReduce`` / a callable(tensor) summing in place across ranks (host callback: gloo tests)."""
        assert self.trainable, "Can't train frozen graph!!!"
        t, dt, b = self._frames_to_device(frames)
        lab = self._labels_to_device(labels_teacher, b)
        self._teacher_logits_dev = self._feed_teacher_logits(teacher_logits, b)          # soft_teacher only; alive until the next step replaces it
        loss = torch.empty(2, dtype=torch.float64, device=self.device)
        mptr = C.c_void_p(mask.data_ptr()) if mask is not None else C.c_void_p(0)
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.numel() == self.spec.n_trainable and mask.device == self.arena.device
        if comm is not None:
            assert allreduce is None, "pass either comm (RCCL inside the engine) or allreduce (host callback), not both"
            hip.check(self.lib.ams_student_train_step_rccl(self._h, C.c_void_p(t.data_ptr()), dt, C.c_void_p(lab.data_ptr()), b,
                                                           int(global_batch or b), float(lr), mptr, C.c_void_p(loss.data_ptr()),
                                                           comm._h, self._stream()), "ams_student_train_step_rccl")
            return loss
        if allreduce is None:
            hip.check(self.lib.ams_student_train_step(self._h, C.c_void_p(t.data_ptr()), dt, C.c_void_p(lab.data_ptr()), b,
                                                      float(lr), mptr, C.c_void_p(loss.data_ptr()), self._stream()),
                      "ams_student_train_step")
            return loss
        from .dist import ArenaAllReduce
        reducer = allreduce if isinstance(allreduce, ArenaAllReduce) else ArenaAllReduce(self.arena, reduce_fn=allreduce)
        cb = hip.ALLREDUCE_CB(reducer)
        rc = self.lib.ams_student_train_step_dp(self._h, C.c_void_p(t.data_ptr()), dt, C.c_void_p(lab.data_ptr()), b,
                                                int(global_batch or b), float(lr), mptr, C.c_void_p(loss.data_ptr()),
                                                cb, None, self._stream())
        if reducer.error is not None:
            raise reducer.error
        hip.check(rc, "ams_student_train_step_dp")
        return loss

    @property
    def adam_step(self) -> int:
        t = C.c_int64()
        hip.check(self.lib.ams_student_get_adam_step(self._h, C.byref(t)))
        return int(t.value)

    @adam_step.setter
    def adam_step(self, value: int) -> None:
        hip.check(self.lib.ams_student_set_adam_step(self._h, int(value)))


class GraphedPredict:
    """The ~60 kernel launches of one inference step captured once into a hipGraph and replayed.

    At batch 1 the step is launch-bound (60 launches x ~3.5 us of host time against ~0.4 ms of kernels); a replay costs
    one launch.  Frames are copied into a static uint8 buffer, labels come back in a static int32 buffer (valid until the
    next call).  Capture goes through torch.cuda.graph: the engine launches on torch's current stream, which is the
    capturing stream inside that context."""

    def __init__(self, engine: StudentEngine, batch: int, mode: int = hip.MODE_FROZEN):
        assert 0 < batch <= engine.max_batch
        self.engine, self.batch, self.mode = engine, batch, mode
        dev = engine.device
        self.frames = torch.zeros((batch, engine.height, engine.width, 3), dtype=torch.uint8, device=dev)
        self.labels = torch.empty((batch, engine.height, engine.width), dtype=torch.int32, device=dev)
        self._run()                                   # warm-up outside capture (lazy one-time initialisations)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._run()

    def _run(self) -> None:
        e = self.engine
        hip.check(e.lib.ams_student_predict(e._h, C.c_void_p(self.frames.data_ptr()), hip.DT_U8, self.batch, self.mode,
                                            C.c_void_p(self.labels.data_ptr()), e._stream()), "ams_student_predict")

    def __call__(self, frames) -> torch.Tensor:
        if not isinstance(frames, torch.Tensor):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        self.frames.copy_(frames, non_blocking=True)
        self.graph.replay()
        return self.labels
