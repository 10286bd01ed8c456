"""ctypes binding of libams_hip.so (C ABI in include/ams_hip.h).

The product has no CPU fallback: if the library is missing or cannot be loaded, importing this module's
``lib()`` raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C ams_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libams_hip.so"

ABI_VERSION = 3

# enums of include/ams_hip.h
ROLE_STEM, ROLE_EXPAND, ROLE_DEPTHWISE, ROLE_PROJECT, ROLE_POOL_CONV, ROLE_ASPP, ROLE_CONCAT_PROJ, ROLE_LOGITS = range(8)
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
DT_F32, DT_U8, DT_I32, DT_F64 = 0, 1, 3, 4          # 2 was bf16 activation storage: measured and dropped (include/ams_hip.h)
MODE_FROZEN, MODE_LIVE = 0, 1
OPT_MATMUL = 1
RESIZE_NEAREST, RESIZE_LINEAR = 0, 1
OPT_FUSE_EXPAND_DW = 2
OPT_FUSE_DW_PROJECT = 3
OPT_FUSE_FIRST_BLOCK = 4
OPT_FUSE_EXPAND_DW_STREAM = 5
OPT_FUSE_BLOCK = 6
OPT_LATE_SUBBATCH = 7
OPT_BLOCK_X6 = 8
OPT_DUAL_STREAM = 10
OPT_TRAIN_RECOMPUTE = 11
OPT_FUSE_DGRAD_BN = 16
OPT_FUSE_GEMM_RED = 17
OPT_EMULATE_BF16_STORAGE = 15
OPT_DUAL_AUTOTUNE = 12
OPT_DUAL_PARTS = 13
OPT_NAN_GRADS = 18
OPT_TRAIN_FWD_F16 = 24
OPT_SOFT_TEACHER = 25
OPT_OVERLAP_WGRAD = 19
OPT_OVERLAP_HEAD = 20
OPT_STREAM_MIN_ROWS = 21
OPT_FUSE_OPERAND_BN = 22
OPT_WGRAD_FORK_EVERY = 23
MATMUL_F32, MATMUL_SPLIT_BF16, MATMUL_SPLIT_BF16_X6, MATMUL_BF16, MATMUL_SPLIT_F16 = 0, 1, 2, 3, 4
MATMUL_DEFAULT = MATMUL_SPLIT_F16      # frozen inference; the fine-tune step forms its split products on three bf16 parts in every mode but MATMUL_F32
(REGION_PARAMS, REGION_STATS, REGION_GRADS, REGION_ADAM_M, REGION_ADAM_V, REGION_FROZEN, REGION_BN_SYNC,
 REGION_LOGITS) = range(8)


class LayerDesc(C.Structure):
    _fields_ = [("role", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("stride", C.c_int32),
                ("rate", C.c_int32), ("act", C.c_int32), ("residual_from", C.c_int32), ("bn_eps", C.c_float),
                ("w_off", C.c_int64), ("gamma_off", C.c_int64), ("beta_off", C.c_int64), ("mean_off", C.c_int64),
                ("var_off", C.c_int64)]


class StudentConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("max_batch", C.c_int32),
                ("num_classes", C.c_int32), ("n_selected", C.c_int32), ("class_indices", C.c_int32 * 32),
                ("n_layers", C.c_int32), ("trainable", C.c_int32), ("act_dtype", C.c_int32),
                ("n_trainable", C.c_int64), ("n_stats", C.c_int64), ("bn_decay", C.c_float),
                ("bn_eps_frozen", C.c_float), ("pixel_scale", C.c_float)]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int32)

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes).  Every symbol declared in include/ams_hip.h appears here; tests check both ways.
SIGNATURES = {
    "ams_last_error": (C.c_char_p, []),
    "ams_abi_version": (C.c_int, []),
    "ams_device_info": (C.c_int, [C.c_char_p, _sz, C.POINTER(_i32), C.POINTER(_i64)]),
    "ams_student_arena_bytes": (C.c_int, [C.POINTER(StudentConfig), C.POINTER(LayerDesc), C.POINTER(_sz)]),
    "ams_student_create": (C.c_int, [C.POINTER(StudentConfig), C.POINTER(LayerDesc), _vp, _sz, C.POINTER(_vp)]),
    "ams_student_destroy": (None, [_vp]),
    "ams_student_region": (C.c_int, [_vp, _i32, C.POINTER(_sz), C.POINTER(_sz)]),
    "ams_student_lowres_size": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "ams_student_layer_tensor": (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ams_student_freeze": (C.c_int, [_vp, _vp]),
    "ams_student_predict": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "ams_student_predict_with_metric": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ams_student_predict_frames": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ams_student_predict_frames_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ams_cross_confusion": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "ams_student_train_step": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _f32, _vp, _vp, _vp]),
    "ams_student_train_step_dp": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp, _vp, ALLREDUCE_CB, _vp, _vp]),
    "ams_comm_unique_id": (C.c_int, [_vp, _sz]),
    "ams_comm_create": (C.c_int, [_vp, _sz, _i32, _i32, C.POINTER(_vp)]),
    "ams_comm_destroy": (None, [_vp]),
    "ams_comm_stats": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "ams_comm_set_timing": (C.c_int, [_vp, _i32]),
    "ams_comm_timing_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "ams_comm_allreduce": (C.c_int, [_vp, _vp, _sz, _i32, _vp]),
    "ams_student_train_step_rccl": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "ams_student_set_option": (C.c_int, [_vp, _i32, _i32]),
    "ams_student_feed_teacher_logits": (C.c_int, [_vp, _vp, _i32, _i32]),
    "ams_student_f16_fallback_layers": (C.c_int, [_vp, C.POINTER(_i32)]),
    "ams_student_set_regularizer": (C.c_int, [_vp, _vp, _i32, _f32]),
    "ams_student_profile": (C.c_int, [_vp, _i32]),
    "ams_student_profile_read": (C.c_int, [_vp, C.c_char_p, _sz, C.POINTER(_sz)]),
    "ams_student_get_adam_step": (C.c_int, [_vp, C.POINTER(_i64)]),
    "ams_student_set_adam_step": (C.c_int, [_vp, _i64]),
    "ams_pack_masked_fp16": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "ams_pack_masked_fp16_scratch": (_sz, [_i64]),
    "ams_k_stem_conv": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _f32, _vp, _vp]),
    "ams_k_depthwise3x3": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ams_k_pointwise": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp]),
    "ams_k_pointwise_split": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ams_k_pointwise_split3": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ams_k_pack_h2i": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "ams_k_pointwise_split_f16": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "ams_k_expand_dw_stream_f16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _vp]),
    "ams_ingest_resize_u8": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "ams_k_dw_project": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ams_k_block_fused": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "ams_k_block_fused_f16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "ams_k_expand_dw": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ams_k_expand_dw_stream": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _i32, _i32, _vp]),
    "ams_k_global_mean": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp, _sz, _vp]),
    "ams_k_global_mean_scratch": (_sz, [_i32, _i32]),
    "ams_k_upsample_argmax": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, _i32, _vp, _vp, _vp,
                                        _vp, _vp]),
    "ams_k_ce_grad": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ams_k_ce_loss_grad": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ams_k_ce_loss_grad_scratch": (_sz, [_i32, _i32, _i32, _i32]),
    "ams_k_ce_loss_grad_soft": (C.c_int, [_vp, _i32, _i32, _i32, _i32, C.POINTER(_i32), _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "ams_debug_launch_table_needs_attr": (C.c_int, [_i32, C.c_uint64, _sz]),
    "ams_debug_reload_knobs": (C.c_int, []),
    "ams_debug_phase_cycles": (C.c_int, [C.c_int32, C.POINTER(C.c_uint64), C.c_int32]),
    "ams_k_pointwise_wgrad": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "ams_k_pointwise_wgrad_scratch": (_sz, [_i64, _i32, _i32]),
    "ams_k_pointwise_wgrad_split": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "ams_k_depthwise3x3_dgrad": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "ams_k_depthwise3x3_wgrad": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "ams_k_pointwise_xform": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ams_k_pointwise_wgrad_xform": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ams_k_pointwise_red": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _sz,
                                     C.POINTER(_i32), _vp, _sz, _vp]),
    "ams_k_depthwise3x3_fwd_bn_scratch": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "ams_k_depthwise3x3_fwd_bn": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _sz, C.POINTER(_i32), _vp]),
    "ams_k_depthwise3x3_dgrad_bn_scratch": (_sz, [_i32, _i32, _i32, _i32]),
    "ams_k_depthwise3x3_dgrad_bn": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _sz,
                                             C.POINTER(_i32), _vp]),
    "ams_k_depthwise3x3_fwd_bn_tiles_scratch": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "ams_k_depthwise3x3_fwd_bn_tiles": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _sz, C.POINTER(_i32), _vp]),
    "ams_k_depthwise3x3_dgrad_bn_apply_scratch": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "ams_k_depthwise3x3_dgrad_bn_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _sz,
                                                    C.POINTER(_i32), _vp]),
    "ams_k_xx_gram_scratch": (_sz, [_i64, _i32]),
    "ams_k_xx_gram": (C.c_int, [_vp, _i64, _i32, _vp, _sz, _vp, _vp, _vp]),
    "ams_k_expand_stats": (C.c_int, [_vp, _i32, _vp, _i32, C.c_double, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ams_k_xdw_train_scratch": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "ams_k_xdw_fwd_stats": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _sz, C.POINTER(_i32), C.POINTER(_i64), _vp]),
    "ams_k_xdw_bwd_reduce": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _sz,
                                      C.POINTER(_i32), C.POINTER(_i64), _vp]),
    "ams_k_xdw_bwd_dx": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ams_k_xdw_stem_scratch": (_sz, [_i32, _i32, _i32]),
    "ams_k_xdw_bwd_reduce_stem": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _sz,
                                           C.POINTER(_i32), C.POINTER(_i64), _vp]),
    "ams_k_xdw_dwe": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ams_k_adam": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _vp]),
}

_lib = None


class AmsHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libams_hip.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("AMS_HIP_LIB", str(LIB_PATH)))
    if not path.exists():
        raise AmsHipError(
            "HIP library %s not found: the AMS student has no CPU fallback. Build it with "
            "`make -C ams_amd/csrc` (hipcc --offload-arch=gfx950)." % path)
    # libams_hip.so needs libamdhip64.so.7; PyTorch-ROCm bundles its own copy under the same soname.  Whichever is
    # loaded first serves both, and device pointers are only valid inside one runtime instance: load torch first.
    import torch  # noqa: F401
    handle = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if handle.ams_abi_version() != ABI_VERSION:
        raise AmsHipError("libams_hip.so ABI %d, binding expects %d" % (handle.ams_abi_version(), ABI_VERSION))
    _lib = handle
    return handle


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ams_last_error()
        raise AmsHipError("%s failed (%d): %s" % (what or "libams_hip call", rc, msg.decode() if msg else "?"))
