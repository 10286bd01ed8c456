/*
 * ams_hip.h — C ABI of the MI355X-native AMS student hot path (libams_hip.so).
 *
 * The reference (modelstreaming/ams) has NO native boundary: its hot path is a Python object,
 * `SemanticNetwork` (reference SemanticNetwork.py:24), whose methods call `tf.Session.run` on a graph
 * assembled by `create_student_v3` (reference utils/graph_utils.py:338-533).  This header is the seam
 * a maintainer binds UNDER that class (ctypes stub in INTEGRATION.md): every entry point names the
 * reference call it replaces.  Plain pointers and sizes only; all `*_dev` pointers are HIP device
 * pointers owned by the caller (the Python host allocates them through PyTorch-ROCm); `stream` is a
 * `hipStream_t` passed as `void*` (0 = the null stream).  All functions return 0 on success or a
 * negative AMS_E_* code; `ams_last_error()` gives the message of the calling thread's last failure.
 *
 * Layouts: activations NHWC, dense; conv weights HWIO, depthwise weights HWC1 (TF layouts, so the
 * reference's `.npy` weight dicts load without transposition); frames `[B,H,W,3]` RGB 0..255 as uint8
 * or float32; label maps `[B,H,W]` (uint8 teacher ids in, int32 subset indices out).
 */
#ifndef AMS_HIP_H
#define AMS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMS_ABI_VERSION 3

enum {
    AMS_OK = 0,
    AMS_E_INVALID = -1,     /* bad argument (shape, dtype, mode, null pointer)           */
    AMS_E_HIP = -2,         /* a HIP runtime call or kernel launch failed                 */
    AMS_E_STATE = -3,       /* call not valid in this state (e.g. training a frozen net)  */
    AMS_E_NOMEM = -4        /* arena too small                                            */
};

/* layer roles: how the engine wires a row of the layer table (ams_amd/spec.py -> Appendix A of SURVEY.md) */
enum {
    AMS_ROLE_STEM = 0,       /* MobilenetV2/Conv: dense 3x3 s2 on the normalised frame          */
    AMS_ROLE_EXPAND = 1,     /* 1x1, BN, ReLU6                                                  */
    AMS_ROLE_DEPTHWISE = 2,  /* 3x3 depthwise (stride 1/2, rate 1/2), BN, ReLU6                 */
    AMS_ROLE_PROJECT = 3,    /* 1x1, BN, linear, optional residual add of the block input       */
    AMS_ROLE_POOL_CONV = 4,  /* image_pooling: global mean -> 1x1 -> BN -> ReLU (per image)     */
    AMS_ROLE_ASPP = 5,       /* aspp0: 1x1 -> BN -> ReLU                                        */
    AMS_ROLE_CONCAT_PROJ = 6,/* concat_projection over [pool branch, aspp0] -> BN -> ReLU       */
    AMS_ROLE_LOGITS = 7      /* logits/semantic: 1x1 + bias                                     */
};
enum { AMS_ACT_NONE = 0, AMS_ACT_RELU = 1, AMS_ACT_RELU6 = 2 };
enum { AMS_DT_F32 = 0, AMS_DT_U8 = 1, /* 2 was AMS_DT_BF16: bf16 activation storage, measured and dropped (profiles/r03_bf16_storage_study.json) */
       AMS_DT_I32 = 3, AMS_DT_F64 = 4 };
enum {
    AMS_MODE_FROZEN = 0,     /* BN with the statistics captured by ams_student_freeze, eps 1e-3 everywhere
                                (reference utils/graph_utils.py:52-76, :362-369; the graph the edge runs) */
    AMS_MODE_LIVE = 1        /* BN with batch statistics, per-layer eps (the live graph, is_training=True) */
};

/* one row of the layer table; offsets are in floats into the trainable / statistics arenas */
typedef struct ams_layer_desc {
    int32_t role;
    int32_t cin, cout;
    int32_t stride, rate;
    int32_t act;
    int32_t residual_from;   /* 1-based index of the layer whose output is added after BN, 0 = none */
    float bn_eps;            /* < 0: no BN (bias instead) */
    int64_t w_off;           /* weights    (trainable arena) */
    int64_t gamma_off;       /* BN gamma   (trainable arena), or bias offset when bn_eps < 0 */
    int64_t beta_off;        /* BN beta    (trainable arena) */
    int64_t mean_off;        /* moving_mean     (statistics arena) */
    int64_t var_off;         /* moving_variance (statistics arena) */
} ams_layer_desc;

typedef struct ams_student_config {
    int32_t abi_version;     /* AMS_ABI_VERSION */
    int32_t height, width;   /* frame size (reference: width = 2*height, run.py:71; not required here) */
    int32_t max_batch;       /* largest B any call will pass */
    int32_t num_classes;     /* 19 (Cityscapes) or 21 (VOC) */
    int32_t n_selected;      /* K: size of the per-video class subset */
    int32_t class_indices[32]; /* the K selected class ids, ascending (np.where(class_weights==1)) */
    int32_t n_layers;
    int32_t trainable;       /* 1: allocate activations/gradients/Adam state for ams_student_train_step */
    int32_t act_dtype;       /* AMS_DT_F32, the only storage type of activations.  bf16 storage of the output-stride-16 section was judged on
                                fine-tuned ("trained-like") weights, emulated by rounding (AMS_OPT_EMULATE_BF16_STORAGE): logits move by
                                8.4e-3 max / 3.3e-3 rms relative — outside the 1e-3 tolerance — although the labels change on 0.1 % of the
                                pixels only and mIoU by 0.002 pt (profiles/r03_bf16_storage_study.json); the bf16 variant that exists is
                                AMS_MATMUL_BF16 = bf16 products over f32 storage */
    int64_t n_trainable;     /* floats in the trainable arena (2 113 043 for Cityscapes) */
    int64_t n_stats;         /* floats in the statistics arena (33 088) */
    float bn_decay;          /* 0.9 (node BatchNorm/Const_2) */
    float bn_eps_frozen;     /* 1e-3 */
    float pixel_scale;       /* 1/127.5 as f32 (node mul_4/x) */
} ams_student_config;

typedef struct ams_student ams_student;

/* arena regions addressable from the host side (ams_student_region) */
enum {
    AMS_REGION_PARAMS = 0,   /* trainable arena, f32[n_trainable], tf.trainable_variables() order      */
    AMS_REGION_STATS = 1,    /* moving_mean / moving_variance, f32[n_stats]                             */
    AMS_REGION_GRADS = 2,    /* d loss / d params, f32[n_trainable] (valid after a train step phase)    */
    AMS_REGION_ADAM_M = 3,   /* f32[n_trainable]                                                         */
    AMS_REGION_ADAM_V = 4,   /* f32[n_trainable]                                                         */
    AMS_REGION_FROZEN = 5,   /* snapshot taken by ams_student_freeze: params followed by stats           */
    AMS_REGION_BN_SYNC = 6,  /* per-layer BN partial sums exchanged across ranks in data-parallel steps  */
    AMS_REGION_LOGITS = 7    /* low-resolution logits of the last forward, f32[B,h,w,num_classes]        */
};

const char* ams_last_error(void);
int ams_abi_version(void);
int ams_device_info(char* name_out, size_t name_cap, int32_t* n_cu, int64_t* hbm_bytes);

/* ---- lifecycle: replaces SemanticNetwork.__init__ (SemanticNetwork.py:32-159) / create_student_v3 ------- */
/* Bytes of device memory the engine needs for this configuration (activations, gradients, Adam, scratch). */
int ams_student_arena_bytes(const ams_student_config* cfg, const ams_layer_desc* layers, size_t* bytes_out);
/* Build the launch plan over caller-owned device memory `arena_dev` (>= arena_bytes, 256-byte aligned).
 * Parameters are NOT initialised: write them through ams_student_region views, then call
 * ams_student_freeze before the first AMS_MODE_FROZEN call. */
int ams_student_create(const ams_student_config* cfg, const ams_layer_desc* layers, void* arena_dev,
                       size_t arena_bytes, ams_student** out);
/* replaces SemanticNetwork.close_model (SemanticNetwork.py:716-717) */
void ams_student_destroy(ams_student* s);
/* byte offset into the arena and element count of a region */
int ams_student_region(const ams_student* s, int32_t region, size_t* offset_bytes, size_t* n_elems);
/* low-resolution feature size (h, w) = output stride 16 of the padded frame */
int ams_student_lowres_size(const ams_student* s, int32_t* h, int32_t* w);
/* Debugging / parity bisection (tools/grad_gap_bisect.py; no reference counterpart: tf.Session.run can fetch any tensor by name): byte offset
 * and element count of a per-layer training tensor of a TRAINABLE student, NHWC [max_batch, Hout, Wout, cout] — which = 0 raw conv output z,
 * 1 activated output a (= the next layer's input), 2 gradient wrt a (after a fine-tune step it holds dz where the step computed dz in
 * place), 3 gradient wrt z of a backbone layer (in place over 2, or a tensor of its own where 2 is read again as a skip gradient).  layer is 1-based (ams_layer_desc order).  Which of them a step actually writes depends on the fusion options: with every
 * AMS_OPT_TRAIN_* / FUSE_* fine-tune option at 0 all of them are. */
int ams_student_layer_tensor(const ams_student* s, int32_t layer, int32_t which, size_t* offset_bytes, size_t* n_elems);

/* ---- server -> edge hand-off: replaces save_to_frozen_graph / trim_graph_frozen / convert_batchnorms -----
 * (SemanticNetwork.py:706-714, utils/graph_utils.py:52-126).  Snapshots params + moving statistics into the
 * FROZEN region and folds every BN into (scale, shift) with eps = bn_eps_frozen.  Device-side, no protobuf. */
int ams_student_freeze(ams_student* s, void* stream);

/* ---- inference: replaces predict_input (SemanticNetwork.py:170-182), north-star `infer` ------------------
 * frames_dev: [B,H,W,3] uint8 or float32; labels_out_dev: int32 [B,H,W], index into the K selected classes. */
int ams_student_predict(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch,
                        int32_t mode, int32_t* labels_out_dev, void* stream);

/* ---- inference + metrics: replaces predict_with_metric (SemanticNetwork.py:196-213) ----------------------
 * teacher_dev: uint8 [B,H,W] teacher class ids (ids outside the subset are ignored, weight 0).
 * conf_mat_dev: int64 [K*K], rows = teacher, cols = student; OVERWRITTEN (the wrapper resets it per call).
 * loss_dev: double[2] = {sum of per-pixel CE over valid pixels, number of valid pixels}; loss = [0]/[1]. */
int ams_student_predict_with_metric(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch,
                                    int32_t mode, const uint8_t* teacher_dev, int32_t* labels_out_dev,
                                    int64_t* conf_mat_dev, double* loss_dev, void* stream);

/* ---- phi-score: replaces calc_cross_miou's confusion matrix (SemanticNetwork.py:124-139, :184-194) -------
 * labels_dev: uint8 [2,H,W] (before, after); conf_mat_dev: int64 [K*K] overwritten. */
/* The edge's per-frame call, several frames at a time: what ams_student_predict_with_metric computes for ONE frame, for each of `batch`
 * frames in one pass — labels [batch,H,W], conf_mats_dev [batch][K][K], losses_dev [batch][2] (teacher_dev NULL: labels only).  A frame at
 * a time the forward is ~45 dependent launches of 30-70 blocks on a 256-CU chip (0.49 ms); two frames take 0.59 ms, three 0.68 ms (MI355X):
 * a caller that can hold a frame for one or two frame times (33 ms at 30 fps) gets 1.7x / 2.2x the frames per second.  Every frame's result
 * is what a one-frame call returns (batch-composition invariance: tests/test_gpu_fullsize.py). */
int ams_student_predict_frames(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch, int32_t mode,
                               const uint8_t* teacher_dev, int32_t* labels_out_dev, int64_t* conf_mats_dev, double* losses_dev, void* stream);

/* The same with the label maps as uint8 [batch,H,W] (an index inside the subset of K <= 32 classes fits a byte): what a caller that takes the
 * results to the HOST uses — a quarter of the device -> host bytes of the int32 maps, widened on the host (SemanticNetwork.predict_input /
 * predict_with_metric return int32 as the reference does: the widening of 0.5 MB costs less there than copying 2 MB out of the pinned buffer). */
int ams_student_predict_frames_u8(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch, int32_t mode,
                                  const uint8_t* teacher_dev, uint8_t* labels_out_dev, int64_t* conf_mats_dev, double* losses_dev, void* stream);

int ams_cross_confusion(const ams_student* s, const uint8_t* labels_dev, int64_t n_pixels, int64_t* conf_mat_dev,
                        void* stream);

/* ---- one optimisation step: replaces sess.run({train, loss}) in _train (SemanticNetwork.py:253-260),
 * north-star `train_step`.  forward (BN batch stats) -> masked mean CE over the K classes -> backward ->
 * BN moving-average update (decay bn_decay) -> Adam (TF1 form, beta 0.9/0.999, eps 1e-8).
 * mask_dev: NULL for 'full_model', else uint8[n_trainable]: entries with 0 are reverted after the Adam step
 * while their moments still advance (utils/graph_utils.py:482-493).  loss_dev: double[2] as above (the loss of
 * the forward pass, before the update).  Adam's step counter lives in the handle and is never reset. */
int ams_student_train_step(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev,
                           int32_t batch, float lr, const uint8_t* mask_dev, double* loss_dev, void* stream);

/* Range of the default product form (AMS_MATMUL_SPLIT_F16): fp16 holds |x| < 65520.  ams_student_freeze checks the frozen weights of every layer
 * that has fp16 panels; a layer with a weight beyond 65504 (or a non-finite one) runs on three bf16 parts (f32's range) until a later freeze finds
 * it inside again — *n_layers = how many layers the last freeze moved.  Activations are not checked. */
int ams_student_f16_fallback_layers(const ams_student* s, int32_t* n_layers);

/* ---- create_student_v3's remaining kwargs (utils/graph_utils.py:338-339; run.py:150 leaves all three off) -------------------------------
 * soft_teacher=True (option AMS_OPT_SOFT_TEACHER): pixel_loss = softmax_cross_entropy_with_logits(logits = filtered_logits, labels =
 * softmax(gather(teacher_labels_logits_pl, class_weights))) (:375-376, 403-404) — per pixel sum_k p_k (logsumexp(z) - z_k), gradient
 * softmax(z) - p; the pixel mask and the mean's denominator still come from the hard labels (:397, 406-408), which every step keeps taking.
 * ams_student_feed_teacher_logits replaces feed_dict[student['teacher_labels_logits_pl']]: f32 [batch, th, tw, num_classes] on the device,
 * valid until the step that uses it has run; it stays armed until replaced or cleared with NULL.  th x tw = height x width is the reference's
 * feed (the loss needs the shape of filtered_logits); a smaller grid (cached low-resolution teacher logits: 1 / 256 of the bytes at output
 * stride 16) is interpolated to height x width like the student's own logits (ResizeBilinear, align_corners) — an extension, at the full size
 * the identity bit for bit.  A step with the option on and nothing fed fails (TensorFlow: "You must feed a value for placeholder tensor").
 *
 * regularize=True / train_biases_only=True: loss += coef * reduce_mean([l2_loss(v) for v in tvars]) with coef = 0.01, l2_loss(v) = sum(v^2) / 2,
 * tvars = the trainable variables, or only those without 'weight' in their name (:451-456; the optimizer still updates every variable — the
 * reference's minimize() has no var_list).  reg_mask_dev: uint8 [n_trainable], 1 on the entries of tvars (it stays referenced: keep it alive;
 * NULL switches the term off); n_vars = len(tvars).  The gradient term (coef / n_vars) v is added after the cross-rank gradient sum of a
 * data-parallel step, the loss term to loss_dev[0] scaled by the valid-pixel count, so that loss_dev[0] / loss_dev[1] is the fetched loss. */
int ams_student_feed_teacher_logits(ams_student* s, const float* teacher_logits_dev, int32_t th, int32_t tw);
int ams_student_set_regularizer(ams_student* s, const uint8_t* reg_mask_dev, int32_t n_vars, float coef);

/* ---- data-parallel split of the same step (one process per GPU) -------------------------
 * Callback form (any transport; what the gloo CPU tests and a host without RCCL use): `cb` is invoked on the host
 * whenever cross-rank sums are needed and must all-reduce (sum) `count` values of `dtype` at arena byte offset
 * `offset_bytes`, ordered after the work already enqueued on `stream`.  With cb == NULL the step is the single-GPU
 * step above.  Call sites: per-BN-layer (sum, sumsq) forward, (CE sum, valid-pixel count), per-BN-layer
 * (sum dy, sum dy*xhat) backward, and the flat gradient arena before Adam.  ams_student_train_step_rccl below is
 * the production form. */
typedef int (*ams_allreduce_cb)(void* user, size_t offset_bytes, size_t count, int32_t dtype);
int ams_student_train_step_dp(ams_student* s, const void* frames_dev, int32_t frames_dtype,
                              const uint8_t* teacher_dev, int32_t batch, int32_t global_batch, float lr,
                              const uint8_t* mask_dev, double* loss_dev, ams_allreduce_cb cb, void* user,
                              void* stream);

/* The same step with the exchange done by the library itself: every cross-rank sum is one ncclAllReduce on `stream`, issued in launch
 * order with no host round trip in between (110 per step: 54 BN forward, loss, 54 BN backward, the gradient arena).  librccl is
 * resolved at run time (dlopen; inside a PyTorch-ROCm process the copy PyTorch loaded is reused), so the library loads without it.
 *   rank 0: ams_comm_unique_id(id, 128) -> ship the 128 bytes to every rank (torch.distributed broadcast, a file, ...)
 *   all   : hipSetDevice(local gpu); ams_comm_create(id, 128, rank, world, &comm)   (world == 1: a no-op communicator)
 * One process per GPU, as the reference runs one process per --gpu (run.py:28). */
typedef struct ams_comm ams_comm;
int ams_comm_unique_id(uint8_t* id_out, size_t cap);
int ams_comm_create(const uint8_t* id, size_t id_len, int32_t rank, int32_t world, ams_comm** out);
void ams_comm_destroy(ams_comm* c);
int ams_comm_stats(const ams_comm* c, int32_t* rank, int32_t* world, int64_t* calls, int64_t* bytes);
/* Diagnostics of a multi-GPU run (bench.py `collective_ms_per_step`): with timing enabled every collective the communicator issues is
 * bracketed by a HIP event pair on its stream (the pair costs a few microseconds of stream time per collective: enable it for a few
 * diagnostic steps, not for the timed ones).  ams_comm_timing_read synchronises the device and returns the summed and the longest span and
 * their count since the last ams_comm_set_timing; a span is the time the stream spent INSIDE the collective (wait for the peers included). */
int ams_comm_set_timing(ams_comm* c, int32_t enable);
int ams_comm_timing_read(ams_comm* c, double* total_ms, double* max_ms, int64_t* spans);
int ams_comm_allreduce(ams_comm* c, void* buf_dev, size_t count, int32_t dtype, void* stream);
int ams_student_train_step_rccl(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev,
                                int32_t batch, int32_t global_batch, float lr, const uint8_t* mask_dev, double* loss_dev,
                                ams_comm* comm, void* stream);

/* ---- options -----------------------------------------------------------------------------------------------
 * AMS_OPT_MATMUL selects how the products of the 1x1-conv layers that would be matrix-pipe bound in exact f32 are formed
 * (few rows, a large weight panel, or >= 20 FLOP per byte: the output-stride-16 layers and the head):
 *   AMS_MATMUL_F32           exact f32 MFMA (v_mfma_f32_16x16x4_f32) everywhere
 *   AMS_MATMUL_SPLIT_BF16_X6 (default) operands split into three bf16 parts (all 24 significand bits), 6 bf16 MFMAs per 32 k,
 *                            f32 accumulate: products at f32 rounding level; 512x1024 logits 4e-5 from the f64 oracle, the
 *                            same as exact f32 and as the f32 CPU oracle (tools/logit_error.py)
 *   AMS_MATMUL_SPLIT_F16     frozen inference with two fp16 parts / 3 MFMAs, f32-level (see the enum below)
 *   AMS_MATMUL_SPLIT_BF16    frozen inference with two parts / 3 MFMAs: +5 % frames/s, ~1e-5 per layer, 2e-4 .. 5e-4 on
 *                            the logits (inside the 1e-3 tolerance, not at f32 level); the fine-tune step stays three-part. */
enum { AMS_OPT_SOFT_TEACHER = 25 /* fine-tune step: 1 = soft-teacher loss (ams_student_feed_teacher_logits); 0 (default) hard labels */,
       AMS_OPT_TRAIN_FWD_F16 = 24 /* fine-tune step under AMS_MATMUL_SPLIT_F16: 1 = the FORWARD 1x1 products of the split layers run on two fp16
                                      parts (3 MFMAs; the per-step weight split leaves fp16 planes beside the bf16 ones), the input-gradient and weight-
                                      gradient products stay on three bf16 parts (gradients span a range fp16 cannot hold); 0 (default) = three bf16
                                      parts everywhere.  Measured on MI355X: 7.455 -> 7.368 ms per 8-frame step — the step's GEMMs are not MFMA-bound
                                      at 17160 rows — so the default keeps one product form for the whole step */,
       AMS_OPT_WGRAD_FORK_EVERY = 23 /* fine-tune step with AMS_OPT_OVERLAP_WGRAD: weight gradients handed to the side stream n at a time, 1 .. 64
                                         (default 1 = each as soon as its operands exist).  A hand-over is an event on the main stream (a gap of
                                         6-8 us in the rocprofv3 timeline), and a weight gradient feeds only the optimizer, so it may start late —
                                         but measured on MI355X batching is SLOWER: 7.98 / 8.07 / 8.16-8.30 / 8.30 / 8.58 ms per step for n = 1 / 2 /
                                         4 / 8 / 16 (a weight gradient that runs beside the input-gradient GEMM of the same dz shares its cache
                                         lines).  Bit-identical for every n. */,
       AMS_OPT_FUSE_OPERAND_BN = 22 /* fine-tune step: 1 (default) BN + activation of every depthwise layer that feeds a project layer is applied by the
                                        CONSUMERS on their operand loads (project GEMM forward, project weight gradient backward: PwArgs / WgArgs
                                        x_mode 1) with the same IEEE operations in the same order, and the depthwise activation is never written:
                                        bit-identical to 0 = the pass written (bn_act).  The same move for dz = A dy + B + C z of the project and the
                                        stride-16 expand layers (x_mode / dy_mode 2, kept at kernel level: ams_k_pointwise_xform) measured SLOWER in
                                        the step on MI355X (8.31 -> 8.64 / 8.87 ms: those GEMMs are bound by their operand path) and is not wired in. */,
       AMS_OPT_NAN_GRADS = 18 /* fine-tune step on a batch WITHOUT a valid pixel: 0 (default) the reference's result — NaN loss (utils/graph_utils.py:408:
                                  tf.reduce_mean over the empty tf.boolean_mask) and ZERO gradients (the backward of that mean over a [0]-shaped tensor is
                                  empty, boolean_mask's gather gradient densifies it to zeros): parameters, Adam moments and BN statistics stay finite;
                                  1 = every gradient NaN instead, for callers who want such a batch to fail loudly */,
       AMS_OPT_OVERLAP_WGRAD = 19 /* fine-tune step: 1 (default) weight gradients on a side stream beside the input-gradient chain, 2 depthwise ones on a
                                     third stream (measured slower), 3 hand-overs alternate between two side streams (measured slower: 8.14 vs 7.98 ms),
                                     0 everything on the caller's stream.  Same bits */,
       AMS_OPT_OVERLAP_HEAD = 20 /* frozen inference: 1 = the image-pooling branch on a side stream beside the aspp0 GEMM; 0 (default): measured slower */,
       AMS_OPT_STREAM_MIN_ROWS = 21 /* frozen inference: rows (frames x pixels at the block's resolution) from which the streaming expand+depthwise
                                       kernels run (default 4096: from two 512x1024 frames per pass on; same bits either way) */,
       AMS_OPT_DUAL_AUTOTUNE = 12 /* with AMS_OPT_DUAL_STREAM = 1: 1 = pick the number of parts per batch size by TIMING the plans inside the first call
                                      with that batch size (median of three passes each; that call synchronises the host and its result then depends
                                      on which plan won); 0 (default) = the static rule: the same call always runs the same plan */,
       AMS_OPT_DUAL_PARTS = 13 /* parts (2 .. 4) of the forced split, AMS_OPT_DUAL_STREAM = n >= 2 */,
       AMS_OPT_EMULATE_BF16_STORAGE = 15 /* STUDY ONLY (tools/bf16_storage_study.py), default 0: frozen inference rounds the depthwise results and
                                            the block inputs of the output-stride-16 section to bf16 after they are written — the values bf16
                                            STORAGE of those tensors would hold; the arithmetic and every other tensor stay f32.  Costs a pass per
                                            tensor: an accuracy probe, not a fast path */,
       AMS_OPT_FUSE_DGRAD_BN = 16 /* fine-tune step, stride-1 blocks that keep their tensors (blocks 7-16): >= 1 the depthwise input gradient,
                                     the expand layer's activation derivative and BN-backward sums, and the depthwise weight gradient come out of
                                     ONE kernel (k_conv.hip: dw3x3_dgrad_bn_kernel); 2 (default) the forward too: the depthwise conv applies the
                                     expand layer's BN + activation on its tap loads and leaves the statistics of its result (dw3x3_fwd_bn_kernel) —
                                     the expand activation is never written; 3 (default) the depthwise layer's OWN apply pass (dz_d = A dy + B + C z_d) is folded into
                                     that backward kernel too, through an LDS ring (k_dw_train.hip); 0 separate passes.  Same mathematics, f32-level differences */,
       AMS_OPT_FUSE_GEMM_RED = 17 /* fine-tune step, BN column reductions in the epilogue of the 1x1 GEMM that holds the values in registers
                                     (pw_common.hpp pw_red_*): bit 0 the forward statistics of the GEMM's own result, bit 1 the BN-backward sums of the
                                     layer whose output gradient the dgrad GEMM produces.  Default 3; 0 = separate reduction passes.  Same mathematics,
                                     f32-level differences (partial sums per row strip instead of per column chunk) */,
       AMS_OPT_TRAIN_RECOMPUTE = 11 /* fine-tune step: >= 1 the early blocks (block input <= 32 channels) run without their 6x-expanded
                                       tensors — every consumer recomputes z_e = x . W_e from the block input (k_xdw_train.hip); 0 the
                                       layer-by-layer step (every tensor materialised); 2 (default) the expand layer's BN statistics additionally come from the
                                       Gram matrix of the block input (z_e is linear in x: sum z = g0 . w, sum z^2 = w^T XX w, XX accumulated in f64 in one
                                       cheap pass over x, k_xx_stats.hip) instead of a pass that recomputes z_e.  Same mathematics, f32-level differences. */,
       AMS_OPT_DUAL_STREAM = 10 /* frozen inference: a batch as two to four parts on as many streams (the caller's and up to three the student
                                   owns, created with the student; one fork and one join per call), each frame computed exactly as in a batch of the
                                   part's size.  0 never; 1 (default) a fixed function of the batch size (from 8 frames on two parts, three at 12, 24 and
                                   48 frames: 1-8 % at 512x1024, tools/sweep_parts.py) — nothing is timed, the call never
                                   synchronises; n >= 2 always AMS_OPT_DUAL_PARTS parts from n frames on (the caller decides). */,
       AMS_OPT_BLOCK_X6 = 8 /* whole-block kernels: 1 (default) the expand products of the blocks with 24 / 32 input channels, and the stem's
                               products in the one-kernel first block (operands from a 258-entry table of the normalised byte values), run as six bf16
                               MFMAs on three-part splits (f32-level, 96 instead of 256 matrix-pipe cycles per 16x16 tile); 0 exact f32 MFMA
                               (bit-identical to the layer-by-layer plan).  Always exact f32 under AMS_MATMUL_F32. */,
       AMS_OPT_LATE_SUBBATCH = 7 /* frozen inference: frames per pass of the output-stride-16 section (blocks 7-16 and the head); 0 = the whole
                                    batch.  Same results bit for bit; a pass whose largest tensor fits the 256 MB Infinity Cache keeps the
                                    writer / reader pairs of that section out of HBM */,
       AMS_OPT_FUSE_BLOCK = 6 /* frozen inference: 1 (default) every early inverted-residual block with Cin <= 32 (expand + depthwise +
                                 project [+ block input]) runs as ONE kernel: neither the 6x-expanded tensor nor the depthwise result reaches
                                 HBM; bit-identical to the layer-by-layer plan.  0: the per-layer / pairwise-fused kernels below */,
       AMS_OPT_FUSE_FIRST_BLOCK = 4 /* frozen inference, stem + depthwise + project of the first block: 0 three kernels, 1 one kernel
                                       with a tile per block (default), 2 one kernel with a tile per wave (measured slower); same bits */,
       AMS_OPT_FUSE_DW_PROJECT = 3 /* frozen inference: 1 = depthwise + project of the stride-16 blocks in one kernel (needs
                                      AMS_MATMUL_SPLIT_BF16), 0 (default) separate kernels: measured no faster */,
       AMS_OPT_FUSE_EXPAND_DW_STREAM = 5 /* frozen inference, split-bf16 modes: expand + depthwise of the stride-16 blocks in one
                                            streaming kernel, bit-identical to the two kernels it replaces: 0 never, 1
                                            (default) where measured faster (Cin 64 / 96, batches of >= 16384 pixels at
                                            that stride), 2 every supported block (Cin 160 too) */,
       AMS_OPT_MATMUL = 1, AMS_OPT_FUSE_EXPAND_DW = 2 /* 0 never, 1 (default) blocks where the fused kernel is faster, 2 every supported block */ };
enum { AMS_MATMUL_F32 = 0, AMS_MATMUL_SPLIT_BF16 = 1, AMS_MATMUL_SPLIT_BF16_X6 = 2,
       AMS_MATMUL_SPLIT_F16 = 4 /* frozen inference: every operand of those layers as TWO fp16 parts, x ~ hi + lo 2^-11 (22 significand bits),
                                   products hi hi + 2^-11 (hi lo + lo hi) on v_mfma_f32_16x16x32_f16 with the cross terms in an accumulator of
                                   their own: 3 MFMAs per 32 k instead of 6, and 4 bytes per value where parts are stored instead of 6 — the
                                   depthwise result of the stride-16 blocks is handed to the project GEMM as ready-made fp16 pairs at the bytes of
                                   f32.  Per product ~3 2^-22 against 2^-24: below the f32 accumulation error of these contractions (512x1024
                                   logits 4e-5 from the f64 oracle, as the three-part form).  Operands must lie within fp16's range (|x| < 65504:
                                   activations and weights of O(1)); the fine-tune step keeps the three-part bf16 form (its gradients span a range
                                   fp16 cannot hold). */,
       AMS_MATMUL_BF16 = 3 /* opt-in bf16 inference variant (BASELINE.json configs[1] says "bf16"): the same late layers with ONE bf16 part per
                              operand = plain bf16 products, f32 accumulate, 1 MFMA per 32 k.  Storage stays f32 and the early blocks stay
                              exact f32.  NOT within the f32 tolerance: bench.py reports its label-mismatch fraction and mIoU delta against
                              the default plan beside its speed; frozen inference only (the fine-tune step ignores it). */ };
int ams_student_set_option(ams_student* s, int32_t option, int32_t value);

/* ---- measurement hook (bench.py roofline leg) -----------------------------------------------------------
 * With profiling enabled every kernel launch of the engine is bracketed by HIP events on the launch stream.
 * ams_student_profile_read synchronises and writes one line per launch: "kernel\tlayer\tms\talgorithmic_bytes\talgorithmic_flops\n"
 * (flops = 0 where the launch is priced by bytes only)
 * (kernel named as rocprofv3 prints it, without namespace and argument list).  `needed` receives the size. */
int ams_student_profile(ams_student* s, int32_t enable);
int ams_student_profile_read(ams_student* s, char* buf, size_t cap, size_t* needed);

/* Adam step counter (t of SURVEY.md Appendix C.10); exposed so host-side checkpoints can carry it. */
int ams_student_get_adam_step(const ams_student* s, int64_t* t);
int ams_student_set_adam_step(ams_student* s, int64_t t);

/* ---- model delta for the downlink: replaces the host loop of run.py:316-336 ------------------------------
 * Casts the masked parameters to fp16 in trainable order: out_half_dev receives sum(mask) halves, and
 * n_out_dev (int64) the count.  mask_dev NULL = all parameters. */
int ams_pack_masked_fp16(const float* params_dev, const uint8_t* mask_dev, int64_t n, uint16_t* out_half_dev,
                         int64_t* n_out_dev, int64_t* scratch_dev, size_t scratch_elems, void* stream);
/* int64 elements of caller-owned device scratch the call above needs (segment counts; nothing is allocated inside) */
size_t ams_pack_masked_fp16_scratch(int64_t n);

/* =====================================================================================================
 * Kernel-level entry points.  Same kernels the engine launches, exposed one by one so that tests/ can
 * check each against the oracle (SURVEY.md §4 test pyramid level 1).  x/y/... are device pointers.
 * ===================================================================================================== */

/* K1+K2: pad(127.5) -> x/127.5-1 -> dense 3x3 stride 2 SAME (3 -> cout) -> y*scale+shift -> act.
 * scale/shift NULL: raw conv output.  stats_dev != NULL: also emit per-channel shifted (sum, sumsq) partials. */
int ams_k_stem_conv(const void* frames, int32_t frames_dtype, int32_t B, int32_t H, int32_t W, const float* w /*[3,3,3,cout]*/,
                    int32_t cout, const float* scale, const float* shift, int32_t act, float pixel_scale,
                    float* y /*[B,Ho,Wo,cout]*/, void* stream);

/* K3: depthwise 3x3, NHWC, SAME, stride 1|2, rate 1|2 -> y*scale+shift -> act (scale NULL: raw). */
int ams_k_depthwise3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w /*[3,3,C,1]*/,
                       int32_t stride, int32_t rate, const float* scale, const float* shift, int32_t act,
                       float* y, void* stream);

/* K4: 1x1 conv as GEMM: y[M,N] = act((x[M,K] @ w[K,N] + img_bias[row/rows_per_img]) * scale + shift) + res.
 * Any of img_bias/scale/shift/res may be NULL.  trans_w != 0: w is given as [N,K] (dgrad). */
int ams_k_pointwise(const float* x, int64_t M, int32_t K, const float* w, int32_t N, int32_t trans_w,
                    const float* img_bias, int64_t rows_per_img, const float* scale, const float* shift,
                    int32_t act, const float* res, float* y, void* stream);

/* K4, split-bf16 form (see AMS_MATMUL_SPLIT_BF16): y = act((x @ w) * scale + shift) + res with w [K,N] f32 split on the
 * fly into bf16 hi/lo panels held in `panels` (uint16, >= 2*N*roundup(K,32) elements).  K % 8 == 0. */
int ams_k_pointwise_split(const float* x, int64_t M, int32_t K, const float* w, int32_t N, const float* scale,
                          const float* shift, int32_t act, const float* res, float* y, uint16_t* panels,
                          size_t panel_elems, void* stream);

/* The same with the three-part split (hi, mid, lo: all 24 significand bits, 6 bf16 MFMAs per 32 k; AMS_MATMUL_SPLIT_BF16_X6);
 * panels >= 3*N*roundup(K,32) uint16. */
int ams_k_pointwise_split3(const float* x, int64_t M, int32_t K, const float* w, int32_t N, const float* scale,
                           const float* shift, int32_t act, const float* res, float* y, uint16_t* panels,
                           size_t panel_elems, void* stream);

/* K4+K3 fused (frozen inference): y = relu6(bn_d(dw3x3(relu6(bn_e(x @ w_exp))))) for an inverted-residual block whose
 * input has Cin <= 64 channels; the expanded tensor never reaches HBM.  x [B,H,W,Cin], w_exp [Cin,Cexp], w_dw [3,3,Cexp,1],
 * y [B,Ho,Wo,Cexp] (SAME, stride 1|2, rate 1).  Cexp % 32 == 0 or % 48 == 0. */
int ams_k_expand_dw(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                    const float* shift_e, int32_t Cexp, const float* w_dw, int32_t stride, int32_t rate, const float* scale_d,
                    const float* shift_d, float* y, void* stream);

/* A whole early block in one kernel (k_block.hip): y = bn_p(relu6(bn_d(dw3x3(relu6(bn_e(x @ w_exp))))) @ w_proj) (+ x when residual != 0:
 * stride 1 and Cout == Cin only).  Cin % 4 == 0 and <= 32, Cexp % 16 == 0 and <= 384, Cout % 4 == 0 and <= 64, stride 1|2, rate 1
 * (the launcher picks the largest measured tile that fits 64 KB of LDS; AMS_E_INVALID for any other shape).
 * panels == NULL: exact f32 products in the k order of ams_k_pointwise — the same bits as ams_k_expand_dw followed by ams_k_pointwise.
 * panels != NULL (scratch of >= 3*Cexp*32 uint16) and Cin > 16: the EXPAND products as six bf16 MFMAs on three-part splits (all 24
 * significand bits, f32-level: what the engine does by default, AMS_OPT_BLOCK_X6); depthwise and project stay exact f32. */
int ams_k_block_fused(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e, const float* shift_e,
                      int32_t Cexp, const float* w_dw, int32_t stride, const float* scale_d, const float* shift_d, const float* w_proj, int32_t Cout,
                      const float* scale_p, const float* shift_p, int32_t residual, float* y, uint16_t* panels, size_t panel_elems, void* stream);

/* K4+K3 fused, streaming form for the stride-16 blocks (Cin in {64, 96, 160}, stride 1, rate 1|2, Cexp % 32 == 0): the same
 * result as ams_k_pointwise_split (parts = 2) / the three-part split (parts = 3) followed by ams_k_depthwise3x3, bit for bit,
 * without writing the expanded tensor.  panels: scratch of >= 3*Cexp*Cin uint16 (w_exp [Cin,Cexp] is split into it).
 * Cin in {16, 24, 32}: exact-f32 products instead (bit-identical to ams_k_pointwise + ams_k_depthwise3x3; parts / presplit /
 * panels unused), and rate = -2 selects the stride-2 depthwise conv (y [B,Ho,Wo,Cexp], SAME).
 * presplit != 0: x is first written as bf16 parts (what the project GEMM of the previous block leaves behind in the engine) and
 * the kernel loads its operand from them; panels then needs 3*B*H*W*Cin more elements.  Same result, bit for bit. */
int ams_k_expand_dw_stream(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                           const float* shift_e, int32_t Cexp, const float* w_dw, int32_t rate, const float* scale_d,
                           const float* shift_d, float* y, uint16_t* panels, size_t panel_elems, int32_t parts, int32_t presplit,
                           void* stream);

/* The two-fp16-part forms (AMS_MATMUL_SPLIT_F16; k_pw_f16.hip, split_bf16.hpp): every operand x ~ hi + lo 2^-11 with hi = f16(x),
 * lo = f16((x - hi) 2^11); products hi hi + 2^-11 (hi lo + lo hi), 3 MFMAs per 32 k, f32 accumulate.
 * ams_k_pack_h2i: f32 [M][C] (C % 8 == 0) -> "H2I": per 8 channels 16 bytes of hi then 16 bytes of lo, 4 C bytes per row (what the streaming
 *   kernels write with y_h2i and the GEMM reads with x_h2i).
 * ams_k_pointwise_split_f16: as ams_k_pointwise_split3 (K4; replaces the Conv2D nodes of model.meta `expanded_conv_N/project`, `aspp0`, ...);
 *   panels >= 2*N*Kp uint16; x_h2i (optional, M*K floats of scratch): x is first packed into H2I and the GEMM loads its operand from there —
 *   same result, bit for bit (x_h2i == x: x is taken as packed already); y_parts (optional, 2*M*N uint16): the result also as two fp16 part planes.
 * ams_k_expand_dw_stream_f16: as ams_k_expand_dw_stream with the fp16 parts (Cin 64 / 96 / 160); presplit 1 | 2 as there (2 = the
 *   weight-register form; panels then needs 2*B*H*W*Cin more elements); y_h2i != 0: y is written in H2I instead of f32.  Bit-identical to
 *   ams_k_pointwise_split_f16 followed by ams_k_depthwise3x3. */
int ams_k_pack_h2i(const float* x, int64_t M, int32_t C, float* out, void* stream);
/* ams_k_block_fused with the expand AND the project products on two fp16 parts (3 MFMAs each; K = 16 included); panels: scratch of
 * >= 2*Cexp*32 + 2*Cout*roundup(Cexp, 32) uint16 */
int ams_k_block_fused_f16(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                          const float* shift_e, int32_t Cexp, const float* w_dw, int32_t stride, const float* scale_d, const float* shift_d,
                          const float* w_proj, int32_t Cout, const float* scale_p, const float* shift_p, int32_t residual, float* y,
                          uint16_t* panels, size_t panel_elems, void* stream);
int ams_k_pointwise_split_f16(const float* x, int64_t M, int32_t K, const float* w, int32_t N, const float* scale, const float* shift,
                              int32_t act, const float* res, float* y, uint16_t* panels, size_t panel_elems, float* x_h2i, uint16_t* y_parts,
                              void* stream);
int ams_k_expand_dw_stream_f16(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                               const float* shift_e, int32_t Cexp, const float* w_dw, int32_t rate, const float* scale_d,
                               const float* shift_d, float* y, uint16_t* panels, size_t panel_elems, int32_t presplit, int32_t y_h2i,
                               void* stream);

/* K3+K4 fused (frozen inference): y = bn_p(relu6(bn_d(dw3x3(e))) @ w_proj) (+ res) — the depthwise output never reaches
 * HBM; the product uses the two-part bf16 split of ams_k_pointwise_split.  e [B,H,W,C] (C % 32 == 0), w_dw [3,3,C,1],
 * stride 1, rate 1|2, w_proj [C,N] with N % 16 == 0 and N/16 in {1..6, 8, 10} or a multiple of 10 or 8 of those; res/y
 * [B,H,W,N]; panels: scratch of >= 2*N*C uint16. */
int ams_k_dw_project(const float* e, int32_t B, int32_t H, int32_t W, int32_t C, const float* w_dw, int32_t rate,
                     const float* scale_d, const float* shift_d, const float* w_proj, int32_t N, const float* scale_p,
                     const float* shift_p, const float* res, float* y, uint16_t* panels, size_t panel_elems, void* stream);

/* Frame ingest (reference run.py:179-183, :415-421: cv2.resize of the decoded frame to [H, 2H] + BGR->RGB; INTER_NEAREST for
 * the teacher label map).  src [Hs,Ws,C] uint8 -> dst [H,W,C] uint8, both device memory.  mode AMS_RESIZE_NEAREST |
 * AMS_RESIZE_LINEAR (OpenCV's 8-bit fixed-point INTER_LINEAR: 11-bit weights, integer passes, the 2x box-average shortcut; bit-identical
 * to the restatement in oracle/cv_resize.py, which hand-derived vectors pin); swap_rb != 0 reverses the 3 channels. */
enum { AMS_RESIZE_NEAREST = 0, AMS_RESIZE_LINEAR = 1 };
int ams_ingest_resize_u8(const uint8_t* src, int32_t Hs, int32_t Ws, int32_t C, int32_t mode, int32_t swap_rb, uint8_t* dst,
                         int32_t H, int32_t W, void* stream);

/* K7: global average pool [B,HW,C] -> [B,C] (two-stage, deterministic); scratch >= ams_k_global_mean_scratch floats. */
int ams_k_global_mean(const float* x, int32_t B, int64_t HW, int32_t C, float* y, float* scratch, size_t scratch_floats,
                      void* stream);
size_t ams_k_global_mean_scratch(int32_t B, int32_t C);

/* K9-K12: bilinear(align_corners) upsample of low-res logits to HxW fused with class gather, argmax and
 * (when teacher != NULL) confusion matrix + CE loss sums.  Full-resolution logits are never written. */
int ams_k_upsample_argmax(const float* logits /*[B,h,w,NC]*/, int32_t B, int32_t h, int32_t w, int32_t NC,
                          const int32_t* class_idx_host, int32_t K, int32_t H, int32_t W, const uint8_t* teacher,
                          int32_t* labels_out, int64_t* conf_mat, double* loss, void* stream);

/* K11 backward: d loss / d low-res logits (zeros for unselected classes); loss_and_count_dev: the double[2]
 * written by ams_k_upsample_argmax ([1] = number of valid pixels, the mean's denominator).
 * class_idx_host (here and above): HOST array of the K selected class ids; it travels by value. */
int ams_k_ce_grad(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host,
                  int32_t K, int32_t H, int32_t W, const uint8_t* teacher, const double* loss_and_count_dev,
                  float* dlogits, void* stream);

/* K11 forward + backward in ONE pass over the pixels (what the fine-tune step runs; replaces graph_utils.py:403-408 and its
 * gradient): loss_dev[0] = CE sum over valid pixels, loss_dev[1] = their number, dlogits [B*h*w, NC] = d(mean CE) / d low-res
 * logits (zeros for unselected classes).  Every pixel's softmax is evaluated once; no atomics touch the gradient (run-to-run
 * identical).  scratch: >= ams_k_ce_loss_grad_scratch floats.  AMS_E_INVALID when one source column spans more than ~20 output
 * columns (callers then use ams_k_upsample_argmax + ams_k_ce_grad). */
int ams_k_ce_loss_grad(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host,
                       int32_t K, int32_t H, int32_t W, const uint8_t* teacher, double* loss_dev, float* dlogits,
                       float* scratch, size_t scratch_floats, void* stream);
size_t ams_k_ce_loss_grad_scratch(int32_t B, int32_t h, int32_t w, int32_t K);
/* The same with soft-teacher targets: teacher_logits f32 [B, th, tw, NC] (ams_student_feed_teacher_logits); `teacher` still masks the pixels. */
int ams_k_ce_loss_grad_soft(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host,
                            int32_t K, int32_t H, int32_t W, const uint8_t* teacher, const float* teacher_logits, int32_t th, int32_t tw,
                            double* loss_dev, float* dlogits, float* scratch, size_t scratch_floats, void* stream);

/* K13: weight gradient of a 1x1 conv: dw[K,N] = x[M,K]^T @ dy[M,N]. scratch: >= ams_k_pointwise_wgrad_scratch floats */
int ams_k_pointwise_wgrad(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, float* dw,
                          float* scratch, size_t scratch_floats, void* stream);
size_t ams_k_pointwise_wgrad_scratch(int64_t M, int32_t K, int32_t N);
/* K13b: the same on the bf16 matrix pipe with both operands split into three bf16 parts (six products, f32-level
 * accuracy): the kernel the fine-tune step uses for its late layers (few pixels, many channel pairs).  Requires
 * 1024 <= M <= 32768, K and N multiples of 4, K*N >= 4096 (AMS_E_INVALID otherwise). */
int ams_k_pointwise_wgrad_split(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, float* dw,
                                float* scratch, size_t scratch_floats, void* stream);

/* K13: depthwise backward: dx (input gradient) and dw[3,3,C,1]. */
int ams_k_depthwise3x3_dgrad(const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, const float* w,
                             int32_t stride, int32_t rate, float* dx, void* stream);
int ams_k_depthwise3x3_wgrad(const float* x, const float* dy, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t stride, int32_t rate, float* dw, float* scratch, size_t scratch_floats,
                             void* stream);

/* A 1x1 conv of the fine-tune step with a BN column reduction in its epilogue (pw_common.hpp pw_red_*; what FusedBatchNormV3 /
 * FusedBatchNormGradV3's reductions are to the reference's graph).  y [M,N] = x [M,K] . w [K,N] (trans_w: w is [N,K], the dgrad orientation).
 *   mode 1: partial rows [rows][2][N] of sum(y - center), sum((y - center)^2)            (forward statistics; center may be NULL)
 *   mode 2: y is the gradient wrt the ACTIVATED output of a BN layer with raw output z [M,N]: y (+ res [M,N] first, if given) is multiplied by
 *           act'(z scale + shift) BEFORE it is stored; partial rows of sum(y), sum(y (z - mean) rstd)   (first half of that layer's BN backward)
 * split 1: the three-part bf16 kernel (panels: >= 3 N Kp bf16 scratch); split 2: the two-fp16-part kernel (mode 1 only: what the fine-tune step's
 * forward runs under AMS_MATMUL_SPLIT_F16; panels >= 2 N Kp); 0: the exact-f32 streaming kernel.  *rows_out = 0 means the kernel
 * chosen for this shape cannot fuse the reduction (y is then the plain product, + res): the caller runs the separate pass.
 * part_floats >= (M / 64 + 8 or 2048) * 2 N. */
int ams_k_pointwise_red(const float* x, int64_t M, int32_t K, const float* w, int32_t N, int32_t trans_w, int32_t split, int32_t mode,
                        const float* center, const float* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                        int32_t act, const float* res, float* y, float* part, size_t part_floats, int32_t* rows_out, uint16_t* panels,
                        size_t panel_elems, void* stream);

/* The 1x1 kernels of the fine-tune step with an elementwise BN pass applied on their OPERAND loads instead of being written first
 * (PwArgs / WgArgs x_mode, dy_mode; AMS_OPT_FUSE_OPERAND_BN): same IEEE operations in the same order as the pass, so the result is
 * bit-identical to running the pass and then the plain kernel.
 *   ams_k_pointwise_xform: y [M,N] = x' . w with x' = act(x * v0[k] + v1[k]) (x_mode 1: BN + activation, what FusedBatchNormV3 + Relu6 are to
 *     the reference's graph) or x' = v0[k] * x + v1[k] + v2[k] * x2 (x_mode 2: the second half of FusedBatchNormGradV3, (v0, v1, v2) = (A, B, C),
 *     x2 = the BN layer's raw output).  split 1: the three-part bf16 kernel (panels >= 3 N Kp), split 2 (x_mode 1 only): the two-fp16-part kernel, else the exact-f32 kernels.  A kernel that
 *     cannot transform on load writes x' into x_tmp [M,K] first (may be NULL: then such shapes fail with AMS_E_INVALID).
 *   ams_k_pointwise_wgrad_xform: dw [K,N] = x'^T . dy' with x' as x_mode 1 (x_mode 0: x' = x) and dy' = d0[n] * dy + d1[n] + d2[n] * dy2
 *     (dy_mode 2; dy_mode 0: dy' = dy); split != 0 the six-product bf16 kernel (shapes as ams_k_pointwise_wgrad_split). */
int ams_k_pointwise_xform(const float* x, int64_t M, int32_t K, const float* w, int32_t N, int32_t trans_w, int32_t split, int32_t x_mode,
                          int32_t x_act, const float* v0, const float* v1, const float* v2, const float* x2, float* y, float* x_tmp,
                          uint16_t* panels, size_t panel_elems, void* stream);
int ams_k_pointwise_wgrad_xform(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, int32_t split, int32_t x_mode, int32_t x_act,
                                const float* v0, const float* v1, int32_t dy_mode, const float* d0, const float* d1, const float* d2,
                                const float* dy2, float* dw, float* scratch, size_t scratch_floats, void* stream);

/* The fine-tune step's one-kernel forms of a stride-1 depthwise layer inside a block that keeps its tensors (what tf.gradients spreads over
 * FusedBatchNormV3 / Relu6 / DepthwiseConv2dNative and their Grad ops, SemanticNetwork.py:253-260 via utils/graph_utils.py:457-496):
 *   forward: zd [B,H,W,C] = dwconv3x3(act(ze * scale + shift), w) (rate 1 | 2, SAME: the ACTIVATION is zero-padded), and partial rows
 *            [rows][2][C] of sum(zd - center), sum((zd - center)^2) (center may be NULL) in scratch; *rows_out = rows;
 *   backward: out [B,H,W,C] = dwconv3x3^T(dz, w) * act'(z_prev * scale + shift), and partial rows [rows][11][C] of sum(out),
 *            sum(out * (z_prev - mean) * rstd) and the nine taps of the depthwise weight gradient sum(act(z_prev * scale + shift) . dz).
 * scratch_floats must be at least the matching *_scratch(). */
size_t ams_k_depthwise3x3_fwd_bn_scratch(int32_t B, int32_t H, int32_t W, int32_t C, int32_t rate);
int ams_k_depthwise3x3_fwd_bn(const float* ze, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t rate,
                              const float* scale, const float* shift, int32_t act, const float* center, float* zd, float* scratch,
                              size_t scratch_floats, int32_t* rows_out, void* stream);
size_t ams_k_depthwise3x3_dgrad_bn_scratch(int32_t B, int32_t H, int32_t W, int32_t C);
int ams_k_depthwise3x3_dgrad_bn(const float* dz, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t rate,
                                const float* z_prev, const float* scale, const float* shift, int32_t act, const float* mean,
                                const float* rstd, float* out, float* scratch, size_t scratch_floats, int32_t* rows_out, void* stream);
/* ... with the depthwise layer's own BN-backward apply pass folded in (AMS_OPT_FUSE_DGRAD_BN = 3, k_dw_train.hip): the gradient entering the
 * transposed conv is dz_d = cA dy + cB + cC zd, formed once per element on the way into an LDS ring (the unfused operations of the apply pass:
 * `out` is bit-identical to ams_k_depthwise3x3_dgrad_bn on the materialised dz_d); partial rows [rows][11][C] as above, split differently. */
size_t ams_k_depthwise3x3_dgrad_bn_apply_scratch(int32_t B, int32_t H, int32_t W, int32_t C, int32_t rate);
/* the forward in the same LDS-tile form (act(ze scale + shift) formed once per element; zd bit-identical to ams_k_depthwise3x3_fwd_bn) */
size_t ams_k_depthwise3x3_fwd_bn_tiles_scratch(int32_t B, int32_t H, int32_t W, int32_t C, int32_t rate);
int ams_k_depthwise3x3_fwd_bn_tiles(const float* ze, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t rate, const float* scale,
                                    const float* shift, int32_t act, const float* center, float* zd, float* scratch, size_t scratch_floats,
                                    int32_t* rows_out, void* stream);
int ams_k_depthwise3x3_dgrad_bn_apply(const float* dy, const float* zd, const float* cA, const float* cB, const float* cC, int32_t B, int32_t H, int32_t W,
                                      int32_t C, const float* w, int32_t rate, const float* z_prev, const float* scale, const float* shift, int32_t act,
                                      const float* mean, const float* rstd, float* out, float* scratch, size_t scratch_floats, int32_t* rows_out,
                                      void* stream);

/* The fine-tune step of an early inverted-residual block WITHOUT its 6x-expanded tensors (k_xdw_train.hip; Cin 8..32, Cexp 32..192 in 16s,
 * depthwise stride 1 | 2, rate 1): every pass recomputes z_e = x . w_exp from the block input x [B,H,W,Cin].  KP = Cin rounded up to 16.
 *   fwd_stats : partial rows  S [2][Cexp] = sum(z_e - center), sum((z_e - center)^2) | XX [KP][KP] = x^T x | g0 [KP] = sum x
 *   bwd_reduce: with dz_d = gradient wrt the depthwise layer's raw output [B,Ho,Wo,Cexp]: dy_e = dwconv^T(dz_d) . act'(z_e sc + sh);
 *               partial rows  S [2][Cexp] = sum dy_e, sum dy_e xhat_e | dWd [9][Cexp] (depthwise weight gradient) | G1 [KP][Cexp] = x^T dy_e
 *   bwd_dx    : dx [B,H,W,Cin] = (cA dy_e + cB + cC z_e) . w_exp^T (+ res)
 *   dwe       : dw_exp [Cin][Cexp] = cA G1 + g0^T cB + cC (XX . w_exp) from the REDUCED rows (G1; XX | g0 contiguous)
 * rows of *stride_out floats, *rows_out of them, in scratch (>= ams_k_xdw_train_scratch floats). */
/* BN statistics of an early block's expand layer from the Gram matrix of the block input (AMS_OPT_TRAIN_RECOMPUTE = 2, k_xx_stats.hip; what
 * FusedBatchNormV3's reduction over z_e = x . W_e is to the reference's graph): ams_k_xx_gram forms XX = x^T x [KP][KP] and g0 = sum x [KP]
 * (KP = Cin rounded up to 16, Cin <= 32) over the M rows of x [M, Cin] on the f64 matrix pipe — xx64: KP KP + KP doubles, xx32 (may be NULL) the
 * same as floats; ams_k_expand_stats turns them into scale / shift / saved mean / rstd (and the moving averages, sums [2][Cexp] about
 * `center`, both optional) of the BN over n pixels: sum z = g0 . w, sum z^2 = w^T XX w per channel, in f64. */
size_t ams_k_xx_gram_scratch(int64_t M, int32_t Cin);
int ams_k_xx_gram(const float* x, int64_t M, int32_t Cin, double* scratch, size_t scratch_doubles, double* xx64, float* xx32, void* stream);
int ams_k_expand_stats(const double* xx64, int32_t Cin, const float* w_exp, int32_t Cexp, double n, const float* center, const float* gamma,
                       const float* beta, float eps, float one_minus_decay, float* moving_mean, float* moving_var, float* scale, float* shift,
                       float* save_mean, float* save_rstd, double* sums, void* stream);

size_t ams_k_xdw_train_scratch(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cexp);
int ams_k_xdw_fwd_stats(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, int32_t Cexp, const float* center,
                        float* scratch, size_t scratch_floats, int32_t* rows_out, int64_t* stride_out, void* stream);
int ams_k_xdw_bwd_reduce(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, int32_t Cexp, const float* sc_e,
                         const float* sh_e, const float* mean_e, const float* rstd_e, int32_t act_e, const float* w_dw, int32_t stride,
                         const float* dz_d, float* scratch, size_t scratch_floats, int32_t* rows_out, int64_t* stride_out, void* stream);
int ams_k_xdw_bwd_dx(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, int32_t Cexp, const float* sc_e,
                     const float* sh_e, int32_t act_e, const float* w_dw, int32_t stride, const float* dz_d, const float* cA, const float* cB,
                     const float* cC, const float* res, float* dx, void* stream);
/* the first block of the network in the same form: the stem conv (3x3 stride 2 over the normalised, 127.5-padded frame, 3 -> 32) is the
 * "expand" layer over its 27-tap patch (k = tap * 3 + channel), followed by the first depthwise conv.  One pass over dz_d [B,H1,W1,32] and
 * the frames: partial rows  S [2][32] | dWd [9][32] | G1 [32][32] (27 rows used) | XX [32][32] | g0 [32]. */
size_t ams_k_xdw_stem_scratch(int32_t B, int32_t fH, int32_t fW);
int ams_k_xdw_bwd_reduce_stem(const void* frames, int32_t frames_dtype, int32_t B, int32_t fH, int32_t fW, float pixel_scale, const float* w_stem,
                              const float* sc_e, const float* sh_e, const float* mean_e, const float* rstd_e, int32_t act_e, const float* w_dw,
                              const float* dz_d, float* scratch, size_t scratch_floats, int32_t* rows_out, int64_t* stride_out, void* stream);
int ams_k_xdw_dwe(const float* G1, const float* xx_g0, int32_t Cin, int32_t Cexp, const float* w_exp, const float* cA, const float* cB,
                  const float* cC, float* dw_exp, void* stream);

/* K14-K16: fused Adam + coordinate-descent mask over a flat arena (TF1 Adam, Appendix C.10). */
int ams_k_adam(float* params, const float* grads, float* m, float* v, const uint8_t* mask, int64_t n, float lr_t,
               float beta1, float beta2, float eps, void* stream);

/* Test hook (no GPU needed): the per-DEVICE bookkeeping behind the one-time kernel attributes (dynamic LDS limits).  Returns 1
 * when (device, kernel_key) has not yet been granted `lds` bytes — i.e. the launcher would call hipFuncSetAttribute now — and
 * records the grant; 0 otherwise.  Two students on two GPUs of one process each get their attributes set. */
int ams_debug_launch_table_needs_attr(int32_t device, uint64_t kernel_key, size_t lds);
/* Tests and tools/ only.  The tuning knobs (environment variables AMS_BLK_TILE, AMS_BLK_HP, AMS_PW_FORCE, AMS_PW_PERCU, AMS_PWX_NO_TAIL, AMS_PWX_FORCE,
 * AMS_PWH_VARIANT, AMS_XDS_FORCE, AMS_XWR_FORCE, AMS_WG6_SPLITS, AMS_WG6_EIGHT_WAVES, AMS_FB_WALK (first block: 0 one tile per block, -2 only the border
 * tiles so, n at most n tiles per walking block): tile / grid / form overrides of single kernels, same results; AMS_BLK_TIMED, AMS_XWR_TIMED: the kernels' phase clocks
 * (ams_debug_phase_cycles; correct results).  AMS_PWH_ABL, AMS_XWR_ABL, AMS_FB_ABL (kernels with loads / MFMAs / stores removed: WRONG results by design; AMS_FB_ABL=32
 * and AMS_XWR_TIMED on the weight-register kernel: clocked forms) exist ONLY in the measurement build libams_hip_measure.so (`make -C ams_amd/csrc measure`, compiled with
 * -DAMS_MEASURE, selected by tools/ through AMS_HIP_LIB): this library has no such kernels, names the variables on stderr and ignores them.  AMS_SIDE_CU_MASK=<hex>: the
 * fine-tune step's side stream confined to the CUs whose bit is set; AMS_EVENT_FLAGS=<hex>: hipEventCreateWithFlags flags of the
 * stream-ordering events — these two change stream semantics and exist for measurements only) are read ONCE, at first use; this re-reads
 * them.  Not thread-safe against launches in flight. */
int ams_debug_reload_knobs(void);
/* tools/ only.  Shader-clock cycles per wave and phase, summed over the launches since the last call, copied to out[0..n) (n <= 8) and cleared.
 * (which = 0 and 2 need the measurement build, see ams_debug_reload_knobs.)
 * which = 0: the walking first block with AMS_FB_ABL=32 ([0] tile decode, [1] stem, [2] wait at the barrier, [3] depthwise + project, [6] wave-tiles);
 * which = 1: the whole-block kernels with AMS_BLK_TIMED=1 ([0] prologue, [1] expand phases, [2] depthwise + project phases, [3] epilogue, [6] waves);
 * which = 2 / 3: the weight-register / the LDS-weight streaming kernel with AMS_XWR_TIMED=1 ([0] E-waves between the step barriers, [1] E-waves at the barrier, [2] / [3]
 * the same for the D-waves, [6] E-wave steps, [7] D-wave steps). */
int ams_debug_phase_cycles(int32_t which, uint64_t* out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* AMS_HIP_H */
