// Internal header of the student engine: the state of one student (ams_student), the per-layer run-time table and the helpers the
// engine's translation units share.  engine_plan.hip lays a student out over its arena, engine_forward.hip holds frozen inference
// and the live (training-mode BN) forward, engine_backward.hip the loss / backward / update half of the fine-tune step, api.hip the C ABI
// of include/ams_hip.h.  Replaces tf.Session.run over the graph built by create_student_v3 (reference utils/graph_utils.py:338-533).
#pragma once
#include <math.h>
#include <stdarg.h>

#include <string>
#include <map>
#include <vector>

#include "kernels.hpp"

namespace ams {

const char* last_error();

// ---- per-launch profiler (HIP events on the launch stream; bench.py's roofline leg) --------------------
struct ProfRec { std::string name; int layer; double bytes, flops, flops_x6; hipEvent_t e0, e1; };
struct Profiler {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void clear() {
        for (auto& r : recs) { pool.push_back(r.e0); pool.push_back(r.e1); }
        recs.clear();
    }
    ~Profiler() {
        clear();
        for (auto e : pool) if (e) (void)hipEventDestroy(e);
    }
};

struct LayerRt {
    ams_layer_desc d;
    int Hin = 0, Win = 0, Hout = 0, Wout = 0;
    int64_t px_in = 0, px_out = 0;         // pixels per image
    // frozen (folded) BN coefficients
    float *fscale = nullptr, *fshift = nullptr;
    // live BN: forward coefficients, saved statistics, backward coefficients
    float *scale = nullptr, *shift = nullptr, *mean = nullptr, *rstd = nullptr, *cA = nullptr, *cB = nullptr, *cC = nullptr;
    double *fsums = nullptr, *bsums = nullptr;      // [2][cout] each, inside the BN_SYNC region
    // training activations (max_batch images each)
    float *z = nullptr, *a = nullptr, *da = nullptr;
    // gradient wrt the raw conv output: written IN PLACE over da, except where da is read again as the gradient over a skip connection (the
    // project layer of a residual block): those layers get a tensor of their own
    float* dzp = nullptr;
    // stride-1 depthwise layer (training): partial rows of its one-kernel backward, kept until the step's batched reduction of the taps
    float* dw_rows = nullptr; size_t dw_rows_floats = 0;
    // frozen weights split into bf16 hi / lo panels [cout][Kp] for the bf16x3 late-layer GEMM (1x1 layers only)
    uint16_t *whi = nullptr, *wlo = nullptr, *wlo3 = nullptr;   // hi, mid (= the 2-part lo), lo of the 3-part split; equally spaced
    uint16_t* whf_mem = nullptr;           // arena home of the fp16 panels; `whf` below is that pointer while the layer's frozen weights fit fp16's
                                           // range (checked by every freeze) and nullptr otherwise: the layer then runs the three-part bf16 form
    uint16_t* whf = nullptr;               // the same panels as two fp16 parts (hi | lo 2^11), plane = cout * Kp apart (AMS_MATMUL_SPLIT_F16)
    int Kp = 0, split_k0 = 0;              // split_k0: first weight row of the panel (concat_projection skips the pool rows)
    float* blk_vecs = nullptr;             // expand layer of a whole-block kernel: [13][cout] table of BN vectors + depthwise taps (freeze)
    double* xx64 = nullptr;                // ... the same sums as doubles (k_xx_stats.hip): the expand layer's BN statistics follow from them
    float* xx_g0 = nullptr;                // expand layer of a recompute block (training): sum x x^T [KP][KP] | sum x [KP] of the last live forward
};

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

}  // namespace ams

struct ams_student {
    ams_student_config cfg;
    std::vector<ams::LayerRt> L;          // 1-based: L[0] unused
    char* arena = nullptr;
    size_t arena_bytes = 0;
    int h = 0, w = 0;                // low-res (output stride 16) size
    int n_backbone = 0;              // index of the last backbone layer
    int iPool = 0, iAspp = 0, iProj = 0, iLogits = 0;
    // regions
    float *params = nullptr, *stats = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    float *fparams = nullptr, *fstats = nullptr;      // frozen snapshot
    double* bn_sync = nullptr; size_t bn_sync_doubles = 0;
    float* logits = nullptr;         // [B,h,w,32]
    uint16_t* xsplit = nullptr; size_t xsplit_plane = 0;         // bf16 parts of a stride-16 block's input (written by the project GEMM before it)
    uint16_t* panel_scratch = nullptr; size_t panel_elems = 0;   // live (training) weights split per launch: hi | lo
    // live weight panels of every 1x1 layer in both orientations (forward [cout][Kp], input gradient [cin][Np]), refreshed by ONE
    // launch at the start of a live forward instead of one small launch per GEMM (47 per step)
    std::vector<ams::SplitJob> tp_jobs;
    std::map<std::pair<const float*, int>, int> tp_index;        // (weight pointer, w_sk == 1) -> job
    ams::SplitJob* tp_jobs_dev = nullptr;
    uint16_t* tp_panels = nullptr; size_t tp_elems = 0;
    int64_t tp_blocks = 0;
    bool tp_fresh = false;                                       // panels hold the split of the CURRENT parameters (this step)
    bool tp_wait = false;                                        // ... once the side stream's split launch (ev_tp) is done: the first GEMM that uses them waits
    hipEvent_t ev_tp = nullptr;
    float* dlogits = nullptr;
    float* ce_scratch = nullptr;       // unnormalised CE gradient planes of the one-pass loss kernel (k_head.hip)
    // fine-tune step of the early blocks without their 6x-expanded tensors (k_xdw_train.hip): AMS_OPT_TRAIN_RECOMPUTE, default on
    int emulate_bf16_storage = 0;      // study only (AMS_OPT_EMULATE_BF16_STORAGE): round d and the block inputs of the stride-16 section to bf16
    int fuse_gemm_red = 3;             // fine-tune step: BN column reductions in the 1x1 GEMM epilogues: bit 0 forward statistics, bit 1 backward sums (AMS_OPT_FUSE_GEMM_RED)
    int fuse_operand_bn = 1;           // fine-tune step: BN + activation of a depthwise layer applied by the project layer's GEMM and weight gradient on their
                                       // operand loads — the depthwise activation is never written (AMS_OPT_FUSE_OPERAND_BN)
    int fuse_dgrad_bn = 3;             // fine-tune step: depthwise input gradient + mask + BN-backward sums of the expand layer in one kernel (AMS_FUSE_DGRAD_BN)
    int train_recompute = 2;           // 1: the expand layer's statistics by recomputing z_e (round 3); 2: from the Gram matrix of the block input (k_xx_stats.hip)
    double* xx_scratch = nullptr; size_t xx_scratch_doubles = 0;       // partial rows of xx_f64_kernel
    float* xt_scratch = nullptr; size_t xt_floats = 0;           // partial rows of those kernels
    float *vec_ones = nullptr, *vec_zeros = nullptr;             // [1024] each: identity BN for a fused kernel's raw output
    float* vec_inv_hw = nullptr;                                  // [1024] x 1 / (h w): d(global mean) / d(feature), the pool branch's backward scale
    float* act[4] = {nullptr, nullptr, nullptr, nullptr};   // inference ping-pong pool
    size_t act_elems = 0;
    float *pooled = nullptr, *pool_a = nullptr, *img_bias = nullptr;          // [B,cin_head], [B,256], [B,256]
    float *d_img_bias = nullptr, *d_pool_a = nullptr, *d_pool_z = nullptr, *d_pooled = nullptr;
    float* im2col = nullptr;         // [B*px1, 32]
    // Backward overlap: the weight gradient of a layer runs on a side stream beside the input gradient / BN backward chain of the
    // main stream (it only feeds the optimizer); the side stream has its own reduction scratch.
    float* scratch2 = nullptr;
    float* scratch3 = nullptr;       // depthwise weight gradients on their own stream (side2), AMS_OVERLAP_WGRAD=2
    hipStream_t side2 = nullptr;
    hipEvent_t ev_xt = nullptr;      // the weight-gradient reductions of a recompute block (side stream) have left xt_scratch
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_head = nullptr;
    // Frozen inference as two half-batches on two streams (forward_frozen_dual).  AMS_OPT_DUAL_STREAM: 0 never, 1 (default) decided per
    // batch size by timing both plans inside the first call with that batch size (>= 16 frames), n >= 2 always from n frames on.
    // Whether it pays is a matter of grid quantisation: at 512x1024 it is +3.5 % at 32-36 frames and -1..-5 % at 24-30 and 40.
    int dual_stream = 1;
    int dual_parts = 2;              // parts when dual_stream >= 2 forces the split (AMS_DUAL_PARTS, 2 .. 4)
    int dual_autotune = 0;           // AMS_OPT_DUAL_AUTOTUNE: time the plans in the first call per batch size (synchronises; opt-in)
    hipStream_t part_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t part_done[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork_dual = nullptr;
    std::map<int, int> dual_choice;  // batch -> number of parts (1 = one stream), filled by the autotune
    int overlap_head = 0;            // frozen inference: image-pooling branch on the side stream beside the aspp0 GEMM (AMS_OVERLAP_HEAD).
                                     // Off: measured 3.63 vs 3.61 ms at 32 frames and 1.90 k vs 2.01 k frames/s at one — the fork / join events
                                     // cost more than the three small launches they hide
    int overlap_wgrad = 1;
    int wgrad_fork_every = 1;        // weight gradients per hand-over to the side stream (AMS_OPT_WGRAD_FORK_EVERY)
    int train_fwd_f16 = 0;           // fine-tune step under AMS_MATMUL_SPLIT_F16: 1 = the FORWARD 1x1 products on two fp16 parts (AMS_OPT_TRAIN_FWD_F16).
                                     // Off: measured 7.455 -> 7.368 ms per 8-frame step (the step's GEMMs are not bound by their MFMAs at 17160 rows), and two
                                     // step-level agreement bars between f32 evaluations (fused vs layer-wise at 34 x 68 x 5 frames, 2 ranks vs 1) move past
                                     // their limits with another product rounding in the forward — not worth 1.2 %
    // create_student_v3's kwargs that run.py leaves off (utils/graph_utils.py:338-339): soft_teacher — the loss's target is softmax(gather(teacher
    // logits)) (:375-376, 403-404), the logits fed per step (ams_student_feed_teacher_logits = feed_dict[teacher_labels_logits_pl]); regularize /
    // train_biases_only — 0.01 * mean of the l2 losses of tvars added to the loss (:451-456; ams_student_set_regularizer)
    int f16_fallback_layers = 0;     // layers whose frozen weights are beyond fp16's range (they run the three-part bf16 form); set by freeze
    int soft_teacher = 0;
    const float* teacher_logits = nullptr; int teacher_th = 0, teacher_tw = 0;
    const uint8_t* reg_mask = nullptr; int reg_nvars = 0; float reg_coef = 0.f;
    int nan_grads = 0;               // a batch without a valid pixel: NaN loss, ZERO gradients — what TensorFlow computes for utils/graph_utils.py:408
                                     // (reduce_mean over the empty boolean_mask: its gradient is an empty tensor, densified to zeros); 1 = NaN gradients (AMS_OPT_NAN_GRADS)
    ~ams_student() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_head) (void)hipEventDestroy(ev_head);
        for (auto& e : part_done) if (e) (void)hipEventDestroy(e);
        for (auto& t : part_stream) if (t) (void)hipStreamDestroy(t);
        if (ev_fork_dual) (void)hipEventDestroy(ev_fork_dual);
        if (side) (void)hipStreamDestroy(side);
        if (side2) (void)hipStreamDestroy(side2);
        if (ev_xt) (void)hipEventDestroy(ev_xt);
        if (ev_tp) (void)hipEventDestroy(ev_tp);
    }
    float* scratch = nullptr; size_t scratch_floats = 0;
    float* tmp_c = nullptr;          // [1024] small per-channel temp
    double* loss_buf = nullptr;      // [2] sum, count (inside BN_SYNC region so DP can all-reduce it)
    int64_t* conf_buf = nullptr;
    int64_t adam_t = 0;
    bool frozen_ready = false;
    int matmul_mode = AMS_MATMUL_SPLIT_F16;       // frozen inference: two fp16 parts (3 MFMAs, f32-level); the fine-tune step: three bf16 parts (6 MFMAs) —
                                                  // every mode but AMS_MATMUL_F32 trains on the three-part form
    int fuse_dw_project = 0;                   // frozen inference: depthwise + project in one kernel on the stride-16 blocks.
                                               // Off by default: measured equal to the two kernels at B = 8 (LDS-read bound:
                                               // 60 b128 reads per wave and 32 channels) and slower at B = 1 (45 blocks)
    int fuse_first_block = 1;                  // frozen inference: stem + depthwise + project of the first block in one kernel:
                                               // 0 three kernels, 1 tile per block (k_first_block.hip), 2 tile per wave (k_block.hip:
                                               // same bits, measured slower here: 488 vs 428 us at 32 frames — the 27-tap byte gather
                                               // per wave outweighs the barriers it saves)
    int64_t stream_min_rows = 4096;            // rows (frames x pixels at the block's resolution) from which the streaming kernels run: two frames of
                                               // 512x1024 and more (measured with the fp16 forms: one frame 0.434 vs 0.431 ms unfused, two 0.501 vs 0.526,
                                               // four 0.675 vs 0.726; same bits either way)
    int fuse_expand_dw_stream = 1;             // frozen inference, split-bf16 modes: expand + depthwise of the stride-16 blocks
                                               // in one streaming kernel (k_xdw_stream.hip): 0 never, 1 where measured
                                               // faster (Cin 64 / 96, >= 16384 rows), 2 also the 160-channel blocks
    int block_x6 = 1;                          // whole-block kernels: expand products of the K = 24 / 32 blocks as six bf16 MFMAs (f32-level)
    int late_subbatch = 0;                     // frozen inference: frames per pass of the output-stride-16 section (0 = the whole batch)
    int fuse_block = 1;                        // frozen inference: a whole early block (Cin <= 32: expand + depthwise + project
                                               // [+ input]) in one kernel, bit-identical to the layer-by-layer plan
    int fuse_expand_dw = 1;                    // frozen inference, expand + depthwise in one kernel: 0 never, 1 where it
                                               // is measured faster (narrow inputs, stride-2 blocks), 2 wherever supported
    ams::Profiler prof;
    hipEvent_t prof_e0 = nullptr;
    double prof_flops = 0.0;         // algorithmic FLOPs of the NEXT profiled launch on the exact-f32 pipe (set right before RUNK)
    double prof_flops_x6 = 0.0;      // ... and those it forms as six bf16 MFMAs on three-part splits
};

namespace ams {

// ---- engine_plan.hip -------------------------------------------------------------------------------------
int student_build(ams_student* s, const ams_student_config* cfg, const ams_layer_desc* layers);
int student_layout(ams_student* s, void* arena, size_t* bytes_out);
void prof_begin(ams_student* s, hipStream_t st);
void prof_end(ams_student* s, hipStream_t st, int layer, double bytes);

// ---- helpers ------------------------------------------------------------------------------------------
static inline PwArgs pw_args(const float* x, int64_t M, int K, int ldx, const float* w, int N, float* y, int ldy) {
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.M = M; a.K = K; a.Kw = K; a.ldx = ldx; a.w = w; a.w_sk = N; a.w_sn = 1; a.N = N;
    a.rows_per_img = 1; a.act = AMS_ACT_NONE; a.y = y; a.ldy = ldy;
    return a;
}

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// launch + profile: LAYER = 1-based layer index (0 = not tied to a layer), BYTES = algorithmic HBM bytes of the launch
#define RUNK(LAYER, BYTES, expr)                                   \
    do {                                                           \
        prof_begin(s, st);                                         \
        int _rc = (expr);                                          \
        prof_end(s, st, (LAYER), (double)(BYTES));                 \
        if (_rc) return _rc;                                       \
    } while (0)

// algorithmic bytes (f32 storage): every operand read once, every result written once
static inline double pw_bytes(const PwArgs& a) {
    // (+ the part planes of the result a project GEMM leaves for the streaming kernel of the next block: 2 bytes per value and part)
    return 4.0 * ((double)a.M * (a.K + a.N + (a.res ? a.N : 0)) + (double)a.Kw * a.N) + (a.ysplit ? 2.0 * a.ysplit_np * (double)a.M * a.N : 0.0);
}
static inline double dw_bytes(const LayerRt& l, int B) { return 4.0 * ((double)B * (l.px_in + l.px_out) * l.d.cin + 9.0 * l.d.cin); }

// cross-rank sums of the data-parallel step: through the library's RCCL communicator on the launch stream (comm), or through a
// host callback (cb: the gloo test hook / any other transport).  `cb` doubles as "a sync is configured" for the callers below.
struct SyncCtx { ams_allreduce_cb cb; void* user; ams_student* s; ams_comm* comm; };

static inline int comm_as_cb(void*, size_t, size_t, int32_t) { return 0; }      // never called: marks SyncCtx::cb when comm is used

static inline int sync_any(const SyncCtx* sc, void* p, size_t n, int dtype, hipStream_t st) {
    if (!sc || !sc->cb) return AMS_OK;
    if (sc->comm) return comm_allreduce(sc->comm, p, n, dtype, st);
    const int rc = sc->cb(sc->user, (size_t)((char*)p - sc->s->arena), n, dtype);
    if (rc) { set_error("all-reduce callback failed (%d)", rc); return AMS_E_STATE; }
    return AMS_OK;
}
static inline int sync_doubles(const SyncCtx* sc, double* p, size_t n, hipStream_t st) { return sync_any(sc, p, n, AMS_DT_F64, st); }

// ---- engine_forward.hip ----------------------------------------------------------------------------------
bool split_pays(const PwArgs& a);
int live_pointwise(ams_student* s, const PwArgs& a, hipStream_t st);
int frozen_pointwise(ams_student* s, int layer, PwArgs a, hipStream_t st, bool* wrote_parts = nullptr, bool force_split = false);
bool train_recompute_block(const ams_student* s, int i);
bool dw_fused_train(const ams_student* s, int i, int B);
bool stem_fused_train(const ams_student* s);
bool dw_fused_train_fwd(const ams_student* s, int i, int B);
bool operand_bn_act(const ams_student* s, int i);      // depthwise layer i: its BN + activation is applied by the project layer's GEMM and weight gradient on their loads of z
size_t red_rows_bound(int64_t M);
int check_call(const ams_student* s, const void* frames, int dtype, int batch);
int run_forward(ams_student* s, const void* frames, int dtype, int batch, int mode, hipStream_t st);
int forward_live(ams_student* s, const void* frames, int dtype, int B, int global_B, bool update_ema, const SyncCtx* sc, hipStream_t st);

// ---- engine_backward.hip ---------------------------------------------------------------------------------
int loss_forward(ams_student* s, const uint8_t* teacher, int B, int32_t* labels, hipStream_t st);
int backward(ams_student* s, const void* frames, int dtype, const uint8_t* teacher, int B, int global_B, const SyncCtx* sc, hipStream_t st);
int train_step_impl(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev, int32_t batch, int32_t global_batch,
                    float lr, const uint8_t* mask_dev, double* loss_dev, ams_allreduce_cb cb, void* user, ams_comm* comm, void* stream);

}  // namespace ams
