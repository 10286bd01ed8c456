// The student engine: owns the launch plan of the AMS hot path over a caller-provided device arena.
//   frozen inference  : stem -> 17 inverted-residual blocks -> head -> fused upsample/argmax(/metrics)
//   live forward      : same graph with training-mode BN (batch statistics), activations kept for backward
//   train step        : live forward -> CE -> backward -> BN moving averages -> Adam (+ coordinate-descent mask)
// Replaces tf.Session.run over the graph built by create_student_v3 (reference utils/graph_utils.py:338-533).
#include <math.h>
#include <stdarg.h>

#include <string>
#include <map>
#include <vector>

#include "kernels.hpp"

namespace ams {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// ---- per-launch profiler (HIP events on the launch stream; bench.py's roofline leg) --------------------
static thread_local const char* g_kname = nullptr;
void note_kernel(const char* name) { if (!g_kname) g_kname = name; }

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
    float* dw_rows = nullptr;
    // frozen weights split into bf16 hi / lo panels [cout][Kp] for the bf16x3 late-layer GEMM (1x1 layers only)
    uint16_t *whi = nullptr, *wlo = nullptr, *wlo3 = nullptr;   // hi, mid (= the 2-part lo), lo of the 3-part split; equally spaced
    int Kp = 0, split_k0 = 0;              // split_k0: first weight row of the panel (concat_projection skips the pool rows)
    float* blk_vecs = nullptr;             // expand layer of a whole-block kernel: [13][cout] table of BN vectors + depthwise taps (freeze)
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

using namespace ams;

struct ams_student {
    ams_student_config cfg;
    std::vector<LayerRt> L;          // 1-based: L[0] unused
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
    std::vector<SplitJob> tp_jobs;
    std::map<std::pair<const float*, int>, int> tp_index;        // (weight pointer, w_sk == 1) -> job
    SplitJob* tp_jobs_dev = nullptr;
    uint16_t* tp_panels = nullptr; size_t tp_elems = 0;
    int64_t tp_blocks = 0;
    bool tp_fresh = false;                                       // panels hold the split of the CURRENT parameters (this step)
    float* dlogits = nullptr;
    float* ce_scratch = nullptr;       // unnormalised CE gradient planes of the one-pass loss kernel (k_head.hip)
    // fine-tune step of the early blocks without their 6x-expanded tensors (k_xdw_train.hip): AMS_OPT_TRAIN_RECOMPUTE, default on
    int emulate_bf16_storage = 0;      // study only (AMS_OPT_EMULATE_BF16_STORAGE): round d and the block inputs of the stride-16 section to bf16
    int fuse_gemm_red = 3;             // fine-tune step: BN column reductions in the 1x1 GEMM epilogues: bit 0 forward statistics, bit 1 backward sums (AMS_OPT_FUSE_GEMM_RED)
    int fuse_dgrad_bn = 2;             // fine-tune step: depthwise input gradient + mask + BN-backward sums of the expand layer in one kernel (AMS_FUSE_DGRAD_BN)
    int train_recompute = 1;
    float* xt_scratch = nullptr; size_t xt_floats = 0;           // partial rows of those kernels
    float *vec_ones = nullptr, *vec_zeros = nullptr;             // [1024] each: identity BN for a fused kernel's raw output
    float* act[4] = {nullptr, nullptr, nullptr, nullptr};   // inference ping-pong pool
    size_t act_elems = 0;
    float *pooled = nullptr, *pool_a = nullptr, *img_bias = nullptr;          // [B,cin_head], [B,256], [B,256]
    float *d_img_bias = nullptr, *d_pool_a = nullptr, *d_pool_z = nullptr, *d_pooled = nullptr;
    float* im2col = nullptr;         // [B*px1, 32]
    float* dz = nullptr;             // head: gradient wrt a raw conv output, max layer size (the backbone writes dz in place, LayerRt::dzp)
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
    ~ams_student() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_head) (void)hipEventDestroy(ev_head);
        for (auto& e : part_done) if (e) (void)hipEventDestroy(e);
        for (auto& t : part_stream) if (t) (void)hipStreamDestroy(t);
        if (ev_fork_dual) (void)hipEventDestroy(ev_fork_dual);
        if (side) (void)hipStreamDestroy(side);
        if (side2) (void)hipStreamDestroy(side2);
        if (ev_xt) (void)hipEventDestroy(ev_xt);
    }
    float* scratch = nullptr; size_t scratch_floats = 0;
    float* tmp_c = nullptr;          // [1024] small per-channel temp
    double* loss_buf = nullptr;      // [2] sum, count (inside BN_SYNC region so DP can all-reduce it)
    int64_t* conf_buf = nullptr;
    int64_t adam_t = 0;
    bool frozen_ready = false;
    int matmul_mode = AMS_MATMUL_SPLIT_BF16_X6;   // late layers: three-part bf16 split (f32-level products) in inference and training
    int fuse_dw_project = 0;                   // frozen inference: depthwise + project in one kernel on the stride-16 blocks.
                                               // Off by default: measured equal to the two kernels at B = 8 (LDS-read bound:
                                               // 60 b128 reads per wave and 32 channels) and slower at B = 1 (45 blocks)
    int fuse_first_block = 1;                  // frozen inference: stem + depthwise + project of the first block in one kernel:
                                               // 0 three kernels, 1 tile per block (k_first_block.hip), 2 tile per wave (k_block.hip:
                                               // same bits, measured slower here: 488 vs 428 us at 32 frames — the 27-tap byte gather
                                               // per wave outweighs the barriers it saves)
    int64_t stream_min_rows = 16384;           // rows (frames x pixels at the block's resolution) from which the streaming kernels run
    int fuse_expand_dw_stream = 1;             // frozen inference, split-bf16 modes: expand + depthwise of the stride-16 blocks
                                               // in one streaming kernel (k_xdw_stream.hip): 0 never, 1 where measured
                                               // faster (Cin 64 / 96, >= 16384 rows), 2 also the 160-channel blocks
    int block_x6 = 1;                          // whole-block kernels: expand products of the K = 24 / 32 blocks as six bf16 MFMAs (f32-level)
    int late_subbatch = 0;                     // frozen inference: frames per pass of the output-stride-16 section (0 = the whole batch)
    int fuse_block = 1;                        // frozen inference: a whole early block (Cin <= 32: expand + depthwise + project
                                               // [+ input]) in one kernel, bit-identical to the layer-by-layer plan
    int fuse_expand_dw = 1;                    // frozen inference, expand + depthwise in one kernel: 0 never, 1 where it
                                               // is measured faster (narrow inputs, stride-2 blocks), 2 wherever supported
    Profiler prof;
    hipEvent_t prof_e0 = nullptr;
    double prof_flops = 0.0;         // algorithmic FLOPs of the NEXT profiled launch on the exact-f32 pipe (set right before RUNK)
    double prof_flops_x6 = 0.0;      // ... and those it forms as six bf16 MFMAs on three-part splits
};

namespace ams {

// the live weight panels: every 1x1 layer a split GEMM may run on, forward and input-gradient orientation.  p0 holds the element
// offset inside tp_panels until the arena is known (create turns it into a pointer).
static void plan_train_panels(ams_student* s) {
    s->tp_jobs.clear();
    s->tp_elems = 0;
    s->tp_blocks = 0;
    if (!s->cfg.trainable) return;
    auto add = [&](int64_t w_off, int64_t sk, int64_t sn, int K, int N) {
        if (K < 32 || K % 8 != 0) return;                  // split_pays() never takes these
        SplitJob j;
        memset(&j, 0, sizeof(j));
        j.w = (const float*)(uintptr_t)w_off;              // offset for now
        j.sk = sk; j.sn = sn; j.K = K; j.N = N; j.Kp = (K + 31) / 32 * 32;
        j.plane = (int64_t)N * j.Kp;
        j.p0 = (uint16_t*)(uintptr_t)s->tp_elems;
        j.first_block = s->tp_blocks;
        s->tp_elems += 3 * (size_t)j.plane;
        s->tp_elems = (s->tp_elems + 127) & ~(size_t)127;
        s->tp_blocks += (j.plane + 255) / 256;
        s->tp_jobs.push_back(j);
    };
    for (int i = 2; i <= s->cfg.n_layers; ++i) {
        const LayerRt& l = s->L[i];
        if (l.d.role == AMS_ROLE_DEPTHWISE || l.d.role == AMS_ROLE_POOL_CONV) continue;
        int64_t w_off = l.d.w_off;
        int K = l.d.cin;
        const int N = l.d.cout;
        if (l.d.role == AMS_ROLE_CONCAT_PROJ) { const int k0 = s->L[s->iPool].d.cout; w_off += (int64_t)k0 * N; K -= k0; }
        add(w_off, N, 1, K, N);                             // forward: element (k, n) at w[k*N + n]
        if (l.d.role != AMS_ROLE_LOGITS) add(w_off, 1, N, N, K);      // input gradient: B operand (k' = n, n' = k) = w[k][n]
    }
}

static int layout(ams_student* s, void* arena, size_t* bytes_out) {
    const ams_student_config& c = s->cfg;
    Carver cv(arena);
    const int64_t nT = c.n_trainable, nS = c.n_stats;
    const int B = c.max_batch;
    s->params = cv.take<float>(nT);
    s->stats = cv.take<float>(nS);
    s->fparams = cv.take<float>(nT);
    s->fstats = cv.take<float>(nS);
    if (c.trainable) {
        s->grads = cv.take<float>(nT);
        s->adam_m = cv.take<float>(nT);
        s->adam_v = cv.take<float>(nT);
    }
    size_t sum_c = 0, max_elems = 0, max_c = 0;
    for (int i = 1; i <= c.n_layers; ++i) {
        LayerRt& l = s->L[i];
        sum_c += l.d.cout;
        const size_t e = (size_t)l.px_out * l.d.cout;
        if (e > max_elems) max_elems = e;
        if ((size_t)l.d.cout > max_c) max_c = l.d.cout;
        if ((size_t)l.d.cin > max_c) max_c = l.d.cin;
    }
    for (int i = 1; i <= c.n_layers; ++i) {
        LayerRt& l = s->L[i];
        l.fscale = cv.take<float>(l.d.cout);
        l.fshift = cv.take<float>(l.d.cout);
        l.scale = cv.take<float>(l.d.cout);
        l.shift = cv.take<float>(l.d.cout);
        l.mean = cv.take<float>(l.d.cout);
        l.rstd = cv.take<float>(l.d.cout);
        l.cA = cv.take<float>(l.d.cout);
        l.cB = cv.take<float>(l.d.cout);
        l.cC = cv.take<float>(l.d.cout);
    }
    for (int i = 2; i <= c.n_layers; ++i) {
        LayerRt& l = s->L[i];
        const int role = l.d.role;
        if (role == AMS_ROLE_DEPTHWISE || role == AMS_ROLE_POOL_CONV) continue;
        int K = l.d.cin;
        l.split_k0 = 0;
        if (role == AMS_ROLE_CONCAT_PROJ) { l.split_k0 = s->L[s->iPool].d.cout; K = l.d.cin - l.split_k0; }
        l.Kp = (K + 31) / 32 * 32;
        {
            const size_t plane = (size_t)l.d.cout * l.Kp;
            l.whi = cv.take<uint16_t>(3 * plane);          // one allocation: the planes must be equally spaced
            l.wlo = l.whi ? l.whi + plane : nullptr;
            l.wlo3 = l.whi ? l.whi + 2 * plane : nullptr;
        }
    }
    if (s->n_backbone >= 3 && s->L[1].d.cout == 32 && s->L[2].d.role == AMS_ROLE_DEPTHWISE) {
        LayerRt& l1 = s->L[1];
        l1.blk_vecs = cv.take<float>(13 * 32);
        l1.Kp = 32;                                        // stem weights as three bf16 parts [32][32] for the first-block kernel's split form
        l1.whi = cv.take<uint16_t>(3 * 32 * 32);
        l1.wlo = l1.whi ? l1.whi + 32 * 32 : nullptr;
        l1.wlo3 = l1.whi ? l1.whi + 2 * 32 * 32 : nullptr;
    }
    for (int i = 2; i + 2 <= s->n_backbone; ++i) {      // whole-block kernels: packed per-channel tables, filled by freeze
        LayerRt& l = s->L[i];
        if (l.d.role == AMS_ROLE_EXPAND && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE && s->L[i + 2].d.role == AMS_ROLE_PROJECT &&
            block_fused_supported(l.d.cin, l.d.cout, s->L[i + 2].d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate, s->L[i + 2].d.residual_from != 0))
            l.blk_vecs = cv.take<float>((size_t)13 * l.d.cout);
    }
    // BN sync region: loss (2 doubles) then per layer fwd sums [2][C], bwd sums [2][C]
    s->bn_sync_doubles = 2 + 4 * sum_c;
    s->bn_sync = cv.take<double>(s->bn_sync_doubles);
    s->loss_buf = s->bn_sync;
    {
        double* p = s->bn_sync ? s->bn_sync + 2 : nullptr;
        for (int i = 1; i <= c.n_layers; ++i) {
            LayerRt& l = s->L[i];
            l.fsums = p; if (p) p += 2 * l.d.cout;
            l.bsums = p; if (p) p += 2 * l.d.cout;
        }
    }
    s->conf_buf = cv.take<int64_t>(32 * 32);
    s->logits = cv.take<float>((size_t)B * s->h * s->w * 32);
    const int head_cin = s->L[s->iPool].d.cin, aspp_c = s->L[s->iPool].d.cout;
    s->pooled = cv.take<float>((size_t)B * head_cin);
    s->pool_a = cv.take<float>((size_t)B * aspp_c);
    s->img_bias = cv.take<float>((size_t)B * aspp_c);
    s->tmp_c = cv.take<float>(4096);
    s->act_elems = (size_t)B * max_elems;
    {
        size_t pl = 0;
        for (int i = 2; i + 1 <= s->n_backbone; ++i) {
            const LayerRt& l = s->L[i];
            if (l.d.role == AMS_ROLE_EXPAND && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE && l.d.cin >= 64 &&      /* split-bf16 forms only */
                expand_dw_stream_supported(l.d.cin, l.d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate) && (size_t)B * l.px_in * l.d.cin > pl)
                pl = (size_t)B * l.px_in * l.d.cin;
        }
        s->xsplit_plane = pl;
        s->xsplit = pl ? cv.take<uint16_t>(3 * pl) : nullptr;
    }
    for (int k = 0; k < 4; ++k) s->act[k] = cv.take<float>(s->act_elems);
    // scratch: column-reduction partials, wgrad splits, depthwise wgrad partials
    size_t sc = colstats_scratch(0, (int)max_c) + 1024;
    if (image_colsum_scratch(B, (int)max_c) > sc) sc = image_colsum_scratch(B, (int)max_c);
    if (c.trainable) {
        for (int i = 1; i <= c.n_layers; ++i) {
            const LayerRt& l = s->L[i];
            size_t need;
            const int64_t M = (int64_t)B * l.px_out;
            if (l.d.role == AMS_ROLE_DEPTHWISE) {
                need = depthwise_wgrad_scratch(B, l.Hin, l.Win, l.d.cin, l.d.stride, l.d.rate);
                if (l.d.stride == 1 && l.d.cin <= 1024) {       // the one-kernel forms of the blocks that keep their tensors
                    const size_t n1 = depthwise_dgrad_bn_scratch(B, l.Hin, l.Win, l.d.cin), n2 = depthwise_fwd_bn_scratch(B, l.Hin, l.Win, l.d.cin, l.d.rate);
                    if (n1 > need) need = n1;
                    if (n2 > need) need = n2;
                }
            } else if (l.d.role == AMS_ROLE_STEM) need = pointwise_wgrad_scratch(M, 27, l.d.cout);
            else need = pointwise_wgrad_scratch(M, l.d.cin, l.d.cout);
            if (need > sc) sc = need;
        }
    }
    s->scratch_floats = sc;
    s->scratch = cv.take<float>(sc);
    if (c.trainable) {
        // one panel buffer for the live split-bf16 GEMMs (forward and dgrad orientation): the stream orders split -> GEMM
        size_t pe = 0;
        for (int i = 2; i <= c.n_layers; ++i) {
            const LayerRt& l = s->L[i];
            if (l.d.role == AMS_ROLE_DEPTHWISE) continue;
            const size_t f = (size_t)l.d.cout * ((l.d.cin + 31) / 32 * 32), b = (size_t)l.d.cin * ((l.d.cout + 31) / 32 * 32);
            if (3 * f > pe) pe = 3 * f;
            if (3 * b > pe) pe = 3 * b;
        }
        s->panel_elems = pe;
        s->panel_scratch = cv.take<uint16_t>(pe);
        plan_train_panels(s);
        s->tp_panels = cv.take<uint16_t>(s->tp_elems);
        s->tp_jobs_dev = cv.take<SplitJob>(s->tp_jobs.size());
    }
    if (c.trainable) {
        s->dlogits = cv.take<float>((size_t)B * s->h * s->w * 32);
        s->ce_scratch = cv.take<float>(ce_loss_grad_scratch(B, s->h, s->w, c.n_selected));
        size_t xt = 0;
        for (int i = 2; i + 1 <= s->n_backbone; ++i) {
            const LayerRt& l = s->L[i];
            if (l.d.role == AMS_ROLE_EXPAND && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE &&
                xdw_train_supported(l.d.cin, l.d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate)) {
                const size_t need = xdw_train_scratch(B, l.Hin, l.Win, l.d.cin, l.d.cout);
                if (need > xt) xt = need;
            }
        }
        if (s->n_backbone >= 3 && s->L[1].d.cout == 32 && s->L[2].d.role == AMS_ROLE_DEPTHWISE && s->L[2].d.stride == 1 && s->L[2].d.rate == 1 &&
            xdw_stem_scratch(B, c.height, c.width) > xt)
            xt = xdw_stem_scratch(B, c.height, c.width);
        s->xt_floats = xt;
        s->xt_scratch = cv.take<float>(xt);
        for (int i = 2; i + 1 <= s->n_backbone; ++i) {
            LayerRt& l = s->L[i];
            if (l.d.role == AMS_ROLE_EXPAND && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE &&
                xdw_train_supported(l.d.cin, l.d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate)) {
                const int KP = (l.d.cin + 15) / 16 * 16;
                l.xx_g0 = cv.take<float>((size_t)KP * KP + KP);
            }
        }
        s->vec_ones = cv.take<float>(1024);
        s->vec_zeros = cv.take<float>(1024);
        s->d_img_bias = cv.take<float>((size_t)B * aspp_c);
        s->d_pool_a = cv.take<float>((size_t)B * aspp_c);
        s->d_pool_z = cv.take<float>((size_t)B * aspp_c);
        s->d_pooled = cv.take<float>((size_t)B * head_cin);
        s->im2col = cv.take<float>((size_t)B * s->L[1].px_out * 32);
        s->dz = cv.take<float>((size_t)B * max_elems);
        s->scratch2 = cv.take<float>(sc);
        s->scratch3 = cv.take<float>(sc);
        for (int i = 1; i <= c.n_layers; ++i) {
            LayerRt& l = s->L[i];
            if (l.d.role == AMS_ROLE_LOGITS) continue;     // logits live in s->logits / s->dlogits
            const size_t e = (size_t)B * l.px_out * l.d.cout;
            l.z = cv.take<float>(e);
            l.a = cv.take<float>(e);
            l.da = cv.take<float>(e);
            if (l.d.residual_from) l.dzp = cv.take<float>(e);
            if (l.d.role == AMS_ROLE_DEPTHWISE && l.d.stride == 1 && i >= 3 && l.d.cin <= 1024)
                l.dw_rows = cv.take<float>(depthwise_dgrad_bn_scratch(B, l.Hin, l.Win, l.d.cin));
        }
    }
    *bytes_out = (cv.off + 255) & ~(size_t)255;
    return AMS_OK;
}

static int build(ams_student* s, const ams_student_config* cfg, const ams_layer_desc* layers) {
    AMS_REQUIRE(cfg && layers, "student: null config");
    AMS_REQUIRE(cfg->abi_version == AMS_ABI_VERSION, "student: ABI version %d, library is %d", cfg->abi_version, AMS_ABI_VERSION);
    AMS_REQUIRE(cfg->height > 0 && cfg->width > 0 && cfg->max_batch > 0, "student: bad frame size / batch");
    AMS_REQUIRE(cfg->n_selected > 0 && cfg->n_selected <= 32 && cfg->num_classes <= 32, "student: class counts out of range");
    AMS_REQUIRE(cfg->n_layers >= 8 && cfg->n_layers < 512, "student: bad layer count");
    AMS_REQUIRE(cfg->act_dtype == AMS_DT_F32, "student: only f32 activation storage is implemented in this build");
    s->cfg = *cfg;
    s->L.assign(cfg->n_layers + 1, LayerRt());
    int H = cfg->height + 1, W = cfg->width + 1;       // the graph pads one row / column of 127.5 first
    for (int i = 1; i <= cfg->n_layers; ++i) {
        LayerRt& l = s->L[i];
        l.d = layers[i - 1];
        const int role = l.d.role;
        if (role == AMS_ROLE_STEM) {
            AMS_REQUIRE(i == 1 && l.d.cin == 3 && l.d.stride == 2, "student: layer 1 must be the 3x3/2 stem");
            l.Hin = H; l.Win = W;
            int p;
            same_pad(H, 3, 2, 1, &l.Hout, &p);
            same_pad(W, 3, 2, 1, &l.Wout, &p);
            H = l.Hout; W = l.Wout;
        } else if (role == AMS_ROLE_DEPTHWISE) {
            AMS_REQUIRE(l.d.cin == l.d.cout, "student: depthwise layer %d must keep the channel count", i);
            l.Hin = H; l.Win = W;
            int p;
            same_pad(H, 3, l.d.stride, l.d.rate, &l.Hout, &p);
            same_pad(W, 3, l.d.stride, l.d.rate, &l.Wout, &p);
            H = l.Hout; W = l.Wout;
        } else if (role == AMS_ROLE_POOL_CONV) {
            l.Hin = l.Win = l.Hout = l.Wout = 1;
            s->iPool = i;
        } else {
            AMS_REQUIRE(l.d.stride == 1, "student: 1x1 layer %d must have stride 1", i);
            l.Hin = l.Hout = H; l.Win = l.Wout = W;
            if (role == AMS_ROLE_ASPP) s->iAspp = i;
            if (role == AMS_ROLE_CONCAT_PROJ) s->iProj = i;
            if (role == AMS_ROLE_LOGITS) s->iLogits = i;
        }
        l.px_in = (int64_t)l.Hin * l.Win;
        l.px_out = (int64_t)l.Hout * l.Wout;
        if (role <= AMS_ROLE_PROJECT) s->n_backbone = i;
        if (l.d.residual_from) AMS_REQUIRE(l.d.residual_from < i && role == AMS_ROLE_PROJECT, "student: bad residual on layer %d", i);
        AMS_REQUIRE(l.d.cout % 4 == 0 || role == AMS_ROLE_LOGITS, "student: layer %d cout=%d not a multiple of 4", i, l.d.cout);
    }
    AMS_REQUIRE(s->iPool && s->iAspp && s->iProj && s->iLogits == cfg->n_layers, "student: head layers missing");
    AMS_REQUIRE(s->iPool == s->n_backbone + 1 && s->iAspp == s->iPool + 1 && s->iProj == s->iAspp + 1,
                "student: head must be image_pooling, aspp0, concat_projection, logits");
    AMS_REQUIRE(s->L[s->iProj].d.cin == s->L[s->iPool].d.cout + s->L[s->iAspp].d.cout, "student: concat width mismatch");
    AMS_REQUIRE(s->L[s->iLogits].d.cout == cfg->num_classes, "student: logits width != num_classes");
    s->h = H; s->w = W;
    return AMS_OK;
}

// ---- helpers ------------------------------------------------------------------------------------------
static PwArgs pw_args(const float* x, int64_t M, int K, int ldx, const float* w, int N, float* y, int ldy) {
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.M = M; a.K = K; a.Kw = K; a.ldx = ldx; a.w = w; a.w_sk = N; a.w_sn = 1; a.N = N;
    a.rows_per_img = 1; a.act = AMS_ACT_NONE; a.y = y; a.ldy = ldy;
    return a;
}

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

static inline void prof_begin(ams_student* s, hipStream_t st) {
    g_kname = nullptr;
    if (!s->prof.on) return;
    s->prof_e0 = s->prof.get();
    (void)hipEventRecord(s->prof_e0, st);
}
static inline void prof_end(ams_student* s, hipStream_t st, int layer, double bytes) {
    if (!s->prof.on) return;
    hipEvent_t e1 = s->prof.get();
    (void)hipEventRecord(e1, st);
    s->prof.recs.push_back(ProfRec{g_kname ? g_kname : "?", layer, bytes, s->prof_flops, s->prof_flops_x6, s->prof_e0, e1});
    s->prof_flops = 0.0;
    s->prof_flops_x6 = 0.0;
}
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
    return 4.0 * ((double)a.M * (a.K + a.N + (a.res ? a.N : 0)) + (double)a.Kw * a.N);
}
static inline double dw_bytes(const LayerRt& l, int B) { return 4.0 * ((double)B * (l.px_in + l.px_out) * l.d.cin + 9.0 * l.d.cin); }

// cross-rank sums of the data-parallel step: through the library's RCCL communicator on the launch stream (comm), or through a
// host callback (cb: the gloo test hook / any other transport).  `cb` doubles as "a sync is configured" for the callers below.
struct SyncCtx { ams_allreduce_cb cb; void* user; ams_student* s; ams_comm* comm; };

static int comm_as_cb(void*, size_t, size_t, int32_t) { return 0; }      // never called: marks SyncCtx::cb when comm is used

static int sync_any(const SyncCtx* sc, void* p, size_t n, int dtype, hipStream_t st) {
    if (!sc || !sc->cb) return AMS_OK;
    if (sc->comm) return comm_allreduce(sc->comm, p, n, dtype, st);
    const int rc = sc->cb(sc->user, (size_t)((char*)p - sc->s->arena), n, dtype);
    if (rc) { set_error("all-reduce callback failed (%d)", rc); return AMS_E_STATE; }
    return AMS_OK;
}
static int sync_doubles(const SyncCtx* sc, double* p, size_t n, hipStream_t st) { return sync_any(sc, p, n, AMS_DT_F64, st); }

// Split-bf16 pays where the exact-f32 kernels are matrix-pipe bound (f32-input MFMA = 157 TFLOP/s against ~5 TB/s of HBM:
// ~31 FLOP per byte): few rows (the streaming kernel needs >= 32768), a weight panel too large for the streaming kernel,
// or an arithmetic intensity 2KN / 4(K+N) of 20 FLOP/B and more (64 -> 384 and wider, at any batch size).
static bool split_pays(const PwArgs& a) {
    if (a.K < 32 || a.K % 8 != 0 || a.M < 256) return false;
    return a.M < 32768 || !pointwise_stream_applies(a) || (int64_t)a.K * a.N >= 40 * (int64_t)(a.K + a.N);
}

// live (training) 1x1 layer or its input gradient: same split-bf16 rule as the frozen path, the weights are split right
// before the launch because they change every step (one small kernel; the panels live in one shared scratch buffer)
static int live_pointwise(ams_student* s, const PwArgs& a, hipStream_t st) {
    const bool split = s->matmul_mode != AMS_MATMUL_F32 && s->panel_scratch && split_pays(a) && a.Kw == a.K && a.ldx % 4 == 0;
    if (!split) return launch_pointwise(a, st);
    // three-part split (6 MFMAs, f32-level products): gradients amplify product error ~1e5 x on this graph, the two-part
    // split of the frozen path would put the step outside the f32 error class
    const int Kp = (a.K + 31) / 32 * 32;
    const size_t plane = (size_t)a.N * Kp;
    if (s->tp_fresh) {                                  // split once per step (forward_live) instead of once per launch
        auto it = s->tp_index.find({a.w, a.w_sk == 1 ? 1 : 0});
        if (it != s->tp_index.end()) {
            const SplitJob& j = s->tp_jobs[it->second];
            if (j.K == a.K && j.N == a.N && j.sk == a.w_sk && j.sn == a.w_sn)
                return launch_pointwise_split3(a, j.p0, j.p0 + j.plane, j.p0 + 2 * j.plane, j.Kp, st);
        }
    }
    AMS_REQUIRE(3 * plane <= s->panel_elems, "live_pointwise: panel scratch too small");
    uint16_t* p0 = s->panel_scratch;
    int rc = launch_split_weights3(a.w, a.w_sk, a.w_sn, a.K, a.N, Kp, p0, p0 + plane, p0 + 2 * plane, st);
    if (rc) return rc;
    return launch_pointwise_split3(a, p0, p0 + plane, p0 + 2 * plane, Kp, st);
}

// frozen 1x1 layer: late layers (few rows, wide K/N: matrix-pipe bound) go through the split-bf16 kernel
static int frozen_pointwise(ams_student* s, int layer, PwArgs a, hipStream_t st, bool* wrote_parts = nullptr, bool force_split = false) {
    const LayerRt& l = s->L[layer];
    const bool split = s->matmul_mode != AMS_MATMUL_F32 && l.whi && (split_pays(a) || (force_split && a.K % 8 == 0 && a.K >= 32));
    // the bf16 parts of the result (a.ysplit) exist only when the split kernel runs with a vector epilogue
    const bool parts = split && a.ysplit && pointwise_split_writes_parts(a);
    if (!parts) a.ysplit = nullptr;
    if (wrote_parts) *wrote_parts = parts;
    if (split && s->matmul_mode == AMS_MATMUL_SPLIT_BF16_X6) RUNK(layer, pw_bytes(a), launch_pointwise_split3(a, l.whi, l.wlo, l.wlo3, l.Kp, st));
    else if (split && s->matmul_mode == AMS_MATMUL_BF16) RUNK(layer, pw_bytes(a), launch_pointwise_split1(a, l.whi, l.Kp, st));
    else if (split) RUNK(layer, pw_bytes(a), launch_pointwise_split(a, l.whi, l.wlo, l.Kp, st));
    else RUNK(layer, pw_bytes(a), launch_pointwise(a, st));
    return AMS_OK;
}

// =======================================================================================================
// frozen inference (BN folded; what the edge device runs)
// =======================================================================================================
static int forward_frozen(ams_student* s, const void* frames, int dtype, const int Bfull, hipStream_t st) {
    const ams_student_config& c = s->cfg;
    int B = Bfull;                             // frames of the current pass: the whole batch, or one sub-batch of the late section
    const float* P = s->fparams;
    float* cur = s->act[0];
    int cur_i = 0;
    int i = 2;
    {
        LayerRt& l = s->L[1];
        LayerRt& ld = s->L[2];
        LayerRt& lj = s->L[3];
        const double in_bytes = (double)B * c.height * c.width * 3 * (dtype == AMS_DT_U8 ? 1 : 4);
        if (s->fuse_first_block && s->n_backbone >= 3 && l.d.cout == 32 && ld.d.role == AMS_ROLE_DEPTHWISE && ld.d.cin == 32 &&
            ld.d.stride == 1 && ld.d.rate == 1 && lj.d.role == AMS_ROLE_PROJECT && lj.d.cin == 32 && lj.d.cout == 16 &&
            !lj.d.residual_from) {
            // stem + depthwise + project of the first block in one kernel: the 32-channel half-resolution tensor stays in LDS
            const double bytes = in_bytes + 4.0 * B * lj.px_out * lj.d.cout + 4.0 * (27 * 32 + 9 * 32 + 32 * 16);
            if (s->fuse_first_block >= 2)
                RUNK(3, bytes, launch_first_block_tiles(frames, dtype, B, c.height, c.width, c.pixel_scale, P + l.d.w_off, l.fscale, l.fshift,
                                                        l.d.act, P + ld.d.w_off, ld.fscale, ld.fshift, ld.d.act, P + lj.d.w_off, lj.fscale,
                                                        lj.fshift, lj.d.act, cur, st, l.blk_vecs));
            else {
                const bool x6 = s->block_x6 && s->matmul_mode != AMS_MATMUL_F32 && l.whi;
                RUNK(3, bytes, launch_first_block(frames, dtype, B, c.height, c.width, c.pixel_scale, P + l.d.w_off, l.fscale, l.fshift,
                                                  l.d.act, P + ld.d.w_off, ld.fscale, ld.fshift, ld.d.act, P + lj.d.w_off, lj.fscale,
                                                  lj.fshift, lj.d.act, cur, st, x6 ? l.whi : nullptr, 32 * 32));
            }
            i = 4;
        } else {
            const double bytes = in_bytes + 4.0 * B * l.px_out * l.d.cout;
            RUNK(1, bytes, launch_stem(frames, dtype, B, c.height, c.width, P + l.d.w_off, l.d.cout, l.fscale, l.fshift, l.d.act,
                                       c.pixel_scale, cur, st));
        }
    }
    int reserved = -1;                         // buffer that holds the late section's input for ALL sub-batches: never a target there
    auto other = [&](int avoid0, int avoid1) { for (int k = 0; k < 4; ++k) if (k != avoid0 && k != avoid1 && k != reserved) return k; return -1; };
    // layer k starts a block whose expand + depthwise run as the streaming kernel (stride-16 blocks, split-bf16 modes, a few
    // frames: below that the launch cannot fill the chip)
    auto stream_ok = [&](int k) {
        if (!(s->fuse_expand_dw_stream && k + 1 <= s->n_backbone && s->L[k].d.role == AMS_ROLE_EXPAND && s->L[k + 1].d.role == AMS_ROLE_DEPTHWISE &&
              (int64_t)B * s->L[k].px_in >= s->stream_min_rows && (int64_t)s->L[k + 1].px_out * s->L[k + 1].d.cout * 4 < 0x7fffffffLL &&
              expand_dw_stream_supported(s->L[k].d.cin, s->L[k].d.cout, s->L[k + 1].d.stride, s->L[k + 1].d.rate)))
            return false;
        // stride 2 on the streaming kernel is correct and tested but measured no faster than the tiled kernel (the expand runs at
        // full resolution either way: 412 vs 376 us on the first such block) — only with option value 2
        if (s->L[k + 1].d.stride != 1 && s->fuse_expand_dw_stream < 2) return false;
        if (s->L[k].d.cin <= 32) return true;                        // exact-f32 form: any matmul mode
        return s->matmul_mode != AMS_MATMUL_F32 && s->L[k].whi && s->L[k].Kp == s->L[k].d.cin;
    };
    const uint16_t* cur_parts = nullptr;       // `cur` as bf16 parts (s->xsplit), when the GEMM that produced it wrote them
    // The output-stride-16 section (blocks 7-16 and the head) can run in sub-batches: its largest tensor, the depthwise result of
    // the 960-channel blocks, is 264 MB at 32 frames — written by one kernel, read by the next, and larger than the 256 MB Infinity
    // Cache.  At 16 frames the writer/reader pairs of that section meet in the cache (and every sub-batch reuses the same addresses).
    int i_late = s->n_backbone + 1;
    for (int k = 2; k <= s->n_backbone; ++k)
        if (s->L[k].d.role == AMS_ROLE_EXPAND && s->L[k].px_in == (int64_t)s->h * s->w) { i_late = k; break; }
    const int sub = (s->late_subbatch > 0 && Bfull > s->late_subbatch && i_late <= s->n_backbone) ? s->late_subbatch : Bfull;
    auto run_blocks = [&](int i_stop) -> int {
    while (i <= s->n_backbone && i < i_stop) {
        // one inverted-residual block: [expand] -> depthwise -> project (+ block input)
        const float* block_in = cur;
        const float* x = cur;
        int x_i = cur_i;
        if (s->fuse_block && i + 2 <= s->n_backbone && s->L[i].d.role == AMS_ROLE_EXPAND && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE &&
            s->L[i + 2].d.role == AMS_ROLE_PROJECT && (!s->L[i + 2].d.residual_from || s->L[i + 2].d.residual_from == i - 1) &&
            block_fused_supported(s->L[i].d.cin, s->L[i].d.cout, s->L[i + 2].d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate,
                                  s->L[i + 2].d.residual_from != 0)) {
            // early blocks: only the block input and output touch HBM (k_block.hip)
            LayerRt& le = s->L[i];
            LayerRt& ld = s->L[i + 1];
            LayerRt& lj = s->L[i + 2];
            const int o = other(cur_i, -1);
            const bool res = lj.d.residual_from != 0;
            const double bytes = 4.0 * ((double)B * (le.px_in * le.d.cin * (res ? 2 : 1) + lj.px_out * lj.d.cout) + (double)le.d.cin * le.d.cout +
                                        9.0 * ld.d.cin + (double)lj.d.cin * lj.d.cout);
            // algorithmic FLOPs (no halo, no padding): the kernel is bound by the exact-f32 matrix pipe, not by HBM
            // three-part split products for the expand layer when K >= 24 (not in the exact-f32 mode; the one- and two-part modes
            // concern the late layers only: the early blocks keep f32-level products there too)
            const bool x6 = s->block_x6 && s->matmul_mode != AMS_MATMUL_F32 && le.whi && le.Kp == 32 && le.d.cin > 16;
            const double fl_e = 2.0 * B * (double)le.px_in * le.d.cin * le.d.cout;
            s->prof_flops = 2.0 * B * ((double)ld.px_out * 9.0 * ld.d.cin + (double)lj.px_out * lj.d.cin * lj.d.cout) + (x6 ? 0.0 : fl_e);
            s->prof_flops_x6 = x6 ? fl_e : 0.0;
            RUNK(i + 2, bytes, launch_block_fused(cur, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, le.fscale, le.fshift, le.d.act, le.d.cout,
                                                  P + ld.d.w_off, ld.d.stride, ld.fscale, ld.fshift, ld.d.act, P + lj.d.w_off, lj.fscale, lj.fshift,
                                                  lj.d.act, lj.d.cout, res, s->act[o], st, le.blk_vecs, x6 ? le.whi : nullptr,
                                                  (int64_t)(le.wlo - le.whi)));
            cur = s->act[o]; cur_i = o; i += 3;
            cur_parts = nullptr;
            continue;
        }
        const bool stream_here = stream_ok(i) && (s->L[i].d.cin <= 96 || cur_parts || s->fuse_expand_dw_stream >= 2);
        if (!stream_here && s->fuse_expand_dw && s->L[i].d.role == AMS_ROLE_EXPAND && i + 1 <= s->n_backbone &&
            s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE &&
            expand_dw_supported(s->L[i].d.cin, s->L[i].d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate) &&
            (s->fuse_expand_dw >= 2 || s->L[i].d.cin <= 24 || s->L[i + 1].d.stride == 2)) {
            // expand + depthwise in one kernel: the 6x-expanded tensor stays in LDS
            LayerRt& le = s->L[i];
            LayerRt& ld = s->L[i + 1];
            const int o = other(cur_i, -1);
            const double bytes = 4.0 * ((double)B * (le.px_in * le.d.cin + ld.px_out * ld.d.cout) + (double)le.d.cin * le.d.cout + 9.0 * ld.d.cin);
            RUNK(i + 1, bytes, launch_expand_dw(x, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, le.fscale, le.fshift, le.d.act, le.d.cout,
                                                P + ld.d.w_off, ld.d.stride, ld.d.rate, ld.fscale, ld.fshift, ld.d.act, s->act[o], st));
            x = s->act[o]; x_i = o; i += 2;
        } else if (stream_here) {
            // stride-16 blocks: expand + depthwise streamed through an LDS ring, split-bf16 products (bit-identical to the two
            // kernels it replaces); the 6x-expanded tensor is never written
            LayerRt& le = s->L[i];
            LayerRt& ld = s->L[i + 1];
            const int o = other(cur_i, -1);
            const int np = s->matmul_mode == AMS_MATMUL_SPLIT_BF16_X6 ? 3 : s->matmul_mode == AMS_MATMUL_BF16 ? 1 : 2;
            const double bytes = 4.0 * ((double)B * (le.px_in * le.d.cin + ld.px_out * ld.d.cout) + (double)le.d.cin * le.d.cout + 9.0 * ld.d.cin);
            const int64_t xplane = (int64_t)B * le.px_in * le.d.cin;
            if (le.d.cin > 96 && cur_parts)
                // 160 -> 960: expand weights in registers, the operand staged once per block in LDS (k_xdw_wreg.hip); with 30 channel
                // chunks the LDS-weight form is bound by its passes over the operand
                RUNK(i + 1, bytes, launch_expand_dw_wreg(cur_parts, xplane, B, le.Hin, le.Win, le.d.cin, le.whi, (int64_t)(le.wlo - le.whi), np, le.fscale,
                                                         le.fshift, le.d.act, le.d.cout, P + ld.d.w_off, ld.d.rate, ld.fscale, ld.fshift, ld.d.act,
                                                         s->act[o], st));
            else
                RUNK(i + 1, bytes, launch_expand_dw_stream(x, cur_parts, xplane, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, le.whi, (int64_t)(le.wlo - le.whi), np, le.fscale,
                                                           le.fshift, le.d.act, le.d.cout, P + ld.d.w_off, ld.d.stride, ld.d.rate, ld.fscale, ld.fshift, ld.d.act,
                                                           s->act[o], st));
            x = s->act[o]; x_i = o; i += 2;
            if (s->emulate_bf16_storage && ld.px_out == (int64_t)s->h * s->w)
                RUN(launch_round_bf16(s->act[o], (int64_t)B * ld.px_out * ld.d.cout, st));          // d as bf16 storage would hold it
        } else {
        if (s->L[i].d.role == AMS_ROLE_EXPAND) {
            LayerRt& l = s->L[i];
            const int o = other(cur_i, -1);
            PwArgs a = pw_args(x, (int64_t)B * l.px_in, l.d.cin, l.d.cin, P + l.d.w_off, l.d.cout, s->act[o], l.d.cout);
            a.scale = l.fscale; a.shift = l.fshift; a.act = l.d.act;
            // an expand layer the streaming kernel can take forms its products the same way when it runs alone (split bf16), so
            // that the result does not depend on batch size or on AMS_OPT_FUSE_EXPAND_DW_STREAM
            const bool streamable = i + 1 <= s->n_backbone && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE && l.Kp == l.d.cin && l.d.cin >= 64 &&
                                    expand_dw_stream_supported(l.d.cin, l.d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate);
            RUN(frozen_pointwise(s, i, a, st, nullptr, streamable));
            x = s->act[o]; x_i = o; ++i;
        }
        {
            LayerRt& l = s->L[i];
            AMS_REQUIRE(l.d.role == AMS_ROLE_DEPTHWISE, "engine: expected depthwise at layer %d", i);
            LayerRt& lpj = s->L[i + 1];
            if (s->fuse_dw_project && s->matmul_mode == AMS_MATMUL_SPLIT_BF16 &&   /* two-part split only */ i + 1 <= s->n_backbone &&
                lpj.d.role == AMS_ROLE_PROJECT && lpj.whi && (int64_t)B * l.px_out < 32768 && (int64_t)B * l.px_out >= 256 &&
                lpj.Kp == l.d.cin && dw_project_supported(l.d.cin, lpj.d.cout, l.d.stride, l.d.rate)) {
                // depthwise + project in one kernel (split-bf16 GEMM that computes its own operand): d never reaches HBM
                const int o = other(cur_i, x_i);
                PwArgs a = pw_args(nullptr, (int64_t)B * lpj.px_in, lpj.d.cin, lpj.d.cin, P + lpj.d.w_off, lpj.d.cout, s->act[o], lpj.d.cout);
                a.scale = lpj.fscale; a.shift = lpj.fshift; a.act = lpj.d.act;
                if (lpj.d.residual_from) { a.res = block_in; a.ldr = lpj.d.cout; }
                const double bytes = 4.0 * ((double)B * (l.px_in * l.d.cin + lpj.px_out * lpj.d.cout * (a.res ? 2 : 1)) +
                                            (double)lpj.d.cin * lpj.d.cout + 9.0 * l.d.cin);
                RUNK(i + 1, bytes, launch_dw_project(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, l.fscale, l.fshift, l.d.act, a,
                                                     lpj.whi, lpj.wlo, lpj.Kp, st));
                cur = s->act[o]; cur_i = o; i += 2;
                cur_parts = nullptr;
                continue;
            }
            const int o = other(cur_i, x_i);
            RUNK(i, dw_bytes(l, B), launch_depthwise(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.stride, l.d.rate, l.fscale,
                                                     l.fshift, l.d.act, s->act[o], st));
            if (s->emulate_bf16_storage && l.px_out == (int64_t)s->h * s->w)
                RUN(launch_round_bf16(s->act[o], (int64_t)B * l.px_out * l.d.cout, st));
            x = s->act[o]; x_i = o; ++i;
        }
        }
        {
            LayerRt& l = s->L[i];
            AMS_REQUIRE(l.d.role == AMS_ROLE_PROJECT, "engine: expected project at layer %d", i);
            const int o = other(cur_i, x_i);
            PwArgs a = pw_args(x, (int64_t)B * l.px_in, l.d.cin, l.d.cin, P + l.d.w_off, l.d.cout, s->act[o], l.d.cout);
            a.scale = l.fscale; a.shift = l.fshift; a.act = l.d.act;
            if (l.d.residual_from) { a.res = block_in; a.ldr = l.d.cout; }
            bool wrote = false;
            if (stream_ok(i + 1) && s->xsplit && (size_t)a.M * a.N <= s->xsplit_plane) {
                // the next block streams: its expand GEMM takes this result as bf16 parts, written here once instead of being
                // split by every channel-chunk block there
                a.ysplit = s->xsplit; a.ysplit_plane = a.M * a.N; a.ysplit_np = s->matmul_mode == AMS_MATMUL_SPLIT_BF16_X6 ? 3 : s->matmul_mode == AMS_MATMUL_BF16 ? 1 : 2;
            }
            if (s->emulate_bf16_storage && l.px_out == (int64_t)s->h * s->w) a.ysplit = nullptr;      // the parts would be those of the unrounded result
            RUN(frozen_pointwise(s, i, a, st, &wrote));
            cur_parts = wrote ? s->xsplit : nullptr;
            cur = s->act[o]; cur_i = o; ++i;
            if (s->emulate_bf16_storage && l.px_out == (int64_t)s->h * s->w)
                RUN(launch_round_bf16(s->act[o], (int64_t)B * l.px_out * l.d.cout, st));            // block input as bf16 storage would hold it
        }
    }
    return AMS_OK;
    };
    // ---- head -------------------------------------------------------------------------------------------
    LayerRt& lp = s->L[s->iPool]; LayerRt& la = s->L[s->iAspp]; LayerRt& lc = s->L[s->iProj]; LayerRt& ll = s->L[s->iLogits];
    const int64_t HW = (int64_t)s->h * s->w;
    auto run_head = [&](int B0) -> int {       // frames B0 .. B0 + B - 1 of the batch
        const int64_t M = (int64_t)B * HW;
        float* pooled = s->pooled + (int64_t)B0 * lp.d.cin;
        float* pool_a = s->pool_a + (int64_t)B0 * lp.d.cout;
        float* img_bias = s->img_bias + (int64_t)B0 * lc.d.cout;
        // The image-pooling branch (global mean -> 1x1 + BN + ReLU -> its share of concat_projection as a per-image bias) is three
        // latency-bound launches on a handful of rows (58 us at 32 frames, 22 us at one).  With overlap_head it runs on the side stream
        // beside the aspp0 GEMM and joins before concat_projection (off by default, see the flag).
        const bool fork = s->overlap_head && !s->prof.on;
        hipStream_t ps = st;
        if (fork) {
            if (!s->side) AMS_CHECK_HIP(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
            if (!s->ev_fork) AMS_CHECK_HIP(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
            if (!s->ev_head) AMS_CHECK_HIP(hipEventCreateWithFlags(&s->ev_head, hipEventDisableTiming));
            AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
            AMS_CHECK_HIP(hipStreamWaitEvent(s->side, s->ev_fork, 0));
            ps = s->side;
        }
        RUNK(s->iPool, 4.0 * M * lp.d.cin, launch_global_mean(cur, B, HW, lp.d.cin, pooled, s->scratch, ps));
        {   // image_pooling conv + BN + ReLU on the pooled vector
            PwArgs a = pw_args(pooled, B, lp.d.cin, lp.d.cin, P + lp.d.w_off, lp.d.cout, pool_a, lp.d.cout);
            a.scale = lp.fscale; a.shift = lp.fshift; a.act = lp.d.act;
            RUNK(s->iPool, pw_bytes(a), launch_pointwise(a, ps));
            // the broadcast pool branch enters concat_projection as a per-image bias: W_proj[0:256]^T . pool
            PwArgs b = pw_args(pool_a, B, lp.d.cout, lp.d.cout, P + lc.d.w_off, lc.d.cout, img_bias, lc.d.cout);
            RUNK(s->iProj, pw_bytes(b), launch_pointwise(b, ps));
        }
        if (fork) AMS_CHECK_HIP(hipEventRecord(s->ev_head, s->side));
        const int o1 = other(cur_i, -1), o2 = other(cur_i, o1);
        PwArgs a = pw_args(cur, M, la.d.cin, la.d.cin, P + la.d.w_off, la.d.cout, s->act[o1], la.d.cout);
        a.scale = la.fscale; a.shift = la.fshift; a.act = la.d.act;
        RUN(frozen_pointwise(s, s->iAspp, a, st));
        if (fork) AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_head, 0));
        PwArgs b = pw_args(s->act[o1], M, la.d.cout, la.d.cout, P + lc.d.w_off + (int64_t)lp.d.cout * lc.d.cout, lc.d.cout,
                           s->act[o2], lc.d.cout);
        b.img_bias = img_bias; b.rows_per_img = HW; b.scale = lc.fscale; b.shift = lc.fshift; b.act = lc.d.act;
        RUN(frozen_pointwise(s, s->iProj, b, st));
        PwArgs d = pw_args(s->act[o2], M, lc.d.cout, lc.d.cout, P + ll.d.w_off, ll.d.cout, s->logits + (int64_t)B0 * HW * 32, 32);
        d.shift = P + ll.d.gamma_off;      // biases
        RUN(frozen_pointwise(s, s->iLogits, d, st));
        return AMS_OK;
    };
    if (sub >= Bfull) {
        RUN(run_blocks(s->n_backbone + 1));
        return run_head(0);
    }
    RUN(run_blocks(i_late));                   // early section: the whole batch
    float* late_in = cur;
    const int late_in_i = cur_i;
    const int64_t late_in_frame = s->L[i_late].px_in * s->L[i_late].d.cin;
    reserved = late_in_i;
    for (int B0 = 0; B0 < Bfull; B0 += sub) {
        B = Bfull - B0 < sub ? Bfull - B0 : sub;
        cur = late_in + (int64_t)B0 * late_in_frame;
        cur_i = late_in_i;
        cur_parts = nullptr;
        i = i_late;
        RUN(run_blocks(s->n_backbone + 1));
        RUN(run_head(B0));
    }
    return AMS_OK;
}


// =======================================================================================================
// live forward: training-mode BN.  z = raw conv output, batch statistics -> (scale, shift), a = act(z*scale+shift)(+res)
// =======================================================================================================
// layer i opens an early block whose fine-tune step runs without the expanded tensors (k_xdw_train.hip)
static bool train_recompute_block(const ams_student* s, int i) {
    if (!s->train_recompute || !s->xt_scratch || i < 2 || i + 1 > s->n_backbone) return false;
    const LayerRt& l = s->L[i];
    const LayerRt& ld = s->L[i + 1];
    return l.xx_g0 && l.d.role == AMS_ROLE_EXPAND && ld.d.role == AMS_ROLE_DEPTHWISE && xdw_train_supported(l.d.cin, l.d.cout, ld.d.stride, ld.d.rate) &&
           expand_dw_supported(l.d.cin, l.d.cout, ld.d.stride, ld.d.rate) && l.d.cout <= 1024 &&
           xdw_train_scratch(s->cfg.max_batch, l.Hin, l.Win, l.d.cin, l.d.cout) <= s->xt_floats;
}

// Stride-1 depthwise layer i of a block that keeps its tensors: its backward is ONE kernel that recomputes the expand layer's activation
// from z_e (backward(), k_conv.hip dw3x3_dgrad_bn_kernel) — so nothing in backward reads a_e, and at fuse_dgrad_bn >= 2 the forward
// does not write it either (dw3x3_fwd_bn_kernel applies the expand layer's BN + activation on its tap loads).
static bool dw_fused_train(const ams_student* s, int i, int B) {
    if (i < 3 || i > s->n_backbone || !s->fuse_dgrad_bn || train_recompute_block(s, i - 1)) return false;
    const LayerRt& l = s->L[i];
    const LayerRt& prev = s->L[i - 1];
    return l.d.role == AMS_ROLE_DEPTHWISE && l.d.stride == 1 && prev.d.role == AMS_ROLE_EXPAND && prev.d.cout == l.d.cin && l.d.cin <= 1024 &&
           depthwise_dgrad_bn_scratch(B, l.Hin, l.Win, l.d.cin) <= s->scratch_floats;
}
// First block: the stem is the "expand" layer of depthwise layer 2, and the backward of the pair is one pass over dz and the frames that
// never reads the stem's activation either (backward(), launch_xdw_bwd_reduce_stem).
static bool stem_fused_train(const ams_student* s) {
    const ams_student_config& c = s->cfg;
    return s->n_backbone >= 2 && s->train_recompute && s->L[1].d.role == AMS_ROLE_STEM && s->L[1].d.cout == 32 && s->L[2].d.role == AMS_ROLE_DEPTHWISE &&
           s->L[2].d.stride == 1 && s->L[2].d.rate == 1 && s->xt_scratch && xdw_stem_scratch(c.max_batch, c.height, c.width) <= s->xt_floats;
}
static bool dw_fused_train_fwd(const ams_student* s, int i, int B) {
    if (s->fuse_dgrad_bn < 2 || i < 2 || i > s->n_backbone) return false;
    if (!(i == 2 ? stem_fused_train(s) : dw_fused_train(s, i, B))) return false;
    return depthwise_fwd_bn_scratch(B, s->L[i].Hin, s->L[i].Win, s->L[i].d.cin, s->L[i].d.rate) <= s->scratch_floats;
}

// most partial rows a GEMM with a fused column reduction can leave behind (PwArgs::red_mode): one per block of the persistent streaming
// kernel (<= 8 per CU), one per 64-row strip of the tiled split kernel
static size_t red_rows_bound(int64_t M) { return M >= 32768 ? 2048 : (size_t)(M / 64 + 8); }

// pre_rows > 0: the kernel that wrote l.z already left the statistics' partial rows [pre_rows][2][C] in s->scratch
static int bn_train(ams_student* s, LayerRt& l, int64_t M_local, double n_global, bool update_ema, const SyncCtx* sc,
                    const float* res, hipStream_t st, int pre_rows = 0, bool act_pass = true) {
    const ams_student_config& c = s->cfg;
    const float* center = s->stats + l.d.mean_off;       // shifted sums: moving_mean is a good, rank-identical centre
    const float omd = 1.0f - c.bn_decay;
    float* mm = update_ema ? s->stats + l.d.mean_off : nullptr;
    float* mv = update_ema ? s->stats + l.d.var_off : nullptr;
    if (pre_rows > 0) {
        if (!sc || !sc->cb) {
            RUN(launch_bn_fwd_finalize_partials(s->scratch, pre_rows, 2 * (int64_t)l.d.cout, l.d.cout, l.fsums, n_global, center,
                                                s->params + l.d.gamma_off, s->params + l.d.beta_off, l.d.bn_eps, omd, mm, mv, l.scale, l.shift,
                                                l.mean, l.rstd, st));
        } else {
            RUN(launch_partials_to_sums(s->scratch, pre_rows, 2 * (int64_t)l.d.cout, l.d.cout, l.fsums, st));
            RUN(sync_doubles(sc, l.fsums, 2 * (size_t)l.d.cout, st));
            RUN(launch_bn_finalize(l.fsums, n_global, l.d.cout, center, s->params + l.d.gamma_off, s->params + l.d.beta_off, l.d.bn_eps, omd,
                                   mm, mv, l.scale, l.shift, l.mean, l.rstd, st));
        }
    } else if (!sc || !sc->cb) {
        // no cross-rank sum between the statistics and their use: the reduction's second stage finishes the BN arithmetic
        RUNK(0, 4.0 * M_local * l.d.cout,
             launch_colstats_bn(l.z, M_local, l.d.cout, center, l.fsums, s->scratch, n_global, s->params + l.d.gamma_off,
                                s->params + l.d.beta_off, l.d.bn_eps, omd, mm, mv, l.scale, l.shift, l.mean, l.rstd, st));
    } else {
        RUNK(0, 4.0 * M_local * l.d.cout, launch_colstats(l.z, M_local, l.d.cout, center, l.fsums, s->scratch, st));
        RUN(sync_doubles(sc, l.fsums, 2 * (size_t)l.d.cout, st));
        RUN(launch_bn_finalize(l.fsums, n_global, l.d.cout, center, s->params + l.d.gamma_off, s->params + l.d.beta_off, l.d.bn_eps, omd,
                               mm, mv, l.scale, l.shift, l.mean, l.rstd, st));
    }
    if (!act_pass) return AMS_OK;              // the consumer applies scale / shift / activation on its own loads of z
    RUNK(0, 4.0 * M_local * l.d.cout * (res ? 3 : 2), launch_bn_act(l.z, M_local, l.d.cout, l.scale, l.shift, l.d.act, res, l.a, st));
    return AMS_OK;
}

static int forward_live(ams_student* s, const void* frames, int dtype, int B, int global_B, bool update_ema, const SyncCtx* sc,
                        hipStream_t st) {
    const ams_student_config& c = s->cfg;
    AMS_REQUIRE(c.trainable, "live forward needs a trainable student (activations are not allocated)");
    const float* P = s->params;
    // the parameters may have changed since the last call (Adam, restore): all live weight panels in one launch
    s->tp_fresh = false;
    if (s->matmul_mode != AMS_MATMUL_F32 && !s->tp_jobs.empty()) {
        RUNK(0, 0.0, launch_split_batch(s->tp_jobs_dev, (int)s->tp_jobs.size(), s->tp_blocks, st));
        s->tp_fresh = true;
    }
    {
        LayerRt& l = s->L[1];
        RUN(launch_stem(frames, dtype, B, c.height, c.width, P + l.d.w_off, l.d.cout, nullptr, nullptr, AMS_ACT_NONE,
                        c.pixel_scale, l.z, st));
        RUN(bn_train(s, l, (int64_t)B * l.px_out, (double)global_B * l.px_out, update_ema, sc, nullptr, st, 0, !dw_fused_train_fwd(s, 2, B)));
    }
    for (int i = 2; i <= s->n_backbone; ++i) {
        LayerRt& l = s->L[i];
        const float* x = s->L[i - 1].a;
        if (train_recompute_block(s, i)) {
            // early block: neither z_e nor a_e is written.  Statistics of z_e = x . W_e straight from x, then the inference kernel
            // expand + BN + ReLU6 + depthwise with the batch statistics -> the depthwise layer's raw output
            LayerRt& ld = s->L[i + 1];
            const float* center = s->stats + l.d.mean_off;
            const float omd = 1.0f - c.bn_decay;
            float* mm = update_ema ? s->stats + l.d.mean_off : nullptr;
            float* mv = update_ema ? s->stats + l.d.var_off : nullptr;
            const double n_e = (double)global_B * l.px_out;
            int rows = 0;
            int64_t fstride = 0;
            RUNK(i, 4.0 * B * l.px_in * l.d.cin,
                 launch_xdw_fwd_stats(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.cout, center, s->xt_scratch, &rows, &fstride, st));
            {   // sums of x and x x^T over this rank's pixels, kept for the expand weight gradient
                const int KP = (l.d.cin + 15) / 16 * 16;
                RUN(launch_reduce_splits(s->xt_scratch + 2 * (int64_t)l.d.cout, rows, (int64_t)KP * KP + KP, l.xx_g0, st, fstride));
            }
            if (!sc || !sc->cb) {
                RUN(launch_bn_fwd_finalize_partials(s->xt_scratch, rows, fstride, l.d.cout, l.fsums, n_e, center,
                                                    s->params + l.d.gamma_off, s->params + l.d.beta_off, l.d.bn_eps, omd, mm, mv, l.scale,
                                                    l.shift, l.mean, l.rstd, st));
            } else {
                RUN(launch_partials_to_sums(s->xt_scratch, rows, fstride, l.d.cout, l.fsums, st));
                RUN(sync_doubles(sc, l.fsums, 2 * (size_t)l.d.cout, st));
                RUN(launch_bn_finalize(l.fsums, n_e, l.d.cout, center, s->params + l.d.gamma_off, s->params + l.d.beta_off, l.d.bn_eps, omd,
                                       mm, mv, l.scale, l.shift, l.mean, l.rstd, st));
            }
            RUNK(i + 1, 4.0 * ((double)B * (l.px_in * l.d.cin + ld.px_out * ld.d.cout)),
                 launch_expand_dw(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.scale, l.shift, l.d.act, l.d.cout, P + ld.d.w_off, ld.d.stride,
                                  ld.d.rate, s->vec_ones, s->vec_zeros, AMS_ACT_NONE, ld.z, st));
            RUN(bn_train(s, ld, (int64_t)B * ld.px_out, (double)global_B * ld.px_out, update_ema, sc, nullptr, st));
            ++i;
            continue;
        }
        int pre_rows = 0;
        if (l.d.role == AMS_ROLE_DEPTHWISE && dw_fused_train_fwd(s, i, B)) {
            // BN + activation of the expand layer on the tap loads (its `a` was not written), the statistics of the result on the way out
            const LayerRt& le = s->L[i - 1];
            RUNK(i, dw_bytes(l, B), launch_depthwise_fwd_bn(le.z, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, le.scale, le.shift, le.d.act,
                                                            s->stats + l.d.mean_off, l.z, s->scratch, &pre_rows, st));
        } else if (l.d.role == AMS_ROLE_DEPTHWISE) {
            RUNK(i, dw_bytes(l, B), launch_depthwise(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.stride, l.d.rate, nullptr, nullptr,
                                                     AMS_ACT_NONE, l.z, st));
        } else {
            PwArgs a = pw_args(x, (int64_t)B * l.px_in, l.d.cin, l.d.cin, P + l.d.w_off, l.d.cout, l.z, l.d.cout);
            // the BN statistics of the result in this GEMM's epilogue, where the kernel chosen can do it
            if ((s->fuse_gemm_red & 1) && red_rows_bound(a.M) * 2 * (size_t)l.d.cout <= s->scratch_floats) {
                a.red_mode = 1; a.red_center = s->stats + l.d.mean_off; a.red_part = s->scratch; a.red_rows_out = &pre_rows;
            }
            RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
        }
        const float* res = l.d.residual_from ? s->L[l.d.residual_from].a : nullptr;
        const bool act_pass = !(l.d.role == AMS_ROLE_EXPAND && dw_fused_train_fwd(s, i + 1, B));
        RUN(bn_train(s, l, (int64_t)B * l.px_out, (double)global_B * l.px_out, update_ema, sc, res, st, pre_rows, act_pass));
    }
    LayerRt& lp = s->L[s->iPool]; LayerRt& la = s->L[s->iAspp]; LayerRt& lc = s->L[s->iProj]; LayerRt& ll = s->L[s->iLogits];
    const float* feat = s->L[s->n_backbone].a;
    const int64_t HW = (int64_t)s->h * s->w, M = (int64_t)B * HW;
    RUN(launch_global_mean(feat, B, HW, lp.d.cin, s->pooled, s->scratch, st));
    {
        PwArgs a = pw_args(s->pooled, B, lp.d.cin, lp.d.cin, P + lp.d.w_off, lp.d.cout, lp.z, lp.d.cout);
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
        RUN(bn_train(s, lp, B, (double)global_B, update_ema, sc, nullptr, st));     // statistics over the batch only
        PwArgs b = pw_args(lp.a, B, lp.d.cout, lp.d.cout, P + lc.d.w_off, lc.d.cout, s->img_bias, lc.d.cout);
        RUNK(0, pw_bytes(b), live_pointwise(s, b, st));
    }
    {
        PwArgs a = pw_args(feat, M, la.d.cin, la.d.cin, P + la.d.w_off, la.d.cout, la.z, la.d.cout);
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
        RUN(bn_train(s, la, M, (double)global_B * HW, update_ema, sc, nullptr, st));
        PwArgs b = pw_args(la.a, M, la.d.cout, la.d.cout, P + lc.d.w_off + (int64_t)lp.d.cout * lc.d.cout, lc.d.cout, lc.z, lc.d.cout);
        b.img_bias = s->img_bias; b.rows_per_img = HW;
        RUNK(0, pw_bytes(b), live_pointwise(s, b, st));
        RUN(bn_train(s, lc, M, (double)global_B * HW, update_ema, sc, nullptr, st));
        PwArgs d = pw_args(lc.a, M, lc.d.cout, lc.d.cout, P + ll.d.w_off, ll.d.cout, s->logits, 32);
        d.shift = P + ll.d.gamma_off;
        RUNK(0, pw_bytes(d), live_pointwise(s, d, st));
    }
    return AMS_OK;
}

// =======================================================================================================
// backward + update
// =======================================================================================================
// BN backward of layer l given da (gradient wrt the layer's activated output): writes dz into s->dz, dgamma/dbeta into grads
static int bn_backward(ams_student* s, LayerRt& l, const float* da, int64_t M_local, double n_global, const SyncCtx* sc,
                       hipStream_t st, float* dz = nullptr) {
    if (!dz) dz = s->dz;
    if (!sc || !sc->cb) {
        RUNK(0, 8.0 * M_local * l.d.cout,
             launch_bn_bwd_reduce_coef(da, l.z, M_local, l.d.cout, l.scale, l.shift, l.d.act, l.mean, l.rstd, l.bsums, s->scratch,
                                       n_global, s->params + l.d.gamma_off, l.cA, l.cB, l.cC, s->grads + l.d.gamma_off,
                                       s->grads + l.d.beta_off, st));
    } else {
        RUNK(0, 8.0 * M_local * l.d.cout,
             launch_bn_bwd_reduce(da, l.z, M_local, l.d.cout, l.scale, l.shift, l.d.act, l.mean, l.rstd, l.bsums, s->scratch, st));
        // gamma/beta gradients from this rank's own sums: the gradient all-reduce adds the ranks up exactly once
        RUN(launch_bn_param_grads(l.bsums, l.d.cout, s->grads + l.d.gamma_off, s->grads + l.d.beta_off, st));
        RUN(sync_doubles(sc, l.bsums, 2 * (size_t)l.d.cout, st));
        RUN(launch_bn_bwd_coef(l.bsums, n_global, l.d.cout, s->params + l.d.gamma_off, l.mean, l.rstd, l.cA, l.cB, l.cC,
                               nullptr, nullptr, st));
    }
    RUNK(0, 12.0 * M_local * l.d.cout,
         launch_bn_bwd_apply(da, l.z, M_local, l.d.cout, l.scale, l.shift, l.d.act, l.cA, l.cB, l.cC, dz, st));
    return AMS_OK;
}

static int pw_wgrad(ams_student* s, const float* x, int ldx, int K, const float* dy, int ldy, int N, int64_t M, float* dw,
                    hipStream_t st, float* scratch = nullptr) {
    WgArgs a;
    a.x = x; a.ldx = ldx; a.K = K; a.dy = dy; a.ldy = ldy; a.N = N; a.M = M; a.dw = dw;
    a.scratch = scratch ? scratch : s->scratch; a.scratch_floats = s->scratch_floats;
    a.allow_split = s->matmul_mode != AMS_MATMUL_F32;
    RUNK(0, 4.0 * ((double)M * (K + N) + (double)K * N), launch_pointwise_wgrad(a, st));
    return AMS_OK;
}

// dx[M,K] = dy[M,N] @ w[K,N]^T (+ extras through the epilogue)
static PwArgs dgrad_args(const float* dy, int64_t M, int N, int ldy, const float* w, int K, float* dx) {
    PwArgs a = pw_args(dy, M, N, ldy, w, K, dx, K);
    a.w_sk = 1; a.w_sn = N;          // B operand (k' = n, n' = k) = w[k][n]
    return a;
}

static int backward(ams_student* s, const void* frames, int dtype, const uint8_t* teacher, int B, int global_B, const SyncCtx* sc,
                    hipStream_t st) {
    const ams_student_config& c = s->cfg;
    const float* P = s->params;
    float* G = s->grads;
    LayerRt& lp = s->L[s->iPool]; LayerRt& la = s->L[s->iAspp]; LayerRt& lc = s->L[s->iProj]; LayerRt& ll = s->L[s->iLogits];
    const int64_t HW = (int64_t)s->h * s->w, M = (int64_t)B * HW;
    const double nHW = (double)global_B * HW;
    const int NC = c.num_classes;
    // d loss / d logits (already divided by the global number of valid pixels)
    if (ce_loss_grad_supported(s->w, c.width))           // second pass of the one-pass loss kernel (train_step_impl ran the first)
        RUNK(0, 0.0, launch_ce_combine(B, s->h, s->w, c.class_indices, c.n_selected, NC, s->loss_buf, s->ce_scratch, s->dlogits, 32, st));
    else
        RUNK(0, 0.0, launch_ce_grad(s->logits, 32, B, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher, NC, s->loss_buf,
                                    s->dlogits, 32, st));
    // logits layer: bias, weights, input gradient
    RUN(launch_colsum(s->dlogits, M, 32, 32, s->tmp_c, s->scratch, st));
    RUN(launch_copy(G + ll.d.gamma_off, s->tmp_c, NC, st));
    RUN(pw_wgrad(s, lc.a, lc.d.cout, lc.d.cout, s->dlogits, 32, NC, M, G + ll.d.w_off, st));
    {
        PwArgs a = dgrad_args(s->dlogits, M, 32, 32, P + ll.d.w_off, ll.d.cin, lc.da);
        a.Kw = NC; a.w_sn = NC;        // w is [cin][NC]; dlogits columns >= NC are zero
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // concat_projection
    RUN(bn_backward(s, lc, lc.da, M, nHW, sc, st));
    const float* Wc_top = P + lc.d.w_off;
    const float* Wc_bot = P + lc.d.w_off + (int64_t)lp.d.cout * lc.d.cout;
    RUN(pw_wgrad(s, la.a, la.d.cout, la.d.cout, s->dz, lc.d.cout, lc.d.cout, M, G + lc.d.w_off + (int64_t)lp.d.cout * lc.d.cout, st));
    {
        PwArgs a = dgrad_args(s->dz, M, lc.d.cout, lc.d.cout, Wc_bot, la.d.cout, la.da);
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // pool branch: the per-image bias collects the column sums of dz_proj
    RUN(launch_image_colsum(s->dz, B, HW, lc.d.cout, lc.d.cout, s->d_img_bias, s->scratch, st));
    RUN(pw_wgrad(s, lp.a, lp.d.cout, lp.d.cout, s->d_img_bias, lc.d.cout, lc.d.cout, B, G + lc.d.w_off, st));
    {
        PwArgs a = dgrad_args(s->d_img_bias, B, lc.d.cout, lc.d.cout, Wc_top, lp.d.cout, s->d_pool_a);
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    {   // BN (over the batch) + ReLU of the pool branch; its dz goes to d_pool_z instead of s->dz (still in use? no: consumed)
        RUN(launch_bn_bwd_reduce(s->d_pool_a, lp.z, B, lp.d.cout, lp.scale, lp.shift, lp.d.act, lp.mean, lp.rstd, lp.bsums, s->scratch, st));
        RUN(launch_bn_param_grads(lp.bsums, lp.d.cout, G + lp.d.gamma_off, G + lp.d.beta_off, st));
        RUN(sync_doubles(sc, lp.bsums, 2 * (size_t)lp.d.cout, st));
        RUN(launch_bn_bwd_coef(lp.bsums, (double)global_B, lp.d.cout, P + lp.d.gamma_off, lp.mean, lp.rstd, lp.cA, lp.cB, lp.cC,
                               nullptr, nullptr, st));
        RUN(launch_bn_bwd_apply(s->d_pool_a, lp.z, B, lp.d.cout, lp.scale, lp.shift, lp.d.act, lp.cA, lp.cB, lp.cC, s->d_pool_z, st));
        RUN(pw_wgrad(s, s->pooled, lp.d.cin, lp.d.cin, s->d_pool_z, lp.d.cout, lp.d.cout, B, G + lp.d.w_off, st));
        PwArgs a = dgrad_args(s->d_pool_z, B, lp.d.cout, lp.d.cout, P + lp.d.w_off, lp.d.cin, s->d_pooled);
        a.scale = s->tmp_c + 2048;      // d mean / d feat = 1/HW, applied as a uniform scale
        RUN(launch_fill(s->tmp_c + 2048, lp.d.cin, (float)(1.0 / (double)HW), st));
        a.shift = s->tmp_c + 3072;
        RUN(launch_fill(s->tmp_c + 3072, lp.d.cin, 0.f, st));
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // aspp0: its input gradient also receives the pooled gradient, broadcast over the image
    LayerRt& lf = s->L[s->n_backbone];
    RUN(bn_backward(s, la, la.da, M, nHW, sc, st));
    RUN(pw_wgrad(s, lf.a, la.d.cin, la.d.cin, s->dz, la.d.cout, la.d.cout, M, G + la.d.w_off, st));
    {
        PwArgs a = dgrad_args(s->dz, M, la.d.cout, la.d.cout, P + la.d.w_off, la.d.cin, lf.da);
        a.img_bias = s->d_pooled; a.rows_per_img = HW;
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // backbone, last layer to first.  A layer's weight gradient only feeds the optimizer: it runs on the side stream while the main
    // stream goes on with the input gradient and the next layer's BN backward (many of these kernels are latency-bound at 8 frames
    // and share the chip well).  Nothing the side stream reads is overwritten inside the step (dz lives in per-layer memory), so the main
    // stream never waits for it before the final join.
    const bool overlap = s->overlap_wgrad && !s->prof.on && s->scratch2;
    if (overlap && !s->side) AMS_CHECK_HIP(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    if (overlap && !s->ev_xt) {
        if (!s->ev_fork) AMS_CHECK_HIP(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        AMS_CHECK_HIP(hipEventCreateWithFlags(&s->ev_xt, hipEventDisableTiming));
    }
    const bool three = overlap && s->overlap_wgrad >= 2 && s->scratch3;      // depthwise weight gradients on a third stream
    if (three && !s->side2) AMS_CHECK_HIP(hipStreamCreateWithFlags(&s->side2, hipStreamNonBlocking));
    bool xt_pending = false;
    // > 0: the kernel that produced this layer's da already multiplied it by the activation's derivative and left the BN-backward partial
    // rows in s->scratch (fused_stride floats apart; fused_dw: the nine taps of the NEXT layer's depthwise weight gradient behind the sums)
    int fused_rows = 0;
    int64_t fused_stride = 0;
    bool fused_dw = false;
    // where those rows are: the depthwise kernel's rows go to their own buffer when the side stream is there to reduce the nine taps
    // (they only feed the optimizer), so that the main stream's next user of s->scratch need not wait for that
    float* fused_buf = s->scratch;
    // reductions that only feed the optimizer (depthwise taps, ...) are collected and run as ONE launch at the end of the pass: their partial
    // rows stay in per-layer memory, so nothing on the way waits for them or signals them
    ReduceJobs deferred;
    for (int i = s->n_backbone; i >= 1; --i) {
        LayerRt& l = s->L[i];
        const int64_t Mo = (int64_t)B * l.px_out;
        // dz of a layer lives in that layer's own memory (in place over da, or dzp): no buffer is reused inside a step, so the main stream
        // never waits for a weight gradient (with two alternating dz buffers it did, ~40 times a step: 0.2 ms of real stalls behind
        // weight gradients that sharing the chip had stretched)
        float* dz = l.dzp ? l.dzp : l.da;
        if (fused_rows > 0) {
            // the depthwise input-gradient kernel of the layer behind this one already applied the activation's derivative and left the
            // partial sums (launch_depthwise_dgrad_bn): second stage of the reduction, then dz = A dy + B + C z
            const double n_l = (double)global_B * l.px_out;
            if (!sc || !sc->cb) {
                RUN(launch_bn_bwd_finalize_partials(fused_buf, fused_rows, fused_stride, l.d.cout, l.bsums, n_l, P + l.d.gamma_off, l.mean,
                                                    l.rstd, l.cA, l.cB, l.cC, G + l.d.gamma_off, G + l.d.beta_off, st));
            } else {
                RUN(launch_partials_to_sums(fused_buf, fused_rows, fused_stride, l.d.cout, l.bsums, st));
                RUN(launch_bn_param_grads(l.bsums, l.d.cout, G + l.d.gamma_off, G + l.d.beta_off, st));
                RUN(sync_doubles(sc, l.bsums, 2 * (size_t)l.d.cout, st));
                RUN(launch_bn_bwd_coef(l.bsums, n_l, l.d.cout, P + l.d.gamma_off, l.mean, l.rstd, l.cA, l.cB, l.cC, nullptr, nullptr, st));
            }
            // the depthwise layer's weight gradient came with the same rows (taps behind the two sums)
            if (fused_dw && !(fused_buf != s->scratch &&
                              deferred.add(fused_buf + 2 * (int64_t)l.d.cout, fused_rows, 9 * (int64_t)l.d.cout, G + s->L[i + 1].d.w_off, fused_stride)))
                RUN(launch_reduce_splits(fused_buf + 2 * (int64_t)l.d.cout, fused_rows, 9 * (int64_t)l.d.cout, G + s->L[i + 1].d.w_off, st, fused_stride));
            RUNK(0, 12.0 * Mo * l.d.cout, launch_bn_bwd_apply(l.da, l.z, Mo, l.d.cout, l.scale, l.shift, AMS_ACT_NONE, l.cA, l.cB, l.cC, dz, st));
            fused_rows = 0;
        } else {
            RUN(bn_backward(s, l, l.da, Mo, (double)global_B * l.px_out, sc, st, dz));
        }
        if (l.d.role == AMS_ROLE_DEPTHWISE && train_recompute_block(s, i - 1)) {
            // early block: from dz of the depthwise layer straight to the gradient of the block input; da_e / dz_e / a_e are recomputed
            // from the block input inside the kernels and never stored (k_xdw_train.hip)
            LayerRt& le = s->L[i - 1];
            LayerRt& lin = s->L[i - 2];
            const float* x = lin.a;
            int rows = 0;
            int64_t stride = 0;
            if (overlap && deferred.n > 0) {
                // the stride-16 blocks are behind us: their deferred reductions (80 MB of partial rows) run on the side stream under the early blocks
                AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
                AMS_CHECK_HIP(hipStreamWaitEvent(s->side, s->ev_fork, 0));
                RUN(launch_reduce_batch(deferred, s->side));
                deferred.n = 0;
            }
            if (xt_pending) { AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_xt, 0)); xt_pending = false; }     // xt_scratch is free again
            RUNK(i, 4.0 * ((double)B * (le.px_in * le.d.cin + l.px_out * l.d.cout)),
                 launch_xdw_bwd_reduce(x, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, le.d.cout, le.scale, le.shift, le.mean, le.rstd, le.d.act,
                                       P + l.d.w_off, l.d.stride, dz, s->xt_scratch, &rows, &stride, st));
            const double n_e = (double)global_B * le.px_out;
            if (!sc || !sc->cb) {
                RUN(launch_bn_bwd_finalize_partials(s->xt_scratch, rows, stride, le.d.cout, le.bsums, n_e, P + le.d.gamma_off, le.mean, le.rstd,
                                                    le.cA, le.cB, le.cC, G + le.d.gamma_off, G + le.d.beta_off, st));
            } else {
                RUN(launch_partials_to_sums(s->xt_scratch, rows, stride, le.d.cout, le.bsums, st));
                RUN(launch_bn_param_grads(le.bsums, le.d.cout, G + le.d.gamma_off, G + le.d.beta_off, st));
                RUN(sync_doubles(sc, le.bsums, 2 * (size_t)le.d.cout, st));
                RUN(launch_bn_bwd_coef(le.bsums, n_e, le.d.cout, P + le.d.gamma_off, le.mean, le.rstd, le.cA, le.cB, le.cC, nullptr, nullptr, st));
            }
            const float* skip = (i + 1 <= s->n_backbone && s->L[i + 1].d.residual_from == i - 2) ? s->L[i + 1].da : nullptr;
            // the block input is the previous block's project layer (BN, no activation): the first half of ITS BN backward rides on the
            // dx pass (sum dx, sum dx xhat as one partial row per block), the separate pass over (dx, z) disappears
            const bool red_dx = (s->fuse_gemm_red & 2) && i - 2 >= 3 && lin.d.role == AMS_ROLE_PROJECT && lin.d.act == AMS_ACT_NONE && lin.z &&
                                (size_t)2048 * 2 * lin.d.cout <= s->scratch_floats;
            int dx_rows = 0;
            RUNK(i - 1, 4.0 * ((double)B * (le.px_in * le.d.cin * (skip ? 3 : 2) + l.px_out * l.d.cout)),
                 launch_xdw_bwd_dx(x, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, le.d.cout, le.scale, le.shift, le.d.act, P + l.d.w_off,
                                   l.d.stride, dz, le.cA, le.cB, le.cC, skip, lin.da, st, red_dx ? lin.z : nullptr, lin.mean, lin.rstd,
                                   red_dx ? s->scratch : nullptr, red_dx ? &dx_rows : nullptr));
            if (dx_rows > 0) { fused_rows = dx_rows; fused_stride = 2 * (int64_t)lin.d.cout; fused_dw = false; fused_buf = s->scratch; }
            // weight gradients from the partial rows: depthwise taps, then the expand weights from (G1 | XX | g0).  They only feed the
            // optimizer: on the side stream, behind the coefficients (recorded before the dx pass, which does not touch the rows)
            const int KP = (le.d.cin + 15) / 16 * 16;
            const int64_t n_dw = 9 * (int64_t)le.d.cout, n_g = (int64_t)KP * le.d.cout;
            float* reduced = s->xt_scratch + (int64_t)rows * stride;
            hipStream_t xs = st;
            if (overlap) {
                AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
                AMS_CHECK_HIP(hipStreamWaitEvent(s->side, s->ev_fork, 0));
                xs = s->side;
            }
            RUN(launch_reduce_splits(s->xt_scratch + 2 * (int64_t)le.d.cout, rows, n_dw, G + l.d.w_off, xs, stride));
            RUN(launch_reduce_splits(s->xt_scratch + 11 * (int64_t)le.d.cout, rows, n_g, reduced, xs, stride));
            RUN(launch_xdw_dwe(reduced, le.xx_g0, le.d.cin, le.d.cout, P + le.d.w_off, le.cA, le.cB, le.cC, G + le.d.w_off, xs));
            if (overlap) { AMS_CHECK_HIP(hipEventRecord(s->ev_xt, s->side)); xt_pending = true; }
            --i;                                       // the expand layer is done
            continue;
        }
        if (i == 2 && stem_fused_train(s)) {
            // first block: the stem is the "expand" layer of this depthwise conv (a 1x1 conv over the 27-tap patch of the frame).  One pass
            // over dz and the frames gives the stem's BN-backward sums, the depthwise weight gradient and the pieces of the stem weight
            // gradient; da / dz of the stem, its im2col matrix and a_stem are never read or written in backward
            LayerRt& le = s->L[1];
            int rows = 0;
            int64_t stride = 0;
            if (xt_pending) { AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_xt, 0)); xt_pending = false; }
            RUNK(i, 4.0 * B * l.px_out * l.d.cout,
                 launch_xdw_bwd_reduce_stem(frames, dtype, B, c.height, c.width, c.pixel_scale, P + le.d.w_off, le.scale, le.shift, le.mean, le.rstd,
                                            le.d.act, P + l.d.w_off, dz, s->xt_scratch, &rows, &stride, st));
            const double n_e = (double)global_B * le.px_out;
            if (!sc || !sc->cb) {
                RUN(launch_bn_bwd_finalize_partials(s->xt_scratch, rows, stride, le.d.cout, le.bsums, n_e, P + le.d.gamma_off, le.mean, le.rstd,
                                                    le.cA, le.cB, le.cC, G + le.d.gamma_off, G + le.d.beta_off, st));
            } else {
                RUN(launch_partials_to_sums(s->xt_scratch, rows, stride, le.d.cout, le.bsums, st));
                RUN(launch_bn_param_grads(le.bsums, le.d.cout, G + le.d.gamma_off, G + le.d.beta_off, st));
                RUN(sync_doubles(sc, le.bsums, 2 * (size_t)le.d.cout, st));
                RUN(launch_bn_bwd_coef(le.bsums, n_e, le.d.cout, P + le.d.gamma_off, le.mean, le.rstd, le.cA, le.cB, le.cC, nullptr, nullptr, st));
            }
            float* reduced = s->xt_scratch + (int64_t)rows * stride;
            RUN(launch_reduce_splits(s->xt_scratch + 2 * 32, rows, 9 * 32, G + l.d.w_off, st, stride));
            RUN(launch_reduce_splits(s->xt_scratch + 11 * 32, rows, 32 * 32 + 32 * 32 + 32, reduced, st, stride));
            RUN(launch_xdw_dwe(reduced, reduced + 32 * 32, 27, 32, P + le.d.w_off, le.cA, le.cB, le.cC, G + le.d.w_off, st));
            break;                                     // the stem is done
        }
        if (l.d.role == AMS_ROLE_STEM) {
            RUN(launch_stem_im2col(frames, dtype, B, c.height, c.width, c.pixel_scale, s->im2col, st));
            RUN(pw_wgrad(s, s->im2col, 32, 27, dz, l.d.cout, l.d.cout, Mo, G + l.d.w_off, st));
            break;
        }
        LayerRt& prev = s->L[i - 1];
        if (dw_fused_train(s, i, B)) {
            // input gradient + activation derivative + BN-backward sums of the expand layer + this layer's weight gradient in one kernel
            // (k_conv.hip): prev.da <- dy, partial rows in s->scratch until the next iteration's second stage
            RUNK(i, dw_bytes(l, B) + 4.0 * B * l.px_in * l.d.cin,
                 launch_depthwise_dgrad_bn(dz, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, prev.z, prev.scale, prev.shift, prev.d.act, prev.mean,
                                           prev.rstd, prev.da, l.dw_rows ? l.dw_rows : s->scratch, &fused_rows, st));
            fused_stride = 11 * (int64_t)l.d.cin;
            fused_dw = true;
            fused_buf = l.dw_rows ? l.dw_rows : s->scratch;
            continue;
        }
        hipStream_t wst = st;
        float* wscratch = s->scratch;
        if (overlap) {
            const bool on2 = three && l.d.role == AMS_ROLE_DEPTHWISE;
            wst = on2 ? s->side2 : s->side;
            wscratch = on2 ? s->scratch3 : s->scratch2;
            AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
            AMS_CHECK_HIP(hipStreamWaitEvent(wst, s->ev_fork, 0));
        }
        if (l.d.role == AMS_ROLE_DEPTHWISE) {
            if (overlap) RUN(launch_depthwise_wgrad(prev.a, dz, B, l.Hin, l.Win, l.d.cin, l.d.stride, l.d.rate, G + l.d.w_off, wscratch, s->scratch_floats, wst));
            else RUNK(i, dw_bytes(l, B), launch_depthwise_wgrad(prev.a, dz, B, l.Hin, l.Win, l.d.cin, l.d.stride, l.d.rate, G + l.d.w_off,
                                                                wscratch, s->scratch_floats, wst));
        } else {
            RUN(pw_wgrad(s, prev.a, l.d.cin, l.d.cin, dz, l.d.cout, l.d.cout, Mo, G + l.d.w_off, wst, wscratch));
        }
        if (l.d.role == AMS_ROLE_DEPTHWISE) {
            RUNK(i, dw_bytes(l, B), launch_depthwise_dgrad(dz, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.stride, l.d.rate, prev.da, st));
        } else {
            PwArgs a = dgrad_args(dz, Mo, l.d.cout, l.d.cout, P + l.d.w_off, l.d.cin, prev.da);
            // the block input also feeds the residual add at the end of this block: add that gradient here
            if (l.d.role == AMS_ROLE_EXPAND && i + 2 <= s->n_backbone && s->L[i + 2].d.residual_from == i - 1) {
                a.res = s->L[i + 2].da; a.ldr = l.d.cin;
            }
            // first half of the previous layer's BN backward in this GEMM's epilogue (activation derivative + the two column sums), where
            // the kernel chosen can do it: the separate pass over (da, z) of that layer disappears
            int red_rows = 0;
            if ((s->fuse_gemm_red & 2) && i - 1 >= 2 && red_rows_bound(Mo) * 2 * (size_t)l.d.cin <= s->scratch_floats) {
                a.red_mode = 2; a.red_z = prev.z; a.red_scale = prev.scale; a.red_shift = prev.shift; a.red_mean = prev.mean; a.red_rstd = prev.rstd;
                a.red_act = prev.d.act; a.red_part = s->scratch; a.red_rows_out = &red_rows;
            }
            RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
            if (red_rows > 0) { fused_rows = red_rows; fused_stride = 2 * (int64_t)l.d.cin; fused_dw = false; fused_buf = s->scratch; }
        }
    }
    RUN(launch_reduce_batch(deferred, st));
    // the optimizer (and the gradient all-reduce) wait for every weight gradient
    if (xt_pending) AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_xt, 0));
    if (overlap) {                                     // everything either side stream still holds (events cover the last launch of each buffer only)
        AMS_CHECK_HIP(hipEventRecord(s->ev_fork, s->side));
        AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_fork, 0));
        if (three) { AMS_CHECK_HIP(hipEventRecord(s->ev_fork, s->side2)); AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_fork, 0)); }
    }
    return AMS_OK;
}

static int loss_forward(ams_student* s, const uint8_t* teacher, int B, int32_t* labels, hipStream_t st) {
    const ams_student_config& c = s->cfg;
    if (!labels && ce_loss_grad_supported(s->w, c.width)) {
        // the fine-tune step: loss sums and the unnormalised gradient in one pass over the pixels
        RUNK(0, 0.0, launch_ce_loss_grad(s->logits, 32, B, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher, c.num_classes,
                                         s->loss_buf, s->ce_scratch, st));
        return AMS_OK;
    }
    return launch_upsample_argmax(s->logits, 32, B, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher,
                                  c.num_classes, labels, s->conf_buf, s->loss_buf, st);
}

}  // namespace ams

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" {

const char* ams_last_error(void) { return ams::last_error(); }
int ams_abi_version(void) { return AMS_ABI_VERSION; }

int ams_device_info(char* name_out, size_t name_cap, int32_t* n_cu, int64_t* hbm_bytes) {
    int dev = 0;
    AMS_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    AMS_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (name_out && name_cap) { snprintf(name_out, name_cap, "%s (%s)", p.name, p.gcnArchName); }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return AMS_OK;
}

int ams_student_arena_bytes(const ams_student_config* cfg, const ams_layer_desc* layers, size_t* bytes_out) {
    AMS_REQUIRE(bytes_out, "arena_bytes: null output");
    ams_student tmp;
    int rc = build(&tmp, cfg, layers);
    if (rc) return rc;
    return layout(&tmp, nullptr, bytes_out);
}

int ams_student_create(const ams_student_config* cfg, const ams_layer_desc* layers, void* arena_dev, size_t arena_bytes,
                       ams_student** out) {
    AMS_REQUIRE(out && arena_dev, "create: null pointer");
    AMS_REQUIRE(((uintptr_t)arena_dev & 255) == 0, "create: arena must be 256-byte aligned");
    ams_student* s = new ams_student();
    int rc = build(s, cfg, layers);
    size_t need = 0;
    if (!rc) rc = layout(s, arena_dev, &need);
    if (!rc && need > arena_bytes) { set_error("create: arena too small (%zu < %zu)", arena_bytes, need); rc = AMS_E_NOMEM; }
    if (rc) { delete s; return rc; }
    s->arena = (char*)arena_dev;
    s->arena_bytes = arena_bytes;
    {   // events are free; STREAMS are not: the runtime multiplexes them onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by
        // default), and two streams that land on one queue serialise.  A student therefore owns only the streams it uses: the part
        // streams appear with the first multi-part call (never inside a graph capture: ensure_part_streams).
        hipError_t e = hipEventCreateWithFlags(&s->ev_fork_dual, hipEventDisableTiming);
        for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&s->part_done[k], hipEventDisableTiming);
        if (e != hipSuccess) { set_error("create: events -> %s", hipGetErrorString(e)); delete s; return AMS_E_HIP; }
    }
    if (s->vec_ones) {
        std::vector<float> ones(1024, 1.0f), zeros(1024, 0.0f);
        hipError_t e = hipMemcpy(s->vec_ones, ones.data(), 1024 * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->vec_zeros, zeros.data(), 1024 * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_error("create: uploading constants -> %s", hipGetErrorString(e)); delete s; return AMS_E_HIP; }
    }
    if (!s->tp_jobs.empty()) {
        for (size_t k = 0; k < s->tp_jobs.size(); ++k) {
            SplitJob& j = s->tp_jobs[k];
            j.w = s->params + (int64_t)(uintptr_t)j.w;
            j.p0 = s->tp_panels + (size_t)(uintptr_t)j.p0;
            s->tp_index[{j.w, j.sk == 1 ? 1 : 0}] = (int)k;
        }
        const hipError_t e = hipMemcpy(s->tp_jobs_dev, s->tp_jobs.data(), s->tp_jobs.size() * sizeof(SplitJob), hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_error("create: uploading the panel table -> %s", hipGetErrorString(e)); delete s; return AMS_E_HIP; }
    }
    if (const char* e = getenv("AMS_BLOCK_X6")) s->block_x6 = atoi(e);                       // tuning knob (see AMS_OPT_BLOCK_X6)
    if (const char* e = getenv("AMS_LATE_SUB")) s->late_subbatch = atoi(e);                  // tuning knob (see AMS_OPT_LATE_SUBBATCH)
    if (const char* e = getenv("AMS_STREAM_MIN_ROWS")) s->stream_min_rows = atoll(e);        // tuning knob
    if (const char* e = getenv("AMS_DUAL_PARTS")) s->dual_parts = atoi(e);                  // tuning knob
    if (const char* e = getenv("AMS_DUAL_STREAM")) s->dual_stream = atoi(e);                // tuning knob (see AMS_OPT_DUAL_STREAM)
    if (const char* e = getenv("AMS_DUAL_AUTOTUNE")) s->dual_autotune = atoi(e);            // tuning knob (see AMS_OPT_DUAL_AUTOTUNE)
    if (const char* e = getenv("AMS_OVERLAP_HEAD")) s->overlap_head = atoi(e);              // tuning knob
    if (const char* e = getenv("AMS_FUSE_BLOCK")) s->fuse_block = atoi(e);                  // tuning knob (see AMS_OPT_FUSE_BLOCK)
    if (const char* e = getenv("AMS_FUSE_XDS")) s->fuse_expand_dw_stream = atoi(e);      // tuning knob (see AMS_OPT_FUSE_EXPAND_DW_STREAM)
    if (const char* e = getenv("AMS_OVERLAP_WGRAD")) s->overlap_wgrad = atoi(e);            // tuning knob: 0 one stream, 1 weight gradients on a side stream, 2 depthwise ones on a third
    if (const char* e = getenv("AMS_FUSE_DGRAD_BN")) s->fuse_dgrad_bn = atoi(e);             // tuning knob
    if (const char* e = getenv("AMS_FUSE_GEMM_RED")) s->fuse_gemm_red = atoi(e);             // tuning knob
    if (const char* e = getenv("AMS_TRAIN_RECOMPUTE")) s->train_recompute = atoi(e);       // tuning knob (see AMS_OPT_TRAIN_RECOMPUTE)
    *out = s;
    return AMS_OK;
}

void ams_student_destroy(ams_student* s) { delete s; }

int ams_student_region(const ams_student* s, int32_t region, size_t* offset_bytes, size_t* n_elems) {
    AMS_REQUIRE(s && offset_bytes && n_elems, "region: null pointer");
    const void* p = nullptr;
    size_t n = 0;
    switch (region) {
        case AMS_REGION_PARAMS: p = s->params; n = s->cfg.n_trainable; break;
        case AMS_REGION_STATS: p = s->stats; n = s->cfg.n_stats; break;
        case AMS_REGION_GRADS: p = s->grads; n = s->cfg.n_trainable; break;
        case AMS_REGION_ADAM_M: p = s->adam_m; n = s->cfg.n_trainable; break;
        case AMS_REGION_ADAM_V: p = s->adam_v; n = s->cfg.n_trainable; break;
        case AMS_REGION_FROZEN: p = s->fparams; n = s->cfg.n_trainable; break;
        case AMS_REGION_BN_SYNC: p = s->bn_sync; n = s->bn_sync_doubles; break;
        case AMS_REGION_LOGITS: p = s->logits; n = (size_t)s->cfg.max_batch * s->h * s->w * 32; break;
        default: set_error("region: unknown region %d", region); return AMS_E_INVALID;
    }
    if (!p) { set_error("region %d is not allocated for this student (trainable=%d)", region, s->cfg.trainable); return AMS_E_STATE; }
    *offset_bytes = (size_t)((const char*)p - s->arena);
    *n_elems = n;
    return AMS_OK;
}

int ams_student_lowres_size(const ams_student* s, int32_t* h, int32_t* w) {
    AMS_REQUIRE(s && h && w, "lowres_size: null pointer");
    *h = s->h; *w = s->w;
    return AMS_OK;
}

int ams_student_freeze(ams_student* s, void* stream) {
    AMS_REQUIRE(s, "freeze: null student");
    hipStream_t st = (hipStream_t)stream;
    RUN(launch_copy(s->fparams, s->params, s->cfg.n_trainable, st));
    RUN(launch_copy(s->fstats, s->stats, s->cfg.n_stats, st));
    for (int i = 1; i <= s->cfg.n_layers; ++i) {
        LayerRt& l = s->L[i];
        if (l.d.bn_eps < 0) continue;
        RUN(launch_bn_fold(s->fparams + l.d.gamma_off, s->fparams + l.d.beta_off, s->fstats + l.d.mean_off, s->fstats + l.d.var_off,
                           s->cfg.bn_eps_frozen, l.d.cout, l.fscale, l.fshift, st));
    }
    for (int i = 1; i + 1 <= s->n_backbone; ++i) {
        LayerRt& l = s->L[i];
        LayerRt& ld = s->L[i + 1];
        if (l.blk_vecs) RUN(launch_block_pack(l.fscale, l.fshift, ld.fscale, ld.fshift, s->fparams + ld.d.w_off, l.d.cout, l.blk_vecs, st));
    }
    if (s->L[1].whi)        // stem: [27][32] -> parts [32][32], k = tap * 3 + channel
        RUN(launch_split_weights3(s->fparams + s->L[1].d.w_off, 32, 1, 27, 32, 32, s->L[1].whi, s->L[1].wlo, s->L[1].wlo3, st));
    for (int i = 2; i <= s->cfg.n_layers; ++i) {
        LayerRt& l = s->L[i];
        if (!l.whi) continue;
        const int K = l.d.cin - l.split_k0;
        RUN(launch_split_weights3(s->fparams + l.d.w_off + (int64_t)l.split_k0 * l.d.cout, l.d.cout, 1, K, l.d.cout, l.Kp, l.whi, l.wlo,
                                  l.wlo3, st));
    }
    s->frozen_ready = true;
    return AMS_OK;
}

static int check_call(const ams_student* s, const void* frames, int dtype, int batch) {
    AMS_REQUIRE(s && frames, "null student or frames");
    AMS_REQUIRE(dtype == AMS_DT_U8 || dtype == AMS_DT_F32, "frames must be uint8 or float32");
    AMS_REQUIRE(batch > 0 && batch <= s->cfg.max_batch, "batch %d outside 1..%d", batch, s->cfg.max_batch);
    return AMS_OK;
}

// A slice of the student's frozen-inference buffers: frames b0 .. b0 + bp - 1 of every activation / head buffer (all sized for max_batch
// frames).  While the guard lives, forward_frozen works inside that slice; the pointers come back whatever way the scope is left.
struct SliceGuard {
    ams_student* s;
    float* act0[4]; uint16_t* xs0; size_t xp0; float *pooled0, *pool_a0, *img_bias0, *logits0, *scratch0;
    SliceGuard(ams_student* s_, int b0, int bp) : s(s_) {
        const ams_student_config& c = s->cfg;
        for (int k = 0; k < 4; ++k) act0[k] = s->act[k];
        xs0 = s->xsplit; xp0 = s->xsplit_plane;
        pooled0 = s->pooled; pool_a0 = s->pool_a; img_bias0 = s->img_bias; logits0 = s->logits; scratch0 = s->scratch;
        const LayerRt& lp = s->L[s->iPool]; const LayerRt& lc = s->L[s->iProj];
        const size_t per_frame = s->act_elems / c.max_batch;
        for (int k = 0; k < 4; ++k) s->act[k] = act0[k] + (size_t)b0 * per_frame;
        if (xs0) { s->xsplit = xs0 + 3 * (xp0 / c.max_batch) * b0; s->xsplit_plane = (xp0 / c.max_batch) * bp; }
        s->pooled = pooled0 + (size_t)b0 * lp.d.cin;
        s->pool_a = pool_a0 + (size_t)b0 * lp.d.cout;
        s->img_bias = img_bias0 + (size_t)b0 * lc.d.cout;
        s->logits = logits0 + (size_t)b0 * s->h * s->w * 32;
        s->scratch = scratch0 + image_colsum_scratch(b0, lp.d.cin);
    }
    ~SliceGuard() {
        for (int k = 0; k < 4; ++k) s->act[k] = act0[k];
        s->xsplit = xs0; s->xsplit_plane = xp0;
        s->pooled = pooled0; s->pool_a = pool_a0; s->img_bias = img_bias0; s->logits = logits0; s->scratch = scratch0;
    }
};

// Frozen inference as two to four parts on as many streams: the parts run the same layer sequence side by side (part 0 on the caller's
// stream, the others on streams the student owns; one fork and one join per step), each in its own slice of every activation buffer.
// A launch of this network rarely fills the chip to the end — tails of 1.05- or 2.1-round grids, latency-bound chains on a few blocks
// per CU — and the other parts' kernels fill those gaps.  Every frame is computed exactly as in a batch of the part's size.
// Whatever happens inside, every part stream is joined back into `st` before this returns.
// the part streams of an n-part plan exist (created on first use, outside any graph capture); false: run the one-stream plan instead
static bool ensure_part_streams(ams_student* s, int nparts, hipStream_t st) {
    bool missing = false;
    for (int p = 1; p < nparts; ++p) missing = missing || !s->part_stream[p - 1];
    if (!missing) return true;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;      // nothing is created inside a capture
    for (int p = 1; p < nparts; ++p)
        if (!s->part_stream[p - 1] && hipStreamCreateWithFlags(&s->part_stream[p - 1], hipStreamNonBlocking) != hipSuccess) return false;
    return true;
}

static int forward_frozen_dual(ams_student* s, const void* frames, int dtype, int batch, hipStream_t st, int nparts = 2) {
    const ams_student_config& c = s->cfg;
    if (nparts < 2) nparts = 2;
    if (nparts > 4) nparts = 4;
    if (nparts > batch) nparts = batch;
    if (nparts < 2 || !ensure_part_streams(s, nparts, st)) return forward_frozen(s, frames, dtype, batch, st);
    AMS_REQUIRE(s->ev_fork_dual && s->part_stream[nparts - 2] && s->part_done[nparts - 2], "dual plan: part streams were not created");
    AMS_CHECK_HIP(hipEventRecord(s->ev_fork_dual, st));
    const size_t frame_bytes = (size_t)c.height * c.width * 3 * (dtype == AMS_DT_U8 ? 1 : 4);
    int rc = AMS_OK;
    int b0 = 0, forked = 0;
    for (int p = 0; p < nparts && !rc; ++p) {
        const int bp = batch / nparts + (p < batch % nparts ? 1 : 0);
        hipStream_t ps = p == 0 ? st : s->part_stream[p - 1];
        if (p > 0) {
            if (hipStreamWaitEvent(ps, s->ev_fork_dual, 0) != hipSuccess) { set_error("dual plan: fork failed"); rc = AMS_E_HIP; break; }
            forked = p;
        }
        {
            SliceGuard slice(s, b0, bp);
            rc = forward_frozen(s, (const char*)frames + (size_t)b0 * frame_bytes, dtype, bp, ps);
        }
        b0 += bp;
    }
    // join every stream that was forked, error or not: no part may still be writing the student's buffers after the return
    for (int p = 1; p <= forked; ++p) {
        if (hipEventRecord(s->part_done[p - 1], s->part_stream[p - 1]) != hipSuccess ||
            hipStreamWaitEvent(st, s->part_done[p - 1], 0) != hipSuccess) {
            (void)hipStreamSynchronize(s->part_stream[p - 1]);        // last resort: a host wait keeps the guarantee
            if (!rc) { set_error("dual plan: join failed"); rc = AMS_E_HIP; }
        }
    }
    return rc;
}

// Parts of the static rule (AMS_OPT_DUAL_STREAM = 1): where the one-stream grids quantise badly at 512 x 1024 (round-2 sweep on MI355X:
// two parts +3.5 % at 32-36 frames and at 64, three at 48; a loss of 1-5 % at 24-30 and 40; nothing either way elsewhere).  A fixed
// function of the batch size: the same call always runs the same plan, and nothing is timed inside a call.
static int dual_parts_static(int batch) {
    if (batch >= 32 && batch <= 36) return 2;
    if (batch == 48) return 3;
    if (batch == 64) return 2;
    return 1;
}

static int run_forward(ams_student* s, const void* frames, int dtype, int batch, int mode, hipStream_t st) {
    if (mode == AMS_MODE_FROZEN) {
        if (!s->frozen_ready) { set_error("predict: ams_student_freeze has not been called"); return AMS_E_STATE; }
        if (s->dual_stream == 0 || s->prof.on || s->late_subbatch != 0 || batch < 2) return forward_frozen(s, frames, dtype, batch, st);
        if (s->dual_stream >= 2) return batch >= s->dual_stream ? forward_frozen_dual(s, frames, dtype, batch, st, s->dual_parts) : forward_frozen(s, frames, dtype, batch, st);
        if (!s->dual_autotune) {
            const int n = dual_parts_static(batch);
            return n > 1 ? forward_frozen_dual(s, frames, dtype, batch, st, n) : forward_frozen(s, frames, dtype, batch, st);
        }
        if (batch < 16) return forward_frozen(s, frames, dtype, batch, st);
        auto it = s->dual_choice.find(batch);
        if (it == s->dual_choice.end()) {
            // AMS_OPT_DUAL_AUTOTUNE (opt-in; this branch synchronises the host): the first call with this batch size times the one-stream
            // plan and the 2- to 4-part plans on these very frames — median of three timed passes each, after a warm-up pass — keeps a
            // multi-part plan only when it wins by more than the timing noise, and finishes with a pass of the chosen plan
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(st, &cap);
            if (cap != hipStreamCaptureStatusNone) return forward_frozen(s, frames, dtype, batch, st);      // no timing inside a capture
            hipEvent_t e0 = nullptr, e1 = nullptr;
            AMS_CHECK_HIP(hipEventCreate(&e0));
            if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("autotune: hipEventCreate failed"); return AMS_E_HIP; }
            float ms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};          // ms[n]: the batch in n parts (n = 1: one stream)
            int rc = AMS_OK;
            hipError_t he = hipSuccess;
            const int max_parts = batch >= 32 ? 4 : batch >= 24 ? 3 : 2;       // parts of at least 8 frames
            for (int n = 1; n <= max_parts && !rc && he == hipSuccess; ++n) {
                float t[3] = {0.f, 0.f, 0.f};
                rc = n > 1 ? forward_frozen_dual(s, frames, dtype, batch, st, n) : forward_frozen(s, frames, dtype, batch, st);      // warm-up
                for (int rep = 0; rep < 3 && !rc && he == hipSuccess; ++rep) {
                    he = hipEventRecord(e0, st);
                    rc = n > 1 ? forward_frozen_dual(s, frames, dtype, batch, st, n) : forward_frozen(s, frames, dtype, batch, st);
                    if (he == hipSuccess) he = hipEventRecord(e1, st);
                    if (he == hipSuccess) he = hipEventSynchronize(e1);
                    if (he == hipSuccess) he = hipEventElapsedTime(&t[rep], e0, e1);
                }
                const float lo = t[0] < t[1] ? t[0] : t[1], hi = t[0] < t[1] ? t[1] : t[0];
                ms[n] = t[2] < lo ? lo : (t[2] > hi ? hi : t[2]);                                  // median of three
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (rc) return rc;
            if (he != hipSuccess) { set_error("autotune: event timing failed: %s", hipGetErrorString(he)); return AMS_E_HIP; }
            int best = 1;
            for (int n = 2; n <= max_parts; ++n)
                if (ms[n] < 0.985f * ms[1] && (best == 1 || ms[n] < ms[best])) best = n;
            it = s->dual_choice.emplace(batch, best).first;
        }
        return it->second > 1 ? forward_frozen_dual(s, frames, dtype, batch, st, it->second) : forward_frozen(s, frames, dtype, batch, st);
    }
    if (mode == AMS_MODE_LIVE) return forward_live(s, frames, dtype, batch, batch, /*update_ema=*/false, nullptr, st);
    set_error("predict: unknown mode %d", mode);
    return AMS_E_INVALID;
}

int ams_student_predict(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch, int32_t mode,
                        int32_t* labels_out_dev, void* stream) {
    RUN(check_call(s, frames_dev, frames_dtype, batch));
    AMS_REQUIRE(labels_out_dev, "predict: null output");
    hipStream_t st = (hipStream_t)stream;
    RUN(run_forward(s, frames_dev, frames_dtype, batch, mode, st));
    const ams_student_config& c = s->cfg;
    return launch_upsample_argmax(s->logits, 32, batch, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, nullptr,
                                  c.num_classes, labels_out_dev, nullptr, nullptr, st);
}

int ams_student_predict_with_metric(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch, int32_t mode,
                                    const uint8_t* teacher_dev, int32_t* labels_out_dev, int64_t* conf_mat_dev, double* loss_dev,
                                    void* stream) {
    RUN(check_call(s, frames_dev, frames_dtype, batch));
    AMS_REQUIRE(teacher_dev && labels_out_dev && conf_mat_dev && loss_dev, "predict_with_metric: null pointer");
    hipStream_t st = (hipStream_t)stream;
    RUN(run_forward(s, frames_dev, frames_dtype, batch, mode, st));
    const ams_student_config& c = s->cfg;
    return launch_upsample_argmax(s->logits, 32, batch, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher_dev,
                                  c.num_classes, labels_out_dev, conf_mat_dev, loss_dev, st);
}

int ams_student_predict_frames(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch, int32_t mode, const uint8_t* teacher_dev,
                               int32_t* labels_out_dev, int64_t* conf_mats_dev, double* losses_dev, void* stream) {
    RUN(check_call(s, frames_dev, frames_dtype, batch));
    AMS_REQUIRE(labels_out_dev, "predict_frames: null output");
    AMS_REQUIRE(teacher_dev == nullptr || (conf_mats_dev && losses_dev), "predict_frames: metrics need conf and loss buffers");
    hipStream_t st = (hipStream_t)stream;
    RUN(run_forward(s, frames_dev, frames_dtype, batch, mode, st));
    const ams_student_config& c = s->cfg;
    return launch_upsample_argmax(s->logits, 32, batch, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher_dev, c.num_classes,
                                  labels_out_dev, teacher_dev ? conf_mats_dev : nullptr, teacher_dev ? losses_dev : nullptr, st, /*per_frame=*/1);
}

int ams_cross_confusion(const ams_student* s, const uint8_t* labels_dev, int64_t n_pixels, int64_t* conf_mat_dev, void* stream) {
    AMS_REQUIRE(s && labels_dev && conf_mat_dev && n_pixels > 0, "cross_confusion: bad argument");
    int32_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = -1;
    for (int k = 0; k < s->cfg.n_selected; ++k) lut[s->cfg.class_indices[k]] = k;
    return launch_cross_confusion(labels_dev, labels_dev + n_pixels, n_pixels, lut, s->cfg.n_selected, conf_mat_dev,
                                  (hipStream_t)stream);
}

static int train_step_impl(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev,
                           int32_t batch, int32_t global_batch, float lr, const uint8_t* mask_dev, double* loss_dev,
                           ams_allreduce_cb cb, void* user, ams_comm* comm, void* stream) {
    RUN(check_call(s, frames_dev, frames_dtype, batch));
    AMS_REQUIRE(teacher_dev, "train_step: null teacher labels");
    if (!s->cfg.trainable) { set_error("train_step: this student was created frozen (trainable=0)"); return AMS_E_STATE; }
    AMS_REQUIRE(global_batch >= batch, "train_step: global batch %d < local batch %d", global_batch, batch);
    hipStream_t st = (hipStream_t)stream;
    if (comm) cb = comm_as_cb;
    SyncCtx sc{cb, user, s, comm};
    const SyncCtx* psc = cb ? &sc : nullptr;
    RUN(forward_live(s, frames_dev, frames_dtype, batch, global_batch, /*update_ema=*/true, psc, st));
    RUN(loss_forward(s, teacher_dev, batch, nullptr, st));
    RUN(sync_doubles(psc, s->loss_buf, 2, st));         // loss sum and valid-pixel count over all ranks
    RUN(backward(s, frames_dev, frames_dtype, teacher_dev, batch, global_batch, psc, st));
    RUN(sync_any(psc, s->grads, (size_t)s->cfg.n_trainable, AMS_DT_F32, st));       // one flat 8.45 MB message
    if (loss_dev) AMS_CHECK_HIP(hipMemcpyAsync(loss_dev, s->loss_buf, 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    // Adam, TF1 form; the step counter is never reset (SemanticNetwork.py:25, :154-156)
    s->adam_t += 1;
    // beta1 / beta2 are f32 tensors in the TF graph (0.9f, 0.999f), and so are their running powers
    const double b1 = (double)0.9f, b2 = (double)0.999f;
    const double lr_t = (double)lr * sqrt(1.0 - pow(b2, (double)s->adam_t)) / (1.0 - pow(b1, (double)s->adam_t));
    s->tp_fresh = false;                               // the update below invalidates the live weight panels
    return launch_adam(s->params, s->grads, s->adam_m, s->adam_v, mask_dev, s->cfg.n_trainable, (float)lr_t, 0.9f, 0.999f, 1e-8f, st);
}

int ams_student_train_step_dp(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev,
                              int32_t batch, int32_t global_batch, float lr, const uint8_t* mask_dev, double* loss_dev,
                              ams_allreduce_cb cb, void* user, void* stream) {
    return train_step_impl(s, frames_dev, frames_dtype, teacher_dev, batch, global_batch, lr, mask_dev, loss_dev, cb, user, nullptr, stream);
}

int ams_student_train_step_rccl(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev,
                                int32_t batch, int32_t global_batch, float lr, const uint8_t* mask_dev, double* loss_dev,
                                ams_comm* comm, void* stream) {
    AMS_REQUIRE(comm, "train_step_rccl: null communicator");
    return train_step_impl(s, frames_dev, frames_dtype, teacher_dev, batch, global_batch, lr, mask_dev, loss_dev, nullptr, nullptr, comm, stream);
}

int ams_student_train_step(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev, int32_t batch,
                           float lr, const uint8_t* mask_dev, double* loss_dev, void* stream) {
    return train_step_impl(s, frames_dev, frames_dtype, teacher_dev, batch, batch, lr, mask_dev, loss_dev, nullptr, nullptr, nullptr, stream);
}

int ams_student_set_option(ams_student* s, int32_t option, int32_t value) {
    AMS_REQUIRE(s, "set_option: null student");
    s->dual_choice.clear();                            // any option may change the plans the autotune compared
    if (option == AMS_OPT_MATMUL) {
        AMS_REQUIRE(value == AMS_MATMUL_F32 || value == AMS_MATMUL_SPLIT_BF16 || value == AMS_MATMUL_SPLIT_BF16_X6 || value == AMS_MATMUL_BF16,
                    "set_option: unknown matmul mode %d", value);
        s->matmul_mode = value;
        return AMS_OK;
    }
    if (option == AMS_OPT_DUAL_AUTOTUNE) {
        s->dual_autotune = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_DUAL_PARTS) {
        AMS_REQUIRE(value >= 2 && value <= 4, "set_option: AMS_OPT_DUAL_PARTS must be 2 .. 4");
        s->dual_parts = value;
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_FIRST_BLOCK) {
        s->fuse_first_block = value < 0 ? 0 : (value > 2 ? 2 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_DW_PROJECT) {
        s->fuse_dw_project = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_EXPAND_DW_STREAM) {
        s->fuse_expand_dw_stream = value < 0 ? 0 : (value > 2 ? 2 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_BLOCK_X6) {
        s->block_x6 = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_LATE_SUBBATCH) {
        s->late_subbatch = value < 0 ? 0 : value;
        return AMS_OK;
    }
    if (option == AMS_OPT_DUAL_STREAM) {
        s->dual_stream = value < 0 ? 0 : value;
        s->dual_choice.clear();
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_BLOCK) {
        s->fuse_block = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_EXPAND_DW) {
        s->fuse_expand_dw = value < 0 ? 0 : (value > 2 ? 2 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_EMULATE_BF16_STORAGE) {
        s->emulate_bf16_storage = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_DGRAD_BN) {
        s->fuse_dgrad_bn = value < 0 ? 0 : (value > 2 ? 2 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_GEMM_RED) {
        s->fuse_gemm_red = value & 3;
        return AMS_OK;
    }
    if (option == AMS_OPT_TRAIN_RECOMPUTE) {
        s->train_recompute = value != 0;
        return AMS_OK;
    }
    set_error("set_option: unknown option %d", option);
    return AMS_E_INVALID;
}

int ams_student_profile(ams_student* s, int32_t enable) {
    AMS_REQUIRE(s, "profile: null student");
    AMS_CHECK_HIP(hipDeviceSynchronize());
    s->prof.clear();
    s->prof.on = enable != 0;
    return AMS_OK;
}

int ams_student_profile_read(ams_student* s, char* buf, size_t cap, size_t* needed) {
    AMS_REQUIRE(s && needed, "profile_read: null pointer");
    AMS_CHECK_HIP(hipDeviceSynchronize());
    std::string out;
    char line[256];
    for (auto& r : s->prof.recs) {
        float ms = 0.f;
        AMS_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        snprintf(line, sizeof(line), "%s\t%d\t%.6f\t%.0f\t%.0f\t%.0f\n", r.name.c_str(), r.layer, ms, r.bytes, r.flops, r.flops_x6);
        out += line;
    }
    *needed = out.size() + 1;
    if (buf && cap >= out.size() + 1) memcpy(buf, out.c_str(), out.size() + 1);
    return AMS_OK;
}

int ams_student_get_adam_step(const ams_student* s, int64_t* t) {
    AMS_REQUIRE(s && t, "get_adam_step: null pointer");
    *t = s->adam_t;
    return AMS_OK;
}
int ams_student_set_adam_step(ams_student* s, int64_t t) {
    AMS_REQUIRE(s && t >= 0, "set_adam_step: bad argument");
    s->adam_t = t;
    return AMS_OK;
}

size_t ams_pack_masked_fp16_scratch(int64_t n) { return n > 0 ? pack_fp16_scratch(n) : 0; }

int ams_pack_masked_fp16(const float* params_dev, const uint8_t* mask_dev, int64_t n, uint16_t* out_half_dev, int64_t* n_out_dev,
                         int64_t* scratch_dev, size_t scratch_elems, void* stream) {
    AMS_REQUIRE(params_dev && out_half_dev && n_out_dev && n > 0, "pack_masked_fp16: bad argument");
    AMS_REQUIRE(scratch_dev && scratch_elems >= pack_fp16_scratch(n), "pack_masked_fp16: scratch too small (need %zu int64)",
                pack_fp16_scratch(n));
    return launch_pack_fp16(params_dev, mask_dev, n, out_half_dev, n_out_dev, scratch_dev, (hipStream_t)stream);
}

// ---- kernel-level entry points -----------------------------------------------------------------------------
int ams_k_stem_conv(const void* frames, int32_t frames_dtype, int32_t B, int32_t H, int32_t W, const float* w, int32_t cout,
                    const float* scale, const float* shift, int32_t act, float pixel_scale, float* y, void* stream) {
    return launch_stem(frames, frames_dtype, B, H, W, w, cout, scale, shift, act, pixel_scale, y, (hipStream_t)stream);
}

int ams_k_depthwise3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t stride, int32_t rate,
                       const float* scale, const float* shift, int32_t act, float* y, void* stream) {
    return launch_depthwise(x, B, H, W, C, w, stride, rate, scale, shift, act, y, (hipStream_t)stream);
}

int ams_k_pointwise(const float* x, int64_t M, int32_t K, const float* w, int32_t N, int32_t trans_w, const float* img_bias,
                    int64_t rows_per_img, const float* scale, const float* shift, int32_t act, const float* res, float* y,
                    void* stream) {
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    if (trans_w) { a.w_sk = 1; a.w_sn = K; }
    a.img_bias = img_bias; a.rows_per_img = rows_per_img > 0 ? rows_per_img : 1;
    a.scale = scale; a.shift = shift; a.act = act; a.res = res; a.ldr = N;
    if (scale && !shift) { set_error("pointwise: scale without shift"); return AMS_E_INVALID; }
    return launch_pointwise(a, (hipStream_t)stream);
}

int ams_k_pointwise_split(const float* x, int64_t M, int32_t K, const float* w, int32_t N, const float* scale, const float* shift,
                          int32_t act, const float* res, float* y, uint16_t* panels, size_t panel_elems, void* stream) {
    const int Kp = (K + 31) / 32 * 32;
    AMS_REQUIRE(panels && panel_elems >= (size_t)2 * N * Kp, "pointwise_split: panel scratch too small (need %zu)", (size_t)2 * N * Kp);
    hipStream_t st = (hipStream_t)stream;
    uint16_t* hi = panels;
    uint16_t* lo = panels + (size_t)N * Kp;
    RUN(launch_split_weights(w, N, 1, K, N, Kp, hi, lo, st));
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    a.scale = scale; a.shift = shift; a.act = act; a.res = res; a.ldr = N;
    if (scale && !shift) { set_error("pointwise_split: scale without shift"); return AMS_E_INVALID; }
    return launch_pointwise_split(a, hi, lo, Kp, st);
}

int ams_k_pointwise_split3(const float* x, int64_t M, int32_t K, const float* w, int32_t N, const float* scale, const float* shift,
                           int32_t act, const float* res, float* y, uint16_t* panels, size_t panel_elems, void* stream) {
    const int Kp = (K + 31) / 32 * 32;
    const size_t plane = (size_t)N * Kp;
    AMS_REQUIRE(panels && panel_elems >= 3 * plane, "pointwise_split3: panel scratch too small (need %zu)", 3 * plane);
    hipStream_t st = (hipStream_t)stream;
    RUN(launch_split_weights3(w, N, 1, K, N, Kp, panels, panels + plane, panels + 2 * plane, st));
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    a.scale = scale; a.shift = shift; a.act = act; a.res = res; a.ldr = N;
    if (scale && !shift) { set_error("pointwise_split3: scale without shift"); return AMS_E_INVALID; }
    return launch_pointwise_split3(a, panels, panels + plane, panels + 2 * plane, Kp, st);
}

int ams_ingest_resize_u8(const uint8_t* src, int32_t Hs, int32_t Ws, int32_t Cn, int32_t mode, int32_t swap_rb, uint8_t* dst, int32_t H,
                         int32_t W, void* stream) {
    return launch_resize_u8(src, Hs, Ws, Cn, mode, swap_rb, dst, H, W, (hipStream_t)stream);
}

int ams_k_dw_project(const float* e, int32_t B, int32_t H, int32_t W, int32_t Cc, const float* w_dw, int32_t rate, const float* scale_d,
                     const float* shift_d, const float* w_proj, int32_t N, const float* scale_p, const float* shift_p,
                     const float* res, float* y, uint16_t* panels, size_t panel_elems, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!dw_project_supported(Cc, N, 1, rate)) { set_error("dw_project: unsupported shape C=%d N=%d rate=%d", Cc, N, rate); return AMS_E_INVALID; }
    AMS_REQUIRE(panels && panel_elems >= (size_t)2 * N * Cc, "dw_project: panel scratch too small (need %zu)", (size_t)2 * N * Cc);
    uint16_t* hi = panels;
    uint16_t* lo = panels + (size_t)N * Cc;
    RUN(launch_split_weights(w_proj, N, 1, Cc, N, Cc, hi, lo, st));
    PwArgs a = pw_args(nullptr, (int64_t)B * H * W, Cc, Cc, w_proj, N, y, N);
    a.scale = scale_p; a.shift = shift_p; a.act = AMS_ACT_NONE;
    if (res) { a.res = res; a.ldr = N; }
    return launch_dw_project(e, B, H, W, Cc, w_dw, rate, scale_d, shift_d, AMS_ACT_RELU6, a, hi, lo, Cc, st);
}

int ams_k_block_fused(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e, const float* shift_e,
                      int32_t Cexp, const float* w_dw, int32_t stride, const float* scale_d, const float* shift_d, const float* w_proj, int32_t Cout,
                      const float* scale_p, const float* shift_p, int32_t residual, float* y, uint16_t* panels, size_t panel_elems, void* stream) {
    if (!block_fused_supported(Cin, Cexp, Cout, stride, 1, residual != 0)) { set_error("block_fused: unsupported shape"); return AMS_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* wparts = nullptr;
    const int64_t plane = (int64_t)Cexp * 32;
    if (panels) {                                  // three-part split of the expand weights into [part][Cexp][32]
        AMS_REQUIRE(Cin <= 32 && panel_elems >= (size_t)3 * plane, "block_fused: panel scratch too small (need %zu)", (size_t)3 * plane);
        RUN(launch_split_weights3(w_exp, Cexp, 1, Cin, Cexp, 32, panels, panels + plane, panels + 2 * plane, st));
        wparts = panels;
    }
    return launch_block_fused(x, B, H, W, Cin, w_exp, scale_e, shift_e, AMS_ACT_RELU6, Cexp, w_dw, stride, scale_d, shift_d, AMS_ACT_RELU6, w_proj,
                              scale_p, shift_p, AMS_ACT_NONE, Cout, residual != 0, y, st, nullptr, wparts, plane);
}

int ams_k_expand_dw(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                    const float* shift_e, int32_t Cexp, const float* w_dw, int32_t stride, int32_t rate, const float* scale_d,
                    const float* shift_d, float* y, void* stream) {
    if (!expand_dw_supported(Cin, Cexp, stride, rate)) { set_error("expand_dw: unsupported shape"); return AMS_E_INVALID; }
    return launch_expand_dw(x, B, H, W, Cin, w_exp, scale_e, shift_e, AMS_ACT_RELU6, Cexp, w_dw, stride, rate, scale_d, shift_d,
                            AMS_ACT_RELU6, y, (hipStream_t)stream);
}

int ams_k_expand_dw_stream(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                           const float* shift_e, int32_t Cexp, const float* w_dw, int32_t rate, const float* scale_d, const float* shift_d,
                           float* y, uint16_t* panels, size_t panel_elems, int32_t parts, int32_t presplit, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int stride = 1;
    if (rate < 0) { stride = -rate; rate = 1; }          // rate = -2 selects stride 2 (Cin <= 32 only)
    if (!expand_dw_stream_supported(Cin, Cexp, stride, rate) || (parts != 2 && parts != 3)) {
        set_error("expand_dw_stream: unsupported shape Cin=%d Cexp=%d rate=%d parts=%d", Cin, Cexp, rate, parts);
        return AMS_E_INVALID;
    }
    if (Cin <= 32)           // exact-f32 form: no panels
        return launch_expand_dw_stream(x, nullptr, 0, B, H, W, Cin, w_exp, nullptr, 0, 0, scale_e, shift_e, AMS_ACT_RELU6, Cexp, w_dw, stride, rate,
                                       scale_d, shift_d, AMS_ACT_RELU6, y, st);
    const size_t plane = (size_t)Cexp * Cin;
    AMS_REQUIRE(panels && panel_elems >= 3 * plane, "expand_dw_stream: panel scratch too small (need %zu)", 3 * plane);
    RUN(launch_split_weights3(w_exp, Cexp, 1, Cin, Cexp, Cin, panels, panels + plane, panels + 2 * plane, st));
    const uint16_t* xp = nullptr;
    const size_t xplane = (size_t)B * H * W * Cin;
    if (presplit) {          // the operand as bf16 parts, as a producing GEMM would leave it (PwArgs::ysplit)
        AMS_REQUIRE(panel_elems >= 3 * plane + 3 * xplane, "expand_dw_stream: panel scratch too small for the pre-split operand (need %zu)",
                    3 * plane + 3 * xplane);
        uint16_t* xq = panels + 3 * plane;
        RUN(launch_split_weights3(x, 1, Cin, Cin, (int)((int64_t)B * H * W), Cin, xq, xq + xplane, xq + 2 * xplane, st));
        xp = xq;
    }
    if (presplit == 2)       // the weight-register form (k_xdw_wreg.hip)
        return launch_expand_dw_wreg(xp, (int64_t)xplane, B, H, W, Cin, panels, (int64_t)plane, parts, scale_e, shift_e, AMS_ACT_RELU6, Cexp, w_dw,
                                     rate, scale_d, shift_d, AMS_ACT_RELU6, y, st);
    return launch_expand_dw_stream(x, xp, (int64_t)xplane, B, H, W, Cin, nullptr, panels, (int64_t)plane, parts, scale_e, shift_e, AMS_ACT_RELU6,
                                   Cexp, w_dw, 1, rate, scale_d, shift_d, AMS_ACT_RELU6, y, st);
}

int ams_k_global_mean(const float* x, int32_t B, int64_t HW, int32_t C, float* y, float* scratch, size_t scratch_floats,
                      void* stream) {
    AMS_REQUIRE(scratch && scratch_floats >= image_colsum_scratch(B, C), "global_mean: scratch too small (need %zu floats)",
                image_colsum_scratch(B, C));
    return launch_global_mean(x, B, HW, C, y, scratch, (hipStream_t)stream);
}
size_t ams_k_global_mean_scratch(int32_t B, int32_t C) { return image_colsum_scratch(B, C); }

int ams_k_upsample_argmax(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host, int32_t K,
                          int32_t H, int32_t W, const uint8_t* teacher, int32_t* labels_out, int64_t* conf_mat, double* loss,
                          void* stream) {
    return launch_upsample_argmax(logits, NC, B, h, w, class_idx_host, K, H, W, teacher, NC, labels_out, conf_mat, loss,
                                  (hipStream_t)stream);
}

int ams_k_ce_grad(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host, int32_t K, int32_t H,
                  int32_t W, const uint8_t* teacher, const double* loss_and_count_dev, float* dlogits, void* stream) {
    return launch_ce_grad(logits, NC, B, h, w, class_idx_host, K, H, W, teacher, NC, loss_and_count_dev, dlogits, NC,
                          (hipStream_t)stream);
}

size_t ams_k_ce_loss_grad_scratch(int32_t B, int32_t h, int32_t w, int32_t K) { return ce_loss_grad_scratch(B, h, w, K); }

int ams_k_ce_loss_grad(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host, int32_t K, int32_t H,
                       int32_t W, const uint8_t* teacher, double* loss_dev, float* dlogits, float* scratch, size_t scratch_floats, void* stream) {
    AMS_REQUIRE(ce_loss_grad_supported(w, W), "ce_loss_grad: %d output columns on %d source columns is outside the one-pass kernel", W, w);
    AMS_REQUIRE(scratch && scratch_floats >= ce_loss_grad_scratch(B, h, w, K), "ce_loss_grad: scratch too small (need %zu floats)",
                ce_loss_grad_scratch(B, h, w, K));
    hipStream_t st = (hipStream_t)stream;
    RUN(launch_ce_loss_grad(logits, NC, B, h, w, class_idx_host, K, H, W, teacher, NC, loss_dev, scratch, st));
    return launch_ce_combine(B, h, w, class_idx_host, K, NC, loss_dev, scratch, dlogits, NC, st);
}

size_t ams_k_pointwise_wgrad_scratch(int64_t M, int32_t K, int32_t N) { return pointwise_wgrad_scratch(M, K, N); }

int ams_k_pointwise_wgrad_split(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, float* dw, float* scratch,
                                size_t scratch_floats, void* stream) {
    if (!pointwise_wgrad_x6_applies(M, K, N, K, N)) { set_error("pointwise_wgrad_split: shape M=%lld K=%d N=%d not supported", (long long)M, K, N); return AMS_E_INVALID; }
    WgArgs a;
    a.x = x; a.ldx = K; a.K = K; a.dy = dy; a.ldy = N; a.N = N; a.M = M; a.dw = dw; a.scratch = scratch; a.scratch_floats = scratch_floats; a.allow_split = 1;
    return launch_pointwise_wgrad(a, (hipStream_t)stream);
}

int ams_k_pointwise_wgrad(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, float* dw, float* scratch,
                          size_t scratch_floats, void* stream) {
    WgArgs a;
    a.x = x; a.ldx = K; a.K = K; a.dy = dy; a.ldy = N; a.N = N; a.M = M; a.dw = dw; a.scratch = scratch; a.scratch_floats = scratch_floats; a.allow_split = 0;
    return launch_pointwise_wgrad(a, (hipStream_t)stream);
}

int ams_k_depthwise3x3_dgrad(const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t stride, int32_t rate,
                             float* dx, void* stream) {
    return launch_depthwise_dgrad(dy, B, H, W, C, w, stride, rate, dx, (hipStream_t)stream);
}

int ams_k_depthwise3x3_wgrad(const float* x, const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride, int32_t rate,
                             float* dw, float* scratch, size_t scratch_floats, void* stream) {
    return launch_depthwise_wgrad(x, dy, B, H, W, C, stride, rate, dw, scratch, scratch_floats, (hipStream_t)stream);
}

int ams_k_pointwise_red(const float* x, int64_t M, int32_t K, const float* w, int32_t N, int32_t trans_w, int32_t split, int32_t mode,
                        const float* center, const float* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                        int32_t act, const float* res, float* y, float* part, size_t part_floats, int32_t* rows_out, uint16_t* panels,
                        size_t panel_elems, void* stream) {
    AMS_REQUIRE(x && w && y && part && rows_out && (mode == 1 || mode == 2), "pointwise_red: bad arguments");
    AMS_REQUIRE(mode == 1 || (z && scale && shift && mean && rstd), "pointwise_red: mode 2 needs z, scale, shift, mean, rstd");
    AMS_REQUIRE(part_floats >= red_rows_bound(M) * 2 * (size_t)N, "pointwise_red: partial-row buffer too small");
    hipStream_t st = (hipStream_t)stream;
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    if (trans_w) { a.w_sk = 1; a.w_sn = K; }
    a.res = res; a.ldr = N;
    a.red_mode = mode; a.red_center = center; a.red_z = z; a.red_scale = scale; a.red_shift = shift; a.red_mean = mean; a.red_rstd = rstd;
    a.red_act = act; a.red_part = part;
    int rows = 0;
    a.red_rows_out = &rows;
    int rc;
    if (split) {
        const int Kp = (K + 31) / 32 * 32;
        const size_t plane = (size_t)N * Kp;
        AMS_REQUIRE(panels && panel_elems >= 3 * plane && K % 8 == 0, "pointwise_red: panel scratch too small (need %zu) or K %% 8", 3 * plane);
        RUN(launch_split_weights3(w, a.w_sk, a.w_sn, K, N, Kp, panels, panels + plane, panels + 2 * plane, st));
        rc = launch_pointwise_split3(a, panels, panels + plane, panels + 2 * plane, Kp, st);
    } else {
        rc = launch_pointwise(a, st);
    }
    *rows_out = rows;
    return rc;
}

size_t ams_k_depthwise3x3_fwd_bn_scratch(int32_t B, int32_t H, int32_t W, int32_t C, int32_t rate) { return depthwise_fwd_bn_scratch(B, H, W, C, rate); }
int ams_k_depthwise3x3_fwd_bn(const float* ze, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t rate, const float* scale,
                              const float* shift, int32_t act, const float* center, float* zd, float* scratch, size_t scratch_floats,
                              int32_t* rows_out, void* stream) {
    AMS_REQUIRE(ze && w && scale && shift && zd && scratch && rows_out, "depthwise3x3_fwd_bn: null pointer");
    AMS_REQUIRE(scratch_floats >= depthwise_fwd_bn_scratch(B, H, W, C, rate), "depthwise3x3_fwd_bn: scratch too small");
    int rows = 0;
    int rc = launch_depthwise_fwd_bn(ze, B, H, W, C, w, rate, scale, shift, act, center, zd, scratch, &rows, (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}
size_t ams_k_depthwise3x3_dgrad_bn_scratch(int32_t B, int32_t H, int32_t W, int32_t C) { return depthwise_dgrad_bn_scratch(B, H, W, C); }
int ams_k_depthwise3x3_dgrad_bn(const float* dz, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t rate, const float* z_prev,
                                const float* scale, const float* shift, int32_t act, const float* mean, const float* rstd, float* out,
                                float* scratch, size_t scratch_floats, int32_t* rows_out, void* stream) {
    AMS_REQUIRE(dz && w && z_prev && scale && shift && mean && rstd && out && scratch && rows_out, "depthwise3x3_dgrad_bn: null pointer");
    AMS_REQUIRE(rate == 1 || rate == 2, "depthwise3x3_dgrad_bn: rate %d", rate);
    AMS_REQUIRE(scratch_floats >= depthwise_dgrad_bn_scratch(B, H, W, C), "depthwise3x3_dgrad_bn: scratch too small");
    int rows = 0;
    int rc = launch_depthwise_dgrad_bn(dz, B, H, W, C, w, rate, z_prev, scale, shift, act, mean, rstd, out, scratch, &rows, (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}

size_t ams_k_xdw_train_scratch(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cexp) { return xdw_train_scratch(B, H, W, Cin, Cexp); }
int ams_k_xdw_fwd_stats(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, int32_t Cexp, const float* center,
                        float* scratch, size_t scratch_floats, int32_t* rows_out, int64_t* stride_out, void* stream) {
    AMS_REQUIRE(x && w_exp && scratch && rows_out && stride_out, "xdw_fwd_stats: null pointer");
    AMS_REQUIRE(xdw_train_supported(Cin, Cexp, 1, 1) && scratch_floats >= xdw_train_scratch(B, H, W, Cin, Cexp), "xdw_fwd_stats: shape or scratch");
    int rows = 0;
    int rc = launch_xdw_fwd_stats(x, B, H, W, Cin, w_exp, Cexp, center, scratch, &rows, stride_out, (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}
int ams_k_xdw_bwd_reduce(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, int32_t Cexp, const float* sc_e,
                         const float* sh_e, const float* mean_e, const float* rstd_e, int32_t act_e, const float* w_dw, int32_t stride,
                         const float* dz_d, float* scratch, size_t scratch_floats, int32_t* rows_out, int64_t* stride_out, void* stream) {
    AMS_REQUIRE(x && w_exp && sc_e && sh_e && mean_e && rstd_e && w_dw && dz_d && scratch && rows_out && stride_out, "xdw_bwd_reduce: null pointer");
    AMS_REQUIRE(xdw_train_supported(Cin, Cexp, stride, 1) && scratch_floats >= xdw_train_scratch(B, H, W, Cin, Cexp), "xdw_bwd_reduce: shape or scratch");
    int rows = 0;
    int rc = launch_xdw_bwd_reduce(x, B, H, W, Cin, w_exp, Cexp, sc_e, sh_e, mean_e, rstd_e, act_e, w_dw, stride, dz_d, scratch, &rows, stride_out,
                                   (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}
int ams_k_xdw_bwd_dx(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, int32_t Cexp, const float* sc_e,
                     const float* sh_e, int32_t act_e, const float* w_dw, int32_t stride, const float* dz_d, const float* cA, const float* cB,
                     const float* cC, const float* res, float* dx, void* stream) {
    AMS_REQUIRE(x && w_exp && sc_e && sh_e && w_dw && dz_d && cA && cB && cC && dx, "xdw_bwd_dx: null pointer");
    return launch_xdw_bwd_dx(x, B, H, W, Cin, w_exp, Cexp, sc_e, sh_e, act_e, w_dw, stride, dz_d, cA, cB, cC, res, dx, (hipStream_t)stream);
}
size_t ams_k_xdw_stem_scratch(int32_t B, int32_t fH, int32_t fW) { return xdw_stem_scratch(B, fH, fW); }
int ams_k_xdw_bwd_reduce_stem(const void* frames, int32_t frames_dtype, int32_t B, int32_t fH, int32_t fW, float pixel_scale, const float* w_stem,
                              const float* sc_e, const float* sh_e, const float* mean_e, const float* rstd_e, int32_t act_e, const float* w_dw,
                              const float* dz_d, float* scratch, size_t scratch_floats, int32_t* rows_out, int64_t* stride_out, void* stream) {
    AMS_REQUIRE(frames && w_stem && sc_e && sh_e && mean_e && rstd_e && w_dw && dz_d && scratch && rows_out && stride_out, "xdw_bwd_reduce_stem: null pointer");
    AMS_REQUIRE(scratch_floats >= xdw_stem_scratch(B, fH, fW), "xdw_bwd_reduce_stem: scratch too small");
    int rows = 0;
    int rc = launch_xdw_bwd_reduce_stem(frames, frames_dtype, B, fH, fW, pixel_scale, w_stem, sc_e, sh_e, mean_e, rstd_e, act_e, w_dw, dz_d, scratch,
                                        &rows, stride_out, (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}
int ams_k_xdw_dwe(const float* G1, const float* xx_g0, int32_t Cin, int32_t Cexp, const float* w_exp, const float* cA, const float* cB,
                  const float* cC, float* dw_exp, void* stream) {
    AMS_REQUIRE(G1 && xx_g0 && w_exp && cA && cB && cC && dw_exp, "xdw_dwe: null pointer");
    return launch_xdw_dwe(G1, xx_g0, Cin, Cexp, w_exp, cA, cB, cC, dw_exp, (hipStream_t)stream);
}

int ams_k_adam(float* params, const float* grads, float* m, float* v, const uint8_t* mask, int64_t n, float lr_t, float beta1,
               float beta2, float eps, void* stream) {
    return launch_adam(params, grads, m, v, mask, n, lr_t, beta1, beta2, eps, (hipStream_t)stream);
}

}  // extern "C"
