// The student engine, part 1 of 4: error text, the per-launch profiler, and the plan of one student over a caller-provided device
// arena (layer geometry, arena layout, live weight-panel table).
//   frozen inference  : stem -> 17 inverted-residual blocks -> head -> fused upsample/argmax(/metrics)      (engine_forward.hip)
//   live forward      : same graph with training-mode BN (batch statistics), activations kept for backward (engine_forward.hip)
//   train step        : live forward -> CE -> backward -> BN moving averages -> Adam (+ coordinate-descent mask) (engine_backward.hip)
// Replaces tf.Session.run over the graph built by create_student_v3 (reference utils/graph_utils.py:338-533).
#include "engine.hpp"

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

void prof_begin(ams_student* s, hipStream_t st) {
    g_kname = nullptr;
    if (!s->prof.on) return;
    s->prof_e0 = s->prof.get();
    (void)hipEventRecord(s->prof_e0, st);
}
void prof_end(ams_student* s, hipStream_t st, int layer, double bytes) {
    if (!s->prof.on) return;
    hipEvent_t e1 = s->prof.get();
    (void)hipEventRecord(e1, st);
    s->prof.recs.push_back(ProfRec{g_kname ? g_kname : "?", layer, bytes, s->prof_flops, s->prof_flops_x6, s->prof_e0, e1});
    s->prof_flops = 0.0;
    s->prof_flops_x6 = 0.0;
}

// the live weight panels: every 1x1 layer a split GEMM may run on, forward and input-gradient orientation.  p0 holds the element
// offset inside tp_panels until the arena is known (create turns it into a pointer).
static void plan_train_panels(ams_student* s) {
    s->tp_jobs.clear();
    s->tp_elems = 0;
    s->tp_blocks = 0;
    if (!s->cfg.trainable) return;
    auto add = [&](int64_t w_off, int64_t sk, int64_t sn, int K, int N, int f16) {
        if (K < 32 || K % 8 != 0) return;                  // split_pays() never takes these
        SplitJob j;
        memset(&j, 0, sizeof(j));
        j.w = (const float*)(uintptr_t)w_off;              // offset for now
        j.sk = sk; j.sn = sn; j.K = K; j.N = N; j.Kp = (K + 31) / 32 * 32;
        j.plane = (int64_t)N * j.Kp;
        j.p0 = (uint16_t*)(uintptr_t)s->tp_elems;
        j.first_block = s->tp_blocks;
        j.f16 = f16;
        s->tp_elems += (f16 ? 5 : 3) * (size_t)j.plane;
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
        add(w_off, N, 1, K, N, 1);                          // forward: element (k, n) at w[k*N + n]; also as two fp16 parts
        if (l.d.role != AMS_ROLE_LOGITS) add(w_off, 1, N, N, K, 0);   // input gradient: B operand (k' = n, n' = k) = w[k][n]
    }
}

int student_layout(ams_student* s, void* arena, size_t* bytes_out) {
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
            l.whf = l.whf_mem = cv.take<uint16_t>(2 * plane);
        }
    }
    if (s->n_backbone >= 3 && s->L[1].d.cout == 32 && s->L[2].d.role == AMS_ROLE_DEPTHWISE) {
        LayerRt& l1 = s->L[1];
        l1.blk_vecs = cv.take<float>(13 * 32);
        l1.Kp = 32;                                        // stem weights as three bf16 parts [32][32] for the first-block kernel's split form
        l1.whi = cv.take<uint16_t>(3 * 32 * 32);
        l1.wlo = l1.whi ? l1.whi + 32 * 32 : nullptr;
        l1.wlo3 = l1.whi ? l1.whi + 2 * 32 * 32 : nullptr;
        l1.whf = l1.whf_mem = cv.take<uint16_t>(2 * 32 * 32);           // ... and as two fp16 parts
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
                if (i >= 3 && expand_dw_supported(s->L[i - 1].d.cin, l.d.cin, l.d.stride, l.d.rate)) {      // statistics rows of the recompute blocks' forward
                    const size_t n0 = expand_dw_stats_scratch(B, l.Hin, l.Win, l.d.cin, l.d.stride);
                    if (n0 > need) need = n0;
                }
                if (l.d.stride == 1 && l.d.cin <= 1024) {       // the one-kernel forms of the blocks that keep their tensors
                    const size_t n1 = depthwise_dgrad_bn_scratch(B, l.Hin, l.Win, l.d.cin), n2 = depthwise_fwd_bn_scratch(B, l.Hin, l.Win, l.d.cin, l.d.rate);
                    const size_t n3 = depthwise_fwd_bn2_scratch(B, l.Hin, l.Win, l.d.cin, l.d.rate);
                    if (n1 > need) need = n1;
                    if (n2 > need) need = n2;
                    if (n3 != (size_t)-1 && n3 > need) need = n3;
                }
            } else if (l.d.role == AMS_ROLE_STEM) need = pointwise_wgrad_scratch(M, 27, l.d.cout);
            else {
                need = pointwise_wgrad_scratch(M, l.d.cin, l.d.cout);
                // partial rows of the column reductions fused into this layer's forward GEMM (N = cout) and input-gradient GEMM (N = cin)
                const size_t red = red_rows_bound(M) * 2 * (size_t)(l.d.cin > l.d.cout ? l.d.cin : l.d.cout);
                if (red > need) need = red;
            }
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
                if (need != (size_t)-1 && need > xt) xt = need;          // (size_t)-1: the map is too large for the kernels' 32-bit offsets — that block keeps the layer-by-layer step
            }
        }
        if (s->n_backbone >= 3 && s->L[1].d.cout == 32 && s->L[2].d.role == AMS_ROLE_DEPTHWISE && s->L[2].d.stride == 1 && s->L[2].d.rate == 1 &&
            xdw_stem_scratch(B, c.height, c.width) != (size_t)-1 && xdw_stem_scratch(B, c.height, c.width) > xt)
            xt = xdw_stem_scratch(B, c.height, c.width);
        s->xt_floats = xt;
        s->xt_scratch = cv.take<float>(xt);
        for (int i = 2; i + 1 <= s->n_backbone; ++i) {
            LayerRt& l = s->L[i];
            if (l.d.role == AMS_ROLE_EXPAND && s->L[i + 1].d.role == AMS_ROLE_DEPTHWISE &&
                xdw_train_supported(l.d.cin, l.d.cout, s->L[i + 1].d.stride, s->L[i + 1].d.rate)) {
                const int KP = (l.d.cin + 15) / 16 * 16;
                l.xx_g0 = cv.take<float>((size_t)KP * KP + KP);
                l.xx64 = cv.take<double>((size_t)KP * KP + KP);
                const size_t nd = xx_stats_scratch_doubles((int64_t)B * l.px_in, l.d.cin);
                if (nd > s->xx_scratch_doubles) s->xx_scratch_doubles = nd;
            }
        }
        s->xx_scratch = s->xx_scratch_doubles ? cv.take<double>(s->xx_scratch_doubles) : nullptr;
        s->vec_ones = cv.take<float>(1024);
        s->vec_zeros = cv.take<float>(1024);
        s->vec_inv_hw = cv.take<float>(1024);
        s->d_img_bias = cv.take<float>((size_t)B * aspp_c);
        s->d_pool_a = cv.take<float>((size_t)B * aspp_c);
        s->d_pool_z = cv.take<float>((size_t)B * aspp_c);
        s->d_pooled = cv.take<float>((size_t)B * head_cin);
        s->im2col = cv.take<float>((size_t)B * s->L[1].px_out * 32);
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
            if (l.d.role == AMS_ROLE_DEPTHWISE && l.d.stride == 1 && i >= 3 && l.d.cin <= 1024) {
                const size_t n1 = depthwise_dgrad_bn_scratch(B, l.Hin, l.Win, l.d.cin), n2 = depthwise_dgrad_bn2_scratch(B, l.Hin, l.Win, l.d.cin, l.d.rate);
                l.dw_rows_floats = n2 != (size_t)-1 && n2 > n1 ? n2 : n1;
                l.dw_rows = cv.take<float>(l.dw_rows_floats);
            }
        }
    }
    *bytes_out = (cv.off + 255) & ~(size_t)255;
    return AMS_OK;
}

int student_build(ams_student* s, const ams_student_config* cfg, const ams_layer_desc* layers) {
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

}  // namespace ams
