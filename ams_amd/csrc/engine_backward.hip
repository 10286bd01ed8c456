// The student engine, part 3 of 4: loss, backward and update of the fine-tune step (reference SemanticNetwork.py:253-260 _train,
// utils/graph_utils.py:403-408 loss, :457-496 optimizer / BN update / masked assignment).
#include "engine.hpp"
#include <functional>
#include <vector>

namespace ams {

// =======================================================================================================
// backward + update
// =======================================================================================================
// BN backward of layer l given da (gradient wrt the layer's activated output): writes dz (default: in place over da), dgamma/dbeta into grads
static int bn_backward(ams_student* s, LayerRt& l, const float* da, int64_t M_local, double n_global, const SyncCtx* sc,
                       hipStream_t st, float* dz = nullptr) {
    if (!dz) dz = const_cast<float*>(da);
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

// xa: the layer whose BN + activation the x operand still needs (x = its raw output z): WgArgs x_mode 1
static int pw_wgrad(ams_student* s, const float* x, int ldx, int K, const float* dy, int ldy, int N, int64_t M, float* dw,
                    hipStream_t st, float* scratch = nullptr, const LayerRt* xa = nullptr) {
    WgArgs a;
    a.x = x; a.ldx = ldx; a.K = K; a.dy = dy; a.ldy = ldy; a.N = N; a.M = M; a.dw = dw;
    a.scratch = scratch ? scratch : s->scratch; a.scratch_floats = s->scratch_floats;
    a.allow_split = s->matmul_mode != AMS_MATMUL_F32;
    if (xa) { a.x = xa->z; a.x_mode = 1; a.x_act = xa->d.act; a.x_v0 = xa->scale; a.x_v1 = xa->shift; }
    RUNK(0, 4.0 * ((double)M * (K + N) + (double)K * N), launch_pointwise_wgrad(a, st));
    return AMS_OK;
}

// dx[M,K] = dy[M,N] @ w[K,N]^T (+ extras through the epilogue)
static PwArgs dgrad_args(const float* dy, int64_t M, int N, int ldy, const float* w, int K, float* dx) {
    PwArgs a = pw_args(dy, M, N, ldy, w, K, dx, K);
    a.w_sk = 1; a.w_sn = N;          // B operand (k' = n, n' = k) = w[k][n]
    return a;
}

int backward(ams_student* s, const void* frames, int dtype, const uint8_t* teacher, int B, int global_B, const SyncCtx* sc,
                    hipStream_t st) {
    const ams_student_config& c = s->cfg;
    const float* P = s->params;
    float* G = s->grads;
    LayerRt& lp = s->L[s->iPool]; LayerRt& la = s->L[s->iAspp]; LayerRt& lc = s->L[s->iProj]; LayerRt& ll = s->L[s->iLogits];
    const int64_t HW = (int64_t)s->h * s->w, M = (int64_t)B * HW;
    const double nHW = (double)global_B * HW;
    const int NC = c.num_classes;
    // d loss / d logits (already divided by the global number of valid pixels).  Without one valid pixel the reference's loss is NaN
    // (utils/graph_utils.py:408: reduce_mean of an empty boolean_mask) but its gradients are ZERO: the backward of reduce_mean over a [0]-shaped
    // tensor is an empty tensor, and boolean_mask's gather gradient densifies it to zeros — the weights survive such a batch.  Default;
    // AMS_OPT_NAN_GRADS = 1 writes NaN gradients instead (a loud failure for callers that prefer one).
    const float empty_val = s->nan_grads ? __builtin_nanf("") : 0.f;
    if (ce_loss_grad_supported(s->w, c.width))           // second pass of the one-pass loss kernel (train_step_impl ran the first)
        RUNK(0, 0.0, launch_ce_combine(B, s->h, s->w, c.class_indices, c.n_selected, NC, s->loss_buf, s->ce_scratch, s->dlogits, 32, st, empty_val));
    else
        RUNK(0, 0.0, launch_ce_grad(s->logits, 32, B, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher, NC, s->loss_buf,
                                    s->dlogits, 32, st, empty_val));
    // A layer's weight gradient only feeds the optimizer: it runs on the side stream while the main stream goes on with the input gradient
    // and the next layer's BN backward (many of these kernels are latency-bound at 8 frames and share the chip well).  Nothing the side
    // stream reads is overwritten inside the step (every dz lives in per-layer memory — in the head too: in place over da), so the main
    // stream never waits for it before the final join.
    const bool overlap = s->overlap_wgrad && !s->prof.on && s->scratch2;
    if (overlap && !s->side) RUN(create_side_stream(&s->side));
    if (overlap && !s->ev_xt) {
        if (!s->ev_fork) RUN(create_sync_event(&s->ev_fork));
        RUN(create_sync_event(&s->ev_xt));
    }
    // Weight gradients are queued and handed to the side stream AMS_OPT_WGRAD_FORK_EVERY at a time.  Every hand-over is an event on the main
    // stream (6-8 us gaps in profiles/r04_train_timeline.txt, ~55 a step) and nothing a queued job reads is overwritten inside the step, so a
    // job may run any time after its operands exist — but batching measured SLOWER (7.98 -> 8.07 / 8.2 / 8.3 / 8.6 ms for 2 / 4 / 8 / 16): a
    // weight gradient that runs beside the input-gradient GEMM of the same dz shares its cache lines.  Default 1 = hand over at once.
    // Without the side stream a job runs where it is queued.
    using WgJob = std::function<int(hipStream_t, float*)>;
    std::vector<WgJob> pend;
    // AMS_OPT_OVERLAP_WGRAD = 3: hand-overs alternate between the two side streams (each has its own reduction scratch), so a weight gradient
    // starts at its hand-over even while the one before it is still running
    const bool alt = overlap && s->overlap_wgrad >= 3 && s->scratch3;
    if (alt && !s->side2) AMS_CHECK_HIP(hipStreamCreateWithFlags(&s->side2, hipStreamNonBlocking));
    bool alt_second = false;
    hipStream_t last_wst = st;
    auto flush_wgrads = [&]() -> int {
        if (pend.empty()) return AMS_OK;
        hipStream_t wst = st;
        float* wscr = s->scratch;
        if (overlap) {
            wst = alt && alt_second ? s->side2 : s->side;
            wscr = alt && alt_second ? s->scratch3 : s->scratch2;
            alt_second = !alt_second;
            AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
            AMS_CHECK_HIP(hipStreamWaitEvent(wst, s->ev_fork, 0));
        }
        last_wst = wst;
        for (auto& j : pend) RUN(j(wst, wscr));
        pend.clear();
        return AMS_OK;
    };
    const int fork_every = overlap ? (s->wgrad_fork_every > 1 ? s->wgrad_fork_every : 1) : 1;
    auto queue_wgrad = [&](WgJob j) -> int {
        pend.push_back(std::move(j));
        return (int)pend.size() >= fork_every ? flush_wgrads() : AMS_OK;
    };
    // logits layer: bias, weights, input gradient
    RUN(launch_colsum(s->dlogits, M, 32, 32, s->tmp_c, s->scratch, st));
    RUN(launch_copy(G + ll.d.gamma_off, s->tmp_c, NC, st));
    RUN(queue_wgrad([=, &lc, &ll](hipStream_t ws, float* wscr) { return pw_wgrad(s, lc.a, lc.d.cout, lc.d.cout, s->dlogits, 32, NC, M, G + ll.d.w_off, ws, wscr); }));
    {
        PwArgs a = dgrad_args(s->dlogits, M, 32, 32, P + ll.d.w_off, ll.d.cin, lc.da);
        a.Kw = NC; a.w_sn = NC;        // w is [cin][NC]; dlogits columns >= NC are zero
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // concat_projection (dz in place over da, like every backbone layer: the side stream may still be reading it when aspp0's dz is formed)
    float* dz_c = lc.da;
    RUN(bn_backward(s, lc, lc.da, M, nHW, sc, st, dz_c));
    const float* Wc_top = P + lc.d.w_off;
    const float* Wc_bot = P + lc.d.w_off + (int64_t)lp.d.cout * lc.d.cout;
    RUN(queue_wgrad([=, &la, &lc, &lp](hipStream_t ws, float* wscr) {
        return pw_wgrad(s, la.a, la.d.cout, la.d.cout, dz_c, lc.d.cout, lc.d.cout, M, G + lc.d.w_off + (int64_t)lp.d.cout * lc.d.cout, ws, wscr);
    }));
    {
        PwArgs a = dgrad_args(dz_c, M, lc.d.cout, lc.d.cout, Wc_bot, la.d.cout, la.da);
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // pool branch: the per-image bias collects the column sums of dz_proj
    RUN(launch_image_colsum(dz_c, B, HW, lc.d.cout, lc.d.cout, s->d_img_bias, s->scratch, st));
    RUN(queue_wgrad([=, &lp, &lc](hipStream_t ws, float* wscr) { return pw_wgrad(s, lp.a, lp.d.cout, lp.d.cout, s->d_img_bias, lc.d.cout, lc.d.cout, B, G + lc.d.w_off, ws, wscr); }));
    {
        PwArgs a = dgrad_args(s->d_img_bias, B, lc.d.cout, lc.d.cout, Wc_top, lp.d.cout, s->d_pool_a);
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    {   // BN (over the batch) + ReLU of the pool branch; its dz goes to d_pool_z
        RUN(launch_bn_bwd_reduce(s->d_pool_a, lp.z, B, lp.d.cout, lp.scale, lp.shift, lp.d.act, lp.mean, lp.rstd, lp.bsums, s->scratch, st));
        RUN(launch_bn_param_grads(lp.bsums, lp.d.cout, G + lp.d.gamma_off, G + lp.d.beta_off, st));
        RUN(sync_doubles(sc, lp.bsums, 2 * (size_t)lp.d.cout, st));
        RUN(launch_bn_bwd_coef(lp.bsums, (double)global_B, lp.d.cout, P + lp.d.gamma_off, lp.mean, lp.rstd, lp.cA, lp.cB, lp.cC,
                               nullptr, nullptr, st));
        RUN(launch_bn_bwd_apply(s->d_pool_a, lp.z, B, lp.d.cout, lp.scale, lp.shift, lp.d.act, lp.cA, lp.cB, lp.cC, s->d_pool_z, st));
        RUN(queue_wgrad([=, &lp](hipStream_t ws, float* wscr) { return pw_wgrad(s, s->pooled, lp.d.cin, lp.d.cin, s->d_pool_z, lp.d.cout, lp.d.cout, B, G + lp.d.w_off, ws, wscr); }));
        PwArgs a = dgrad_args(s->d_pool_z, B, lp.d.cout, lp.d.cout, P + lp.d.w_off, lp.d.cin, s->d_pooled);
        a.scale = s->vec_inv_hw;        // d mean / d feat = 1/HW, applied as a uniform scale (constants uploaded at create)
        a.shift = s->vec_zeros;
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // aspp0: its input gradient also receives the pooled gradient, broadcast over the image
    LayerRt& lf = s->L[s->n_backbone];
    float* dz_a = la.da;
    RUN(bn_backward(s, la, la.da, M, nHW, sc, st, dz_a));
    RUN(queue_wgrad([=, &lf, &la](hipStream_t ws, float* wscr) { return pw_wgrad(s, lf.a, la.d.cin, la.d.cin, dz_a, la.d.cout, la.d.cout, M, G + la.d.w_off, ws, wscr); }));
    {
        PwArgs a = dgrad_args(dz_a, M, la.d.cout, la.d.cout, P + la.d.w_off, la.d.cin, lf.da);
        a.img_bias = s->d_pooled; a.rows_per_img = HW;
        RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
    }
    // backbone, last layer to first
    const bool three = overlap && s->overlap_wgrad >= 2 && s->scratch3;      // depthwise weight gradients on a third stream (3: see `alt`)
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
        // stride-16 depthwise layer whose masked gradient and sums came out of the project layer's input-gradient GEMM: its apply pass
        // (dz = A dy + B + C z) is formed inside the depthwise backward kernel below, on the way into that kernel's LDS ring (k_dw_train.hip)
        const bool fold_apply = s->fuse_dgrad_bn >= 3 && fused_rows > 0 && !fused_dw && l.d.role == AMS_ROLE_DEPTHWISE && l.d.cin % 64 == 0 && dw_fused_train(s, i, B) &&
                                l.dw_rows && depthwise_dgrad_bn2_scratch(B, l.Hin, l.Win, l.d.cin, l.d.rate) <= l.dw_rows_floats;
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
            if (!fold_apply) RUNK(0, 12.0 * Mo * l.d.cout, launch_bn_bwd_apply(l.da, l.z, Mo, l.d.cout, l.scale, l.shift, AMS_ACT_NONE, l.cA, l.cB, l.cC, dz, st));
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
                // the stride-16 blocks are behind us: their deferred reductions (80 MB of partial rows) run on the side stream under the early
                // blocks (queued: handed over with this block's own reductions below)
                ReduceJobs dj = deferred;
                pend.push_back([dj](hipStream_t xs, float*) mutable { return launch_reduce_batch(dj, xs); });
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
            // optimizer: on the side stream, behind the coefficients.  The hand-over event is recorded AFTER the dx pass was queued (flush_wgrads
            // below), so these reductions start when dx is done, not beside it: ordering is what matters here (xt_scratch and cA / cB / cC are
            // final either way), and recording before the dx launch measured no faster
            const int KP = (le.d.cin + 15) / 16 * 16;
            const int64_t n_dw = 9 * (int64_t)le.d.cout, n_g = (int64_t)KP * le.d.cout;
            float* reduced = s->xt_scratch + (int64_t)rows * stride;
            {
                float* xt = s->xt_scratch;
                const int cin_e = le.d.cin, cout_e = le.d.cout;
                const float* g0 = le.xx_g0;
                const float *we = P + le.d.w_off, *cA = le.cA, *cB = le.cB, *cC = le.cC;
                float *gdw = G + l.d.w_off, *gwe = G + le.d.w_off;
                pend.push_back([=](hipStream_t xs, float*) -> int {
                    RUN(launch_reduce_splits(xt + 2 * (int64_t)cout_e, rows, n_dw, gdw, xs, stride));
                    RUN(launch_reduce_splits(xt + 11 * (int64_t)cout_e, rows, n_g, reduced, xs, stride));
                    RUN(launch_xdw_dwe(reduced, g0, cin_e, cout_e, we, cA, cB, cC, gwe, xs));
                    return AMS_OK;
                });
                RUN(flush_wgrads());                  // with whatever weight gradients are queued: one event for all of them
            }
            if (overlap) { AMS_CHECK_HIP(hipEventRecord(s->ev_xt, last_wst)); xt_pending = true; }
            --i;                                       // the expand layer is done
            continue;
        }
        if (i == 2 && stem_fused_train(s)) {
            RUN(flush_wgrads());                      // the last weight gradients start under the stem kernels, not behind them
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
            if (fold_apply)
                RUNK(i, dw_bytes(l, B) + 8.0 * B * l.px_in * l.d.cin,
                     launch_depthwise_dgrad_bn2(l.da, l.z, l.cA, l.cB, l.cC, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, prev.z, prev.scale,
                                                prev.shift, prev.d.act, prev.mean, prev.rstd, prev.da, l.dw_rows, &fused_rows, st));
            else
            RUNK(i, dw_bytes(l, B) + 4.0 * B * l.px_in * l.d.cin,
                 launch_depthwise_dgrad_bn(dz, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, prev.z, prev.scale, prev.shift, prev.d.act, prev.mean,
                                           prev.rstd, prev.da, l.dw_rows ? l.dw_rows : s->scratch, &fused_rows, st));
            fused_stride = 11 * (int64_t)l.d.cin;
            fused_dw = true;
            fused_buf = l.dw_rows ? l.dw_rows : s->scratch;
            continue;
        }
        if (l.d.role == AMS_ROLE_DEPTHWISE) {
            hipStream_t wst = st;
            float* wscratch = s->scratch;
            if (overlap) {
                wst = three ? s->side2 : s->side;
                wscratch = three ? s->scratch3 : s->scratch2;
                AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
                AMS_CHECK_HIP(hipStreamWaitEvent(wst, s->ev_fork, 0));
            }
            if (overlap) RUN(launch_depthwise_wgrad(prev.a, dz, B, l.Hin, l.Win, l.d.cin, l.d.stride, l.d.rate, G + l.d.w_off, wscratch, s->scratch_floats, wst));
            else RUNK(i, dw_bytes(l, B), launch_depthwise_wgrad(prev.a, dz, B, l.Hin, l.Win, l.d.cin, l.d.stride, l.d.rate, G + l.d.w_off,
                                                                wscratch, s->scratch_floats, wst));
        } else {
            const LayerRt* xa = (l.d.role == AMS_ROLE_PROJECT && operand_bn_act(s, i - 1)) ? &prev : nullptr;
            const float* xop = prev.a;
            const int cin = l.d.cin, cout = l.d.cout;
            float* dwp = G + l.d.w_off;
            RUN(queue_wgrad([=](hipStream_t ws, float* wscr) { return pw_wgrad(s, xop, cin, cin, dz, cout, cout, Mo, dwp, ws, wscr, xa); }));
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
            if ((s->fuse_gemm_red & 2) && i - 1 >= 2) {
                a.red_mode = 2; a.red_z = prev.z; a.red_scale = prev.scale; a.red_shift = prev.shift; a.red_mean = prev.mean; a.red_rstd = prev.rstd;
                a.red_act = prev.d.act; a.red_part = s->scratch; a.red_part_floats = s->scratch_floats; a.red_rows_out = &red_rows;
            }
            RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
            if (red_rows > 0) { fused_rows = red_rows; fused_stride = 2 * (int64_t)l.d.cin; fused_dw = false; fused_buf = s->scratch; }
        }
    }
    RUN(flush_wgrads());
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

int loss_forward(ams_student* s, const uint8_t* teacher, int B, int32_t* labels, hipStream_t st) {
    const ams_student_config& c = s->cfg;
    if (!labels && ce_loss_grad_supported(s->w, c.width)) {
        // the fine-tune step: loss sums and the unnormalised gradient in one pass over the pixels
        const bool soft = s->soft_teacher != 0;
        RUNK(0, 0.0, launch_ce_loss_grad(s->logits, 32, B, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher, c.num_classes,
                                         s->loss_buf, s->ce_scratch, st, soft ? s->teacher_logits : nullptr, s->teacher_th, s->teacher_tw));
        return AMS_OK;
    }
    AMS_REQUIRE(labels || !s->soft_teacher, "soft_teacher: %d output columns on %d logit columns is outside the one-pass loss kernel", c.width, s->w);
    return launch_upsample_argmax(s->logits, 32, B, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher,
                                  c.num_classes, labels, s->conf_buf, s->loss_buf, st);
}

int train_step_impl(ams_student* s, const void* frames_dev, int32_t frames_dtype, const uint8_t* teacher_dev,
                           int32_t batch, int32_t global_batch, float lr, const uint8_t* mask_dev, double* loss_dev,
                           ams_allreduce_cb cb, void* user, ams_comm* comm, void* stream) {
    RUN(check_call(s, frames_dev, frames_dtype, batch));
    AMS_REQUIRE(teacher_dev, "train_step: null teacher labels");
    if (!s->cfg.trainable) { set_error("train_step: this student was created frozen (trainable=0)"); return AMS_E_STATE; }
    AMS_REQUIRE(global_batch >= batch, "train_step: global batch %d < local batch %d", global_batch, batch);
    // soft_teacher=True without the feed: TensorFlow's "You must feed a value for placeholder tensor" (teacher_labels_logits_pl)
    AMS_REQUIRE(!s->soft_teacher || s->teacher_logits, "train_step: soft_teacher is on but no teacher logits are fed (ams_student_feed_teacher_logits)");
    hipStream_t st = (hipStream_t)stream;
    if (comm) cb = comm_as_cb;
    SyncCtx sc{cb, user, s, comm};
    const SyncCtx* psc = cb ? &sc : nullptr;
    RUN(forward_live(s, frames_dev, frames_dtype, batch, global_batch, /*update_ema=*/true, psc, st));
    RUN(loss_forward(s, teacher_dev, batch, nullptr, st));
    RUN(sync_doubles(psc, s->loss_buf, 2, st));         // loss sum and valid-pixel count over all ranks
    RUN(backward(s, frames_dev, frames_dtype, teacher_dev, batch, global_batch, psc, st));
    RUN(sync_any(psc, s->grads, (size_t)s->cfg.n_trainable, AMS_DT_F32, st));       // one flat 8.45 MB message
    // regularize=True: after the cross-rank sum of the gradients (every rank holds the same variables: the term is not a per-shard quantity)
    if (s->reg_mask)
        RUN(launch_l2_regularizer(s->params, s->grads, s->reg_mask, s->cfg.n_trainable, s->reg_nvars, s->reg_coef, reinterpret_cast<double*>(s->tmp_c),
                                  s->loss_buf, st));
    if (loss_dev) AMS_CHECK_HIP(hipMemcpyAsync(loss_dev, s->loss_buf, 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
    // Adam, TF1 form; the step counter is never reset (SemanticNetwork.py:25, :154-156)
    s->adam_t += 1;
    // beta1 / beta2 are f32 tensors in the TF graph (0.9f, 0.999f), and so are their running powers
    const double b1 = (double)0.9f, b2 = (double)0.999f;
    const double lr_t = (double)lr * sqrt(1.0 - pow(b2, (double)s->adam_t)) / (1.0 - pow(b1, (double)s->adam_t));
    s->tp_fresh = false;                               // the update below invalidates the live weight panels
    return launch_adam(s->params, s->grads, s->adam_m, s->adam_v, mask_dev, s->cfg.n_trainable, (float)lr_t, 0.9f, 0.999f, 1e-8f, st);
}

}  // namespace ams
