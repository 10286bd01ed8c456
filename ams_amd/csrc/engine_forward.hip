// The student engine, part 2 of 4: frozen inference (BN folded; what the edge device runs), its multi-stream plans, and the live
// forward with training-mode BN (reference SemanticNetwork.py:170-182 predict_input, utils/graph_utils.py:373-402 the training graph).
#include "engine.hpp"

namespace ams {

// Split-bf16 pays where the exact-f32 kernels are matrix-pipe bound (f32-input MFMA = 157 TFLOP/s against ~5 TB/s of HBM:
// ~31 FLOP per byte): few rows (the streaming kernel needs >= 32768), a weight panel too large for the streaming kernel,
// or an arithmetic intensity 2KN / 4(K+N) of 20 FLOP/B and more (64 -> 384 and wider, at any batch size).
bool split_pays(const PwArgs& a) {
    if (a.K < 32 || a.K % 8 != 0 || a.M < 256) return false;
    return a.M < 32768 || !pointwise_stream_applies(a) || (int64_t)a.K * a.N >= 40 * (int64_t)(a.K + a.N);
}

// live (training) 1x1 layer or its input gradient: same split-bf16 rule as the frozen path, the weights are split right
// before the launch because they change every step (one small kernel; the panels live in one shared scratch buffer)
int live_pointwise(ams_student* s, const PwArgs& a, hipStream_t st) {
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
            if (j.K == a.K && j.N == a.N && j.sk == a.w_sk && j.sn == a.w_sn) {
                if (s->tp_wait) { AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_tp, 0)); s->tp_wait = false; }      // the split ran on the side stream
                // forward products of the step on two fp16 parts (3 MFMAs): activations and weights are O(1) there; the input-gradient GEMMs
                // (red_mode 2, transposed panels: no fp16 planes) keep three bf16 parts — gradients span a range fp16 cannot hold
                if (s->matmul_mode == AMS_MATMUL_SPLIT_F16 && s->train_fwd_f16 && j.f16 && pointwise_f16_applies(a))
                    return launch_pointwise_split_f16(a, j.p0 + 3 * j.plane, j.plane, j.Kp, st);
                return launch_pointwise_split3(a, j.p0, j.p0 + j.plane, j.p0 + 2 * j.plane, j.Kp, st);
            }
        }
    }
    AMS_REQUIRE(3 * plane <= s->panel_elems, "live_pointwise: panel scratch too small");
    uint16_t* p0 = s->panel_scratch;
    int rc = launch_split_weights3(a.w, a.w_sk, a.w_sn, a.K, a.N, Kp, p0, p0 + plane, p0 + 2 * plane, st);
    if (rc) return rc;
    return launch_pointwise_split3(a, p0, p0 + plane, p0 + 2 * plane, Kp, st);
}

// frozen 1x1 layer: late layers (few rows, wide K/N: matrix-pipe bound) go through the split-bf16 kernel
int frozen_pointwise(ams_student* s, int layer, PwArgs a, hipStream_t st, bool* wrote_parts, bool force_split) {
    const LayerRt& l = s->L[layer];
    const bool split = s->matmul_mode != AMS_MATMUL_F32 && l.whi && (split_pays(a) || (force_split && a.K % 8 == 0 && a.K >= 32));
    // the bf16 parts of the result (a.ysplit) exist only when the split kernel runs with a vector epilogue
    const bool parts = split && a.ysplit && pointwise_split_writes_parts(a);
    if (!parts) a.ysplit = nullptr;
    if (wrote_parts) *wrote_parts = parts;
    if (split && s->matmul_mode == AMS_MATMUL_SPLIT_F16 && l.whf && pointwise_f16_applies(a))
        RUNK(layer, pw_bytes(a), launch_pointwise_split_f16(a, l.whf, (int64_t)l.d.cout * l.Kp, l.Kp, st));
    else if (a.x_fmt != 0) { set_error("frozen_pointwise: layer %d was handed fp16 pairs but does not run the fp16 product", layer); return AMS_E_STATE; }
    else if (split && s->matmul_mode == AMS_MATMUL_SPLIT_F16) RUNK(layer, pw_bytes(a), launch_pointwise_split3(a, l.whi, l.wlo, l.wlo3, l.Kp, st));
    else if (split && s->matmul_mode == AMS_MATMUL_SPLIT_BF16_X6) RUNK(layer, pw_bytes(a), launch_pointwise_split3(a, l.whi, l.wlo, l.wlo3, l.Kp, st));
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
                const bool h16 = s->block_x6 && s->matmul_mode == AMS_MATMUL_SPLIT_F16 && l.whf && lj.whf && lj.Kp == 32;
                const bool x6 = !h16 && s->block_x6 && s->matmul_mode != AMS_MATMUL_F32 && l.whi;
                RUNK(3, bytes, launch_first_block(frames, dtype, B, c.height, c.width, c.pixel_scale, P + l.d.w_off, l.fscale, l.fshift,
                                                  l.d.act, P + ld.d.w_off, ld.fscale, ld.fshift, ld.d.act, P + lj.d.w_off, lj.fscale,
                                                  lj.fshift, lj.d.act, cur, st, x6 ? l.whi : nullptr, 32 * 32, h16 ? l.whf : nullptr, 32 * 32,
                                                  h16 ? lj.whf : nullptr, (int64_t)lj.d.cout * lj.Kp));
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
        bool d_h2i = false;                    // x (the depthwise result) is stored as fp16 pairs (PwArgs::x_fmt 1)
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
            // fp16 form (AMS_MATMUL_SPLIT_F16): expand (any K, 16 included) and project products as 3 fp16 MFMAs each
            const bool h16 = s->block_x6 && s->matmul_mode == AMS_MATMUL_SPLIT_F16 && le.whf && lj.whf && le.Kp == 32;
            const bool x6 = !h16 && s->block_x6 && s->matmul_mode != AMS_MATMUL_F32 && le.whi && le.Kp == 32 && le.d.cin > 16;
            const double fl_e = 2.0 * B * (double)le.px_in * le.d.cin * le.d.cout, fl_p = 2.0 * B * (double)lj.px_out * lj.d.cin * lj.d.cout;
            s->prof_flops = 2.0 * B * (double)ld.px_out * 9.0 * ld.d.cin + (h16 ? 0.0 : fl_p) + (x6 || h16 ? 0.0 : fl_e);
            s->prof_flops_x6 = h16 ? fl_e + fl_p : x6 ? fl_e : 0.0;
            RUNK(i + 2, bytes, launch_block_fused(cur, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, le.fscale, le.fshift, le.d.act, le.d.cout,
                                                  P + ld.d.w_off, ld.d.stride, ld.fscale, ld.fshift, ld.d.act, P + lj.d.w_off, lj.fscale, lj.fshift,
                                                  lj.d.act, lj.d.cout, res, s->act[o], st, le.blk_vecs, x6 ? le.whi : nullptr,
                                                  (int64_t)(le.wlo - le.whi), h16 ? le.whf : nullptr, (int64_t)le.d.cout * le.Kp, h16 ? lj.whf : nullptr,
                                                  (int64_t)lj.d.cout * lj.Kp, lj.Kp));
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
            // (Cin <= 32 streams on the exact-f32 form whatever the mode: no fp16 parts there, so no fp16-pair hand-over of d either)
            const bool h16 = s->matmul_mode == AMS_MATMUL_SPLIT_F16 && le.whf && le.d.cin > 32;
            const int np = h16 ? AMS_NP_F16 : s->matmul_mode == AMS_MATMUL_SPLIT_BF16_X6 || s->matmul_mode == AMS_MATMUL_SPLIT_F16 ? 3 : s->matmul_mode == AMS_MATMUL_BF16 ? 1 : 2;
            const uint16_t* wparts = h16 ? le.whf : le.whi;
            const int64_t wplane = h16 ? (int64_t)le.d.cout * le.Kp : (int64_t)(le.wlo - le.whi);
            // fp16 form: the depthwise result goes to the project GEMM as fp16 pairs (PwArgs::x_fmt 1) when that GEMM will run the fp16 product
            d_h2i = false;
            if (h16 && i + 2 <= s->n_backbone && s->L[i + 2].d.role == AMS_ROLE_PROJECT && s->L[i + 2].whf && ld.d.cout % 8 == 0 && !s->emulate_bf16_storage) {
                const LayerRt& lj = s->L[i + 2];
                PwArgs pj = pw_args(nullptr, (int64_t)B * lj.px_in, lj.d.cin, lj.d.cin, nullptr, lj.d.cout, nullptr, lj.d.cout);
                d_h2i = split_pays(pj) && pointwise_f16_applies(pj);
            }
            const double bytes = 4.0 * ((double)B * (le.px_in * le.d.cin + ld.px_out * ld.d.cout) + (double)le.d.cin * le.d.cout + 9.0 * ld.d.cin);
            const int64_t xplane = (int64_t)B * le.px_in * le.d.cin;
            if (le.d.cin > 96 && cur_parts)
                // 160 -> 960: expand weights in registers, the operand staged once per block in LDS (k_xdw_wreg.hip); with 30 channel
                // chunks the LDS-weight form is bound by its passes over the operand
                RUNK(i + 1, bytes, launch_expand_dw_wreg(cur_parts, xplane, B, le.Hin, le.Win, le.d.cin, wparts, wplane, np, le.fscale,
                                                         le.fshift, le.d.act, le.d.cout, P + ld.d.w_off, ld.d.rate, ld.fscale, ld.fshift, ld.d.act,
                                                         s->act[o], st, d_h2i ? 1 : 0));
            else
                RUNK(i + 1, bytes, launch_expand_dw_stream(x, cur_parts, xplane, B, le.Hin, le.Win, le.d.cin, P + le.d.w_off, wparts, wplane, np, le.fscale,
                                                           le.fshift, le.d.act, le.d.cout, P + ld.d.w_off, ld.d.stride, ld.d.rate, ld.fscale, ld.fshift, ld.d.act,
                                                           s->act[o], st, d_h2i ? 1 : 0));
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
            a.x_fmt = d_h2i ? 1 : 0;
            bool wrote = false;
            if (stream_ok(i + 1) && s->xsplit && (size_t)a.M * a.N <= s->xsplit_plane) {
                // the next block streams: its expand GEMM takes this result as bf16 parts, written here once instead of being
                // split by every channel-chunk block there
                a.ysplit = s->xsplit; a.ysplit_plane = a.M * a.N; a.ysplit_np = s->matmul_mode == AMS_MATMUL_SPLIT_BF16_X6 ? 3 : s->matmul_mode == AMS_MATMUL_BF16 ? 1 : 2;
                a.ysplit_fmt = s->matmul_mode == AMS_MATMUL_SPLIT_F16 ? 1 : 0;
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
            if (!s->side) RUN(create_side_stream(&s->side));
            if (!s->ev_fork) RUN(create_sync_event(&s->ev_fork));
            if (!s->ev_head) RUN(create_sync_event(&s->ev_head));
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
bool train_recompute_block(const ams_student* s, int i) {
    if (!s->train_recompute || !s->xt_scratch || i < 2 || i + 1 > s->n_backbone) return false;
    const LayerRt& l = s->L[i];
    const LayerRt& ld = s->L[i + 1];
    return l.xx_g0 && l.d.role == AMS_ROLE_EXPAND && ld.d.role == AMS_ROLE_DEPTHWISE && xdw_train_supported(l.d.cin, l.d.cout, ld.d.stride, ld.d.rate) &&
           expand_dw_supported(l.d.cin, l.d.cout, ld.d.stride, ld.d.rate) && l.d.cout <= 1024 &&
           xdw_train_scratch(s->cfg.max_batch, l.Hin, l.Win, l.d.cin, l.d.cout) != (size_t)-1 &&      // too large for 32-bit offsets: layer-by-layer
           xdw_train_scratch(s->cfg.max_batch, l.Hin, l.Win, l.d.cin, l.d.cout) <= s->xt_floats;
}

// Stride-1 depthwise layer i of a block that keeps its tensors: its backward is ONE kernel that recomputes the expand layer's activation
// from z_e (backward(), k_conv.hip dw3x3_dgrad_bn_kernel) — so nothing in backward reads a_e, and at fuse_dgrad_bn >= 2 the forward
// does not write it either (dw3x3_fwd_bn_kernel applies the expand layer's BN + activation on its tap loads).
bool dw_fused_train(const ams_student* s, int i, int B) {
    if (i < 3 || i > s->n_backbone || !s->fuse_dgrad_bn || train_recompute_block(s, i - 1)) return false;
    const LayerRt& l = s->L[i];
    const LayerRt& prev = s->L[i - 1];
    return l.d.role == AMS_ROLE_DEPTHWISE && l.d.stride == 1 && prev.d.role == AMS_ROLE_EXPAND && prev.d.cout == l.d.cin && l.d.cin <= 1024 &&
           depthwise_dgrad_bn_scratch(B, l.Hin, l.Win, l.d.cin) <= s->scratch_floats;
}
// First block: the stem is the "expand" layer of depthwise layer 2, and the backward of the pair is one pass over dz and the frames that
// never reads the stem's activation either (backward(), launch_xdw_bwd_reduce_stem).
bool stem_fused_train(const ams_student* s) {
    const ams_student_config& c = s->cfg;
    return s->n_backbone >= 2 && s->train_recompute && s->L[1].d.role == AMS_ROLE_STEM && s->L[1].d.cout == 32 && s->L[2].d.role == AMS_ROLE_DEPTHWISE &&
           s->L[2].d.stride == 1 && s->L[2].d.rate == 1 && s->xt_scratch && xdw_stem_scratch(c.max_batch, c.height, c.width) != (size_t)-1 &&
           xdw_stem_scratch(c.max_batch, c.height, c.width) <= s->xt_floats;
}
bool dw_fused_train_fwd(const ams_student* s, int i, int B) {
    if (s->fuse_dgrad_bn < 2 || i < 2 || i > s->n_backbone) return false;
    if (!(i == 2 ? stem_fused_train(s) : dw_fused_train(s, i, B))) return false;
    return depthwise_fwd_bn_scratch(B, s->L[i].Hin, s->L[i].Win, s->L[i].d.cin, s->L[i].d.rate) <= s->scratch_floats;
}

// depthwise layer i feeds a project layer: with AMS_OPT_FUSE_OPERAND_BN bit 0 its activation a = act(z scale + shift) is never written — the
// project GEMM (forward) and the project weight gradient (backward) apply it on their operand loads (PwArgs / WgArgs x_mode 1)
bool operand_bn_act(const ams_student* s, int i) {
    if (!(s->fuse_operand_bn & 1) || i < 2 || i + 1 > s->n_backbone) return false;
    const LayerRt& l = s->L[i];
    const LayerRt& lj = s->L[i + 1];
    return l.d.role == AMS_ROLE_DEPTHWISE && lj.d.role == AMS_ROLE_PROJECT && lj.d.cin == l.d.cout && l.d.cout % 4 == 0 && l.d.cout <= 1024;
}

// most partial rows a GEMM with a fused column reduction can leave behind (PwArgs::red_mode): one per 64-row strip of the tiled split kernel
// (it also takes layers of >= 32768 rows when the panel is too large for the streaming kernel), one per block of the persistent streaming
// kernel (<= 8 per CU).  Sizing only: the launchers compare the exact row count with PwArgs::red_part_floats and drop the fusion when the
// rows would not fit (the caller then runs the separate reduction pass).
size_t red_rows_bound(int64_t M) {
    const size_t tiled = (size_t)(M / 64 + 8), streaming = M >= 32768 ? 8 * 512 : 0;
    return tiled > streaming ? tiled : streaming;
}

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

int forward_live(ams_student* s, const void* frames, int dtype, int B, int global_B, bool update_ema, const SyncCtx* sc,
                        hipStream_t st) {
    const ams_student_config& c = s->cfg;
    AMS_REQUIRE(c.trainable, "live forward needs a trainable student (activations are not allocated)");
    const float* P = s->params;
    // the parameters may have changed since the last call (Adam, restore): all live weight panels in one launch
    s->tp_fresh = false;
    s->tp_wait = false;
    if (s->matmul_mode != AMS_MATMUL_F32 && !s->tp_jobs.empty()) {
        // the first GEMM that reads a panel is a millisecond away (the early blocks run exact f32): the split runs on the side stream beside
        // the stem and the first blocks, and that GEMM waits for its event (live_pointwise)
        hipStream_t ts = st;
        if (s->overlap_wgrad && !s->prof.on && s->scratch2) {
            if (!s->side) RUN(create_side_stream(&s->side));
            if (!s->ev_fork) RUN(create_sync_event(&s->ev_fork));
            if (!s->ev_tp) RUN(create_sync_event(&s->ev_tp));
            AMS_CHECK_HIP(hipEventRecord(s->ev_fork, st));
            AMS_CHECK_HIP(hipStreamWaitEvent(s->side, s->ev_fork, 0));
            ts = s->side;
        }
        RUNK(0, 0.0, launch_split_batch(s->tp_jobs_dev, (int)s->tp_jobs.size(), s->tp_blocks, ts,
                                        s->train_fwd_f16 && s->matmul_mode == AMS_MATMUL_SPLIT_F16));
        if (ts != st) { AMS_CHECK_HIP(hipEventRecord(s->ev_tp, ts)); s->tp_wait = true; }
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
            if (s->train_recompute >= 2 && l.xx64 && s->xx_scratch && l.d.cin <= 32 &&
                xx_stats_scratch_doubles((int64_t)B * l.px_in, l.d.cin) <= s->xx_scratch_doubles) {
                // z_e is linear in x: its sums follow from XX = x^T x and g0 = sum x, formed in f64 in ONE cheap pass over x (k_xx_stats.hip);
                // the float copy is what the expand weight gradient reads (this rank's pixels), the doubles are summed over the ranks first
                const int KP = (l.d.cin + 15) / 16 * 16;
                RUNK(i, 4.0 * B * l.px_in * l.d.cin, launch_xx_gram(x, (int64_t)B * l.px_in, l.d.cin, s->xx_scratch, l.xx64, l.xx_g0, st));
                RUN(sync_doubles(sc, l.xx64, (size_t)KP * KP + KP, st));
                RUN(launch_expand_stats(l.xx64, l.d.cin, P + l.d.w_off, l.d.cout, n_e, center, s->params + l.d.gamma_off, s->params + l.d.beta_off,
                                        l.d.bn_eps, omd, mm, mv, l.scale, l.shift, l.mean, l.rstd, l.fsums, st));
            } else {
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
            }
            // ... which leaves the statistics of that raw output behind as one partial row per tile (no separate pass over z_d)
            int d_rows = 0;
            const bool d_stats = (s->fuse_gemm_red & 1) && expand_dw_stats_scratch(B, l.Hin, l.Win, l.d.cout, ld.d.stride) <= s->scratch_floats;
            RUNK(i + 1, 4.0 * ((double)B * (l.px_in * l.d.cin + ld.px_out * ld.d.cout)),
                 launch_expand_dw(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.scale, l.shift, l.d.act, l.d.cout, P + ld.d.w_off, ld.d.stride,
                                  ld.d.rate, s->vec_ones, s->vec_zeros, AMS_ACT_NONE, ld.z, st, d_stats ? s->stats + ld.d.mean_off : nullptr,
                                  d_stats ? s->scratch : nullptr, d_stats ? &d_rows : nullptr));
            RUN(bn_train(s, ld, (int64_t)B * ld.px_out, (double)global_B * ld.px_out, update_ema, sc, nullptr, st, d_rows, !operand_bn_act(s, i + 1)));
            ++i;
            continue;
        }
        int pre_rows = 0;
        if (l.d.role == AMS_ROLE_DEPTHWISE && dw_fused_train_fwd(s, i, B)) {
            // BN + activation of the expand layer on the tap loads (its `a` was not written), the statistics of the result on the way out
            const LayerRt& le = s->L[i - 1];
            // (from 64 channels on in the LDS-tile form of k_dw_train.hip: BN + activation once per element instead of once per tap load)
            if (s->fuse_dgrad_bn >= 3 && l.d.cin % 64 == 0 && depthwise_fwd_bn2_scratch(B, l.Hin, l.Win, l.d.cin, l.d.rate) <= s->scratch_floats)
                RUNK(i, dw_bytes(l, B), launch_depthwise_fwd_bn2(le.z, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, le.scale, le.shift, le.d.act,
                                                                 s->stats + l.d.mean_off, l.z, s->scratch, &pre_rows, st));
            else
            RUNK(i, dw_bytes(l, B), launch_depthwise_fwd_bn(le.z, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.rate, le.scale, le.shift, le.d.act,
                                                            s->stats + l.d.mean_off, l.z, s->scratch, &pre_rows, st));
        } else if (l.d.role == AMS_ROLE_DEPTHWISE) {
            RUNK(i, dw_bytes(l, B), launch_depthwise(x, B, l.Hin, l.Win, l.d.cin, P + l.d.w_off, l.d.stride, l.d.rate, nullptr, nullptr,
                                                     AMS_ACT_NONE, l.z, st));
        } else {
            PwArgs a = pw_args(x, (int64_t)B * l.px_in, l.d.cin, l.d.cin, P + l.d.w_off, l.d.cout, l.z, l.d.cout);
            if (l.d.role == AMS_ROLE_PROJECT && operand_bn_act(s, i - 1)) {
                // the depthwise layer's activation was not written: BN + activation on this GEMM's loads of its raw output
                const LayerRt& ld = s->L[i - 1];
                a.x = ld.z; a.x_mode = 1; a.x_act = ld.d.act; a.x_v0 = ld.scale; a.x_v1 = ld.shift; a.x_tmp = ld.a;
            }
            // the BN statistics of the result in this GEMM's epilogue, where the kernel chosen can do it
            if (s->fuse_gemm_red & 1) {
                a.red_mode = 1; a.red_center = s->stats + l.d.mean_off; a.red_part = s->scratch; a.red_part_floats = s->scratch_floats; a.red_rows_out = &pre_rows;
            }
            RUNK(0, pw_bytes(a), live_pointwise(s, a, st));
        }
        const float* res = l.d.residual_from ? s->L[l.d.residual_from].a : nullptr;
        const bool act_pass = !(l.d.role == AMS_ROLE_EXPAND && dw_fused_train_fwd(s, i + 1, B)) && !operand_bn_act(s, i);
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
    if (s->tp_wait) { AMS_CHECK_HIP(hipStreamWaitEvent(st, s->ev_tp, 0)); s->tp_wait = false; }      // no GEMM used a panel (tiny frames): join anyway
    return AMS_OK;
}

int check_call(const ams_student* s, const void* frames, int dtype, int batch) {
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

// Parts of the static rule (AMS_OPT_DUAL_STREAM = 1).  Round-5 sweep on MI355X at 512 x 1024 (tools/sweep_parts.py, 1 / 2 / 3 / 4 parts at 8 .. 64
// frames): two parts are 1-8 % ahead of one stream at every size from 8 frames on (0.94 vs 0.98 ms at 8, 1.81 vs 1.96 at 20, 2.56 vs 2.78 at 32,
// 3.28 vs 3.52 at 40, 5.12 vs 5.36 at 64), three at 12, 24 and 48 (1.27 vs 1.29, 2.11 vs 2.22, 3.91 vs 4.00 against two).  A fixed function
// of the batch size: the same call always runs the same plan, and nothing is timed inside a call.
static int dual_parts_static(int batch) {
    if (batch < 8) return 1;
    if (batch == 12 || batch == 24 || batch == 48) return 3;
    return 2;
}

int run_forward(ams_student* s, const void* frames, int dtype, int batch, int mode, hipStream_t st) {
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

}  // namespace ams
