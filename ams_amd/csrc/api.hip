// The student engine, part 4 of 4: the C ABI of include/ams_hip.h — student lifetime, the calls behind SemanticNetwork's methods, options,
// the profiler read-out, and the kernel-level entry points (ams_k_*) the parity tests drive.
#include "engine.hpp"

using namespace ams;

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
    int rc = student_build(&tmp, cfg, layers);
    if (rc) return rc;
    return student_layout(&tmp, nullptr, bytes_out);
}

int ams_student_create(const ams_student_config* cfg, const ams_layer_desc* layers, void* arena_dev, size_t arena_bytes,
                       ams_student** out) {
    AMS_REQUIRE(out && arena_dev, "create: null pointer");
    AMS_REQUIRE(((uintptr_t)arena_dev & 255) == 0, "create: arena must be 256-byte aligned");
    ams_student* s = new ams_student();
    int rc = student_build(s, cfg, layers);
    size_t need = 0;
    if (!rc) rc = student_layout(s, arena_dev, &need);
    if (!rc && need > arena_bytes) { set_error("create: arena too small (%zu < %zu)", arena_bytes, need); rc = AMS_E_NOMEM; }
    if (rc) { delete s; return rc; }
    s->arena = (char*)arena_dev;
    s->arena_bytes = arena_bytes;
    {   // events are free; STREAMS are not: the runtime multiplexes them onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by
        // default), and two streams that land on one queue serialise.  A student therefore owns only the streams it uses: the part
        // streams appear with the first multi-part call (never inside a graph capture: ensure_part_streams).
        int e = create_sync_event(&s->ev_fork_dual);
        for (int k = 0; k < 3 && e == AMS_OK; ++k) e = create_sync_event(&s->part_done[k]);
        if (e != AMS_OK) { delete s; return e; }
    }
    if (s->vec_ones) {
        std::vector<float> ones(1024, 1.0f), zeros(1024, 0.0f);
        hipError_t e = hipMemcpy(s->vec_ones, ones.data(), 1024 * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->vec_zeros, zeros.data(), 1024 * sizeof(float), hipMemcpyHostToDevice);
        std::vector<float> inv(1024, (float)(1.0 / ((double)s->h * s->w)));
        if (e == hipSuccess) e = hipMemcpy(s->vec_inv_hw, inv.data(), 1024 * sizeof(float), hipMemcpyHostToDevice);
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

int ams_student_layer_tensor(const ams_student* s, int32_t layer, int32_t which, size_t* offset_bytes, size_t* n_elems) {
    AMS_REQUIRE(s && offset_bytes && n_elems, "layer_tensor: null pointer");
    AMS_REQUIRE(layer >= 1 && layer <= s->cfg.n_layers && which >= 0 && which <= 3, "layer_tensor: layer %d / which %d out of range", layer, which);
    const LayerRt& l = s->L[layer];
    const float* p = which == 0 ? l.z : which == 1 ? l.a : which == 2 ? l.da : (l.dzp ? l.dzp : l.da);
    if (!p) { set_error("layer_tensor: layer %d has no such tensor (trainable=%d)", layer, s->cfg.trainable); return AMS_E_STATE; }
    *offset_bytes = (size_t)((const char*)p - s->arena);
    *n_elems = (size_t)s->cfg.max_batch * l.px_out * l.d.cout;
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
    if (s->L[1].whf_mem) RUN(launch_split_weights_f16(s->fparams + s->L[1].d.w_off, 32, 1, 27, 32, 32, s->L[1].whf_mem, s->L[1].whf_mem + 32 * 32, st));
    for (int i = 2; i <= s->cfg.n_layers; ++i) {
        LayerRt& l = s->L[i];
        if (!l.whi) continue;
        const int K = l.d.cin - l.split_k0;
        RUN(launch_split_weights3(s->fparams + l.d.w_off + (int64_t)l.split_k0 * l.d.cout, l.d.cout, 1, K, l.d.cout, l.Kp, l.whi, l.wlo,
                                  l.wlo3, st));
        if (l.whf_mem)
            RUN(launch_split_weights_f16(s->fparams + l.d.w_off + (int64_t)l.split_k0 * l.d.cout, l.d.cout, 1, K, l.d.cout, l.Kp, l.whf_mem,
                                         l.whf_mem + (size_t)l.d.cout * l.Kp, st));
    }
    // fp16's range: a weight of 65520 or more has hi = inf, lo = -inf -> NaN logits.  Every freeze checks the frozen weights of the
    // layers that hold fp16 panels (one small launch, one 4-byte-per-layer copy, one wait on `stream`: freeze is a phase boundary, not the hot
    // path) and a layer outside the range loses its fp16 form until a later freeze finds it inside again: it runs on three bf16 parts (f32's
    // range) like the fine-tune step.  Activations are not checked: see INTEGRATION.md "Range of the fp16 product form".
    {
        std::vector<WeightRange> jobs;
        std::vector<int> which;
        for (int i = 1; i <= s->cfg.n_layers; ++i) {
            LayerRt& l = s->L[i];
            l.whf = l.whf_mem;
            if (!l.whf_mem) continue;
            const int64_t n = i == 1 ? 27 * 32 : (int64_t)l.d.cin * l.d.cout;
            jobs.push_back(WeightRange{s->fparams + l.d.w_off, n});
            which.push_back(i);
        }
        if (!jobs.empty()) {
            std::vector<int> over(jobs.size(), 0);
            RUN(weights_beyond(jobs, 65504.f, reinterpret_cast<int*>(s->tmp_c), over.data(), st));
            s->f16_fallback_layers = 0;
            for (size_t k = 0; k < jobs.size(); ++k)
                if (over[k]) { s->L[which[k]].whf = nullptr; ++s->f16_fallback_layers; }
        }
    }
    s->frozen_ready = true;
    return AMS_OK;
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

int ams_student_predict_frames_u8(ams_student* s, const void* frames_dev, int32_t frames_dtype, int32_t batch, int32_t mode, const uint8_t* teacher_dev,
                                  uint8_t* labels_out_dev, int64_t* conf_mats_dev, double* losses_dev, void* stream) {
    RUN(check_call(s, frames_dev, frames_dtype, batch));
    AMS_REQUIRE(labels_out_dev, "predict_frames_u8: null output");
    AMS_REQUIRE(teacher_dev == nullptr || (conf_mats_dev && losses_dev), "predict_frames_u8: metrics need conf and loss buffers");
    hipStream_t st = (hipStream_t)stream;
    RUN(run_forward(s, frames_dev, frames_dtype, batch, mode, st));
    const ams_student_config& c = s->cfg;
    return launch_upsample_argmax(s->logits, 32, batch, s->h, s->w, c.class_indices, c.n_selected, c.height, c.width, teacher_dev, c.num_classes,
                                  reinterpret_cast<int32_t*>(labels_out_dev), teacher_dev ? conf_mats_dev : nullptr, teacher_dev ? losses_dev : nullptr, st,
                                  /*per_frame=*/1, /*labels_u8=*/1);
}

int ams_cross_confusion(const ams_student* s, const uint8_t* labels_dev, int64_t n_pixels, int64_t* conf_mat_dev, void* stream) {
    AMS_REQUIRE(s && labels_dev && conf_mat_dev && n_pixels > 0, "cross_confusion: bad argument");
    int32_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = -1;
    for (int k = 0; k < s->cfg.n_selected; ++k) lut[s->cfg.class_indices[k]] = k;
    return launch_cross_confusion(labels_dev, labels_dev + n_pixels, n_pixels, lut, s->cfg.n_selected, conf_mat_dev,
                                  (hipStream_t)stream);
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

int ams_student_f16_fallback_layers(const ams_student* s, int32_t* n_layers) {
    AMS_REQUIRE(s && n_layers, "f16_fallback_layers: null pointer");
    *n_layers = s->f16_fallback_layers;
    return AMS_OK;
}

int ams_student_feed_teacher_logits(ams_student* s, const float* teacher_logits_dev, int32_t th, int32_t tw) {
    AMS_REQUIRE(s, "feed_teacher_logits: null student");
    if (!teacher_logits_dev) { s->teacher_logits = nullptr; s->teacher_th = s->teacher_tw = 0; return AMS_OK; }
    AMS_REQUIRE(th >= 1 && tw >= 1 && th <= s->cfg.height && tw <= s->cfg.width, "feed_teacher_logits: %d x %d teacher logits for %d x %d labels", th, tw,
                s->cfg.height, s->cfg.width);
    s->teacher_logits = teacher_logits_dev; s->teacher_th = th; s->teacher_tw = tw;
    return AMS_OK;
}

int ams_student_set_regularizer(ams_student* s, const uint8_t* reg_mask_dev, int32_t n_vars, float coef) {
    AMS_REQUIRE(s, "set_regularizer: null student");
    if (!reg_mask_dev) { s->reg_mask = nullptr; s->reg_nvars = 0; s->reg_coef = 0.f; return AMS_OK; }
    AMS_REQUIRE(s->cfg.trainable, "set_regularizer: this student was created frozen");
    AMS_REQUIRE(n_vars > 0 && coef >= 0.f, "set_regularizer: n_vars=%d coef=%g", n_vars, (double)coef);
    s->reg_mask = reg_mask_dev; s->reg_nvars = n_vars; s->reg_coef = coef;
    return AMS_OK;
}

int ams_student_set_option(ams_student* s, int32_t option, int32_t value) {
    AMS_REQUIRE(s, "set_option: null student");
    s->dual_choice.clear();                            // any option may change the plans the autotune compared
    if (option == AMS_OPT_MATMUL) {
        AMS_REQUIRE(value == AMS_MATMUL_F32 || value == AMS_MATMUL_SPLIT_BF16 || value == AMS_MATMUL_SPLIT_BF16_X6 || value == AMS_MATMUL_BF16 ||
                        value == AMS_MATMUL_SPLIT_F16,
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
        s->fuse_dgrad_bn = value < 0 ? 0 : (value > 3 ? 3 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_GEMM_RED) {
        s->fuse_gemm_red = value & 3;
        return AMS_OK;
    }
    if (option == AMS_OPT_TRAIN_RECOMPUTE) {
        s->train_recompute = value < 0 ? 0 : (value > 2 ? 2 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_FUSE_OPERAND_BN) {
        s->fuse_operand_bn = value & 1;
        return AMS_OK;
    }
    if (option == AMS_OPT_WGRAD_FORK_EVERY) {
        AMS_REQUIRE(value >= 1 && value <= 64, "set_option: AMS_OPT_WGRAD_FORK_EVERY must be in 1 .. 64");
        s->wgrad_fork_every = value;
        return AMS_OK;
    }
    if (option == AMS_OPT_TRAIN_FWD_F16) {
        s->train_fwd_f16 = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_SOFT_TEACHER) {
        s->soft_teacher = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_NAN_GRADS) {
        s->nan_grads = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_OVERLAP_WGRAD) {
        s->overlap_wgrad = value < 0 ? 0 : (value > 3 ? 3 : value);
        return AMS_OK;
    }
    if (option == AMS_OPT_OVERLAP_HEAD) {
        s->overlap_head = value != 0;
        return AMS_OK;
    }
    if (option == AMS_OPT_STREAM_MIN_ROWS) {
        AMS_REQUIRE(value >= 0, "set_option: AMS_OPT_STREAM_MIN_ROWS must be >= 0");
        s->stream_min_rows = value;
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

int ams_k_pack_h2i(const float* x, int64_t M, int32_t Cn, float* out, void* stream) { return launch_pack_h2i(x, M, Cn, out, (hipStream_t)stream); }

int ams_k_pointwise_split_f16(const float* x, int64_t M, int32_t K, const float* w, int32_t N, const float* scale, const float* shift, int32_t act,
                              const float* res, float* y, uint16_t* panels, size_t panel_elems, float* x_h2i, uint16_t* y_parts, void* stream) {
    const int Kp = (K + 31) / 32 * 32;
    const size_t plane = (size_t)N * Kp;
    AMS_REQUIRE(panels && panel_elems >= 2 * plane, "pointwise_split_f16: panel scratch too small (need %zu)", 2 * plane);
    hipStream_t st = (hipStream_t)stream;
    RUN(launch_split_weights_f16(w, N, 1, K, N, Kp, panels, panels + plane, st));
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    a.scale = scale; a.shift = shift; a.act = act; a.res = res; a.ldr = N;
    if (scale && !shift) { set_error("pointwise_split_f16: scale without shift"); return AMS_E_INVALID; }
    if (x_h2i) {                                   // the operand as fp16 pairs, as the streaming expand + depthwise kernels leave it
        if (x_h2i != x) RUN(launch_pack_h2i(x, M, K, x_h2i, st));      // x_h2i == x: x IS the packed operand already (tools/bench_kernel.py)
        a.x = x_h2i; a.x_fmt = 1;
    }
    if (y_parts) {                                 // the result also as two fp16 part planes [2][M][N] (the next block's operand)
        AMS_REQUIRE(pointwise_split_writes_parts(a), "pointwise_split_f16: this shape has no vector epilogue to write parts from");
        a.ysplit = y_parts; a.ysplit_plane = M * (int64_t)N; a.ysplit_np = 2; a.ysplit_fmt = 1;
    }
    AMS_REQUIRE(pointwise_f16_applies(a), "pointwise_split_f16: unsupported shape M=%lld K=%d", (long long)M, K);
    return launch_pointwise_split_f16(a, panels, (int64_t)plane, Kp, st);
}

int ams_k_expand_dw_stream_f16(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e,
                               const float* shift_e, int32_t Cexp, const float* w_dw, int32_t rate, const float* scale_d, const float* shift_d,
                               float* y, uint16_t* panels, size_t panel_elems, int32_t presplit, int32_t y_h2i, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!expand_dw_stream_supported(Cin, Cexp, 1, rate) || Cin < 64) {
        set_error("expand_dw_stream_f16: unsupported shape Cin=%d Cexp=%d rate=%d", Cin, Cexp, rate);
        return AMS_E_INVALID;
    }
    const size_t plane = (size_t)Cexp * Cin, xplane = (size_t)B * H * W * Cin;
    AMS_REQUIRE(panels && panel_elems >= 2 * plane + (presplit ? 2 * xplane : 0), "expand_dw_stream_f16: panel scratch too small (need %zu)",
                2 * plane + (presplit ? 2 * xplane : 0));
    RUN(launch_split_weights_f16(w_exp, Cexp, 1, Cin, Cexp, Cin, panels, panels + plane, st));
    const uint16_t* xp = nullptr;
    if (presplit) {                                // the operand as two fp16 part planes, as a producing GEMM leaves it (PwArgs::ysplit_fmt 1)
        uint16_t* xq = panels + 2 * plane;
        RUN(launch_split_weights_f16(x, 1, Cin, Cin, (int)((int64_t)B * H * W), Cin, xq, xq + xplane, st));
        xp = xq;
    }
    if (presplit == 2)
        return launch_expand_dw_wreg(xp, (int64_t)xplane, B, H, W, Cin, panels, (int64_t)plane, AMS_NP_F16, scale_e, shift_e, AMS_ACT_RELU6, Cexp, w_dw,
                                     rate, scale_d, shift_d, AMS_ACT_RELU6, y, st, y_h2i ? 1 : 0);
    return launch_expand_dw_stream(x, xp, (int64_t)xplane, B, H, W, Cin, nullptr, panels, (int64_t)plane, AMS_NP_F16, scale_e, shift_e, AMS_ACT_RELU6,
                                   Cexp, w_dw, 1, rate, scale_d, shift_d, AMS_ACT_RELU6, y, st, y_h2i ? 1 : 0);
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

int ams_k_block_fused_f16(const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* w_exp, const float* scale_e, const float* shift_e,
                          int32_t Cexp, const float* w_dw, int32_t stride, const float* scale_d, const float* shift_d, const float* w_proj, int32_t Cout,
                          const float* scale_p, const float* shift_p, int32_t residual, float* y, uint16_t* panels, size_t panel_elems, void* stream) {
    if (!block_fused_supported(Cin, Cexp, Cout, stride, 1, residual != 0)) { set_error("block_fused_f16: unsupported shape"); return AMS_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    const int kp_p = (Cexp + 31) / 32 * 32;
    const int64_t pe = (int64_t)Cexp * 32, pp = (int64_t)Cout * kp_p;
    AMS_REQUIRE(panels && Cin <= 32 && panel_elems >= (size_t)(2 * pe + 2 * pp), "block_fused_f16: panel scratch too small (need %zu)", (size_t)(2 * pe + 2 * pp));
    uint16_t* he = panels;
    uint16_t* hp = panels + 2 * pe;
    RUN(launch_split_weights_f16(w_exp, Cexp, 1, Cin, Cexp, 32, he, he + pe, st));
    RUN(launch_split_weights_f16(w_proj, Cout, 1, Cexp, Cout, kp_p, hp, hp + pp, st));
    return launch_block_fused(x, B, H, W, Cin, w_exp, scale_e, shift_e, AMS_ACT_RELU6, Cexp, w_dw, stride, scale_d, shift_d, AMS_ACT_RELU6, w_proj,
                              scale_p, shift_p, AMS_ACT_NONE, Cout, residual != 0, y, st, nullptr, nullptr, 0, he, pe, hp, pp, kp_p);
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

int ams_k_ce_loss_grad_soft(const float* logits, int32_t B, int32_t h, int32_t w, int32_t NC, const int32_t* class_idx_host, int32_t K, int32_t H,
                            int32_t W, const uint8_t* teacher, const float* teacher_logits, int32_t th, int32_t tw, double* loss_dev, float* dlogits,
                            float* scratch, size_t scratch_floats, void* stream) {
    AMS_REQUIRE(teacher_logits, "ce_loss_grad_soft: null teacher logits");
    AMS_REQUIRE(ce_loss_grad_supported(w, W), "ce_loss_grad: %d output columns on %d source columns is outside the one-pass kernel", W, w);
    AMS_REQUIRE(scratch && scratch_floats >= ce_loss_grad_scratch(B, h, w, K), "ce_loss_grad: scratch too small (need %zu floats)",
                ce_loss_grad_scratch(B, h, w, K));
    hipStream_t st = (hipStream_t)stream;
    RUN(launch_ce_loss_grad(logits, NC, B, h, w, class_idx_host, K, H, W, teacher, NC, loss_dev, scratch, st, teacher_logits, th, tw));
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
    hipStream_t st = (hipStream_t)stream;
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    if (trans_w) { a.w_sk = 1; a.w_sn = K; }
    a.res = res; a.ldr = N;
    a.red_mode = mode; a.red_center = center; a.red_z = z; a.red_scale = scale; a.red_shift = shift; a.red_mean = mean; a.red_rstd = rstd;
    a.red_act = act; a.red_part = part; a.red_part_floats = part_floats;         // rows that would not fit: the launch runs unfused, *rows_out = 0
    int rows = 0;
    a.red_rows_out = &rows;
    int rc;
    if (split == 2) {                              // two fp16 parts (mode 1 only: the forward statistics)
        const int Kp = (K + 31) / 32 * 32;
        const size_t plane = (size_t)N * Kp;
        AMS_REQUIRE(panels && panel_elems >= 2 * plane && pointwise_f16_applies(a), "pointwise_red: fp16 form needs mode 1, K %% 8 == 0 and %zu panel elements", 2 * plane);
        RUN(launch_split_weights_f16(w, a.w_sk, a.w_sn, K, N, Kp, panels, panels + plane, st));
        rc = launch_pointwise_split_f16(a, panels, (int64_t)plane, Kp, st);
    } else if (split) {
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

int ams_k_pointwise_xform(const float* x, int64_t M, int32_t K, const float* w, int32_t N, int32_t trans_w, int32_t split, int32_t x_mode,
                          int32_t x_act, const float* v0, const float* v1, const float* v2, const float* x2, float* y, float* x_tmp,
                          uint16_t* panels, size_t panel_elems, void* stream) {
    AMS_REQUIRE(x && w && y && v0 && v1 && (x_mode == 1 || (x_mode == 2 && v2 && x2)), "pointwise_xform: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PwArgs a = pw_args(x, M, K, K, w, N, y, N);
    if (trans_w) { a.w_sk = 1; a.w_sn = K; }
    a.x_mode = x_mode; a.x_act = x_act; a.x_v0 = v0; a.x_v1 = v1; a.x_v2 = v2; a.x2 = x2; a.x_tmp = x_tmp;
    if (split == 2) {                              // two fp16 parts (x_mode 1 only)
        const int Kp = (K + 31) / 32 * 32;
        const size_t plane = (size_t)N * Kp;
        AMS_REQUIRE(panels && panel_elems >= 2 * plane && pointwise_f16_applies(a), "pointwise_xform: fp16 form needs x_mode 1, K %% 8 == 0, K <= 1024 and %zu panel elements", 2 * plane);
        RUN(launch_split_weights_f16(w, a.w_sk, a.w_sn, K, N, Kp, panels, panels + plane, st));
        return launch_pointwise_split_f16(a, panels, (int64_t)plane, Kp, st);
    }
    if (split) {
        const int Kp = (K + 31) / 32 * 32;
        const size_t plane = (size_t)N * Kp;
        AMS_REQUIRE(panels && panel_elems >= 3 * plane && K % 8 == 0, "pointwise_xform: panel scratch too small (need %zu) or K %% 8", 3 * plane);
        RUN(launch_split_weights3(w, a.w_sk, a.w_sn, K, N, Kp, panels, panels + plane, panels + 2 * plane, st));
        return launch_pointwise_split3(a, panels, panels + plane, panels + 2 * plane, Kp, st);
    }
    return launch_pointwise(a, st);
}

int ams_k_pointwise_wgrad_xform(const float* x, const float* dy, int64_t M, int32_t K, int32_t N, int32_t split, int32_t x_mode, int32_t x_act,
                                const float* v0, const float* v1, int32_t dy_mode, const float* d0, const float* d1, const float* d2,
                                const float* dy2, float* dw, float* scratch, size_t scratch_floats, void* stream) {
    AMS_REQUIRE(x && dy && dw && scratch, "pointwise_wgrad_xform: null pointer");
    if (split && !pointwise_wgrad_x6_applies(M, K, N, K, N)) { set_error("pointwise_wgrad_xform: shape M=%lld K=%d N=%d outside the split kernel", (long long)M, K, N); return AMS_E_INVALID; }
    WgArgs a;
    a.x = x; a.ldx = K; a.K = K; a.dy = dy; a.ldy = N; a.N = N; a.M = M; a.dw = dw; a.scratch = scratch; a.scratch_floats = scratch_floats;
    a.allow_split = split != 0;
    a.x_mode = x_mode; a.x_act = x_act; a.x_v0 = v0; a.x_v1 = v1;
    a.dy_mode = dy_mode; a.dy_v0 = d0; a.dy_v1 = d1; a.dy_v2 = d2; a.dy2 = dy2;
    return launch_pointwise_wgrad(a, (hipStream_t)stream);
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

size_t ams_k_depthwise3x3_fwd_bn_tiles_scratch(int32_t B, int32_t H, int32_t W, int32_t C, int32_t rate) { return depthwise_fwd_bn2_scratch(B, H, W, C, rate); }
int ams_k_depthwise3x3_fwd_bn_tiles(const float* ze, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, int32_t rate, const float* scale,
                                    const float* shift, int32_t act, const float* center, float* zd, float* scratch, size_t scratch_floats,
                                    int32_t* rows_out, void* stream) {
    AMS_REQUIRE(ze && w && scale && shift && zd && scratch && rows_out, "depthwise3x3_fwd_bn_tiles: null pointer");
    AMS_REQUIRE(scratch_floats >= depthwise_fwd_bn2_scratch(B, H, W, C, rate), "depthwise3x3_fwd_bn_tiles: scratch too small");
    int rows = 0;
    int rc = launch_depthwise_fwd_bn2(ze, B, H, W, C, w, rate, scale, shift, act, center, zd, scratch, &rows, (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}

size_t ams_k_depthwise3x3_dgrad_bn_apply_scratch(int32_t B, int32_t H, int32_t W, int32_t C, int32_t rate) { return depthwise_dgrad_bn2_scratch(B, H, W, C, rate); }
int ams_k_depthwise3x3_dgrad_bn_apply(const float* dy, const float* zd, const float* cA, const float* cB, const float* cC, int32_t B, int32_t H, int32_t W,
                                      int32_t C, const float* w, int32_t rate, const float* z_prev, const float* scale, const float* shift, int32_t act,
                                      const float* mean, const float* rstd, float* out, float* scratch, size_t scratch_floats, int32_t* rows_out,
                                      void* stream) {
    AMS_REQUIRE(dy && zd && cA && cB && cC && w && z_prev && scale && shift && mean && rstd && out && scratch && rows_out, "depthwise3x3_dgrad_bn_apply: null pointer");
    AMS_REQUIRE(rate == 1 || rate == 2, "depthwise3x3_dgrad_bn_apply: rate %d", rate);
    AMS_REQUIRE(scratch_floats >= depthwise_dgrad_bn2_scratch(B, H, W, C, rate), "depthwise3x3_dgrad_bn_apply: scratch too small");
    int rows = 0;
    int rc = launch_depthwise_dgrad_bn2(dy, zd, cA, cB, cC, B, H, W, C, w, rate, z_prev, scale, shift, act, mean, rstd, out, scratch, &rows, (hipStream_t)stream);
    *rows_out = rows;
    return rc;
}

size_t ams_k_xx_gram_scratch(int64_t M, int32_t Cin) { return xx_stats_scratch_doubles(M, Cin); }
int ams_k_xx_gram(const float* x, int64_t M, int32_t Cin, double* scratch, size_t scratch_doubles, double* xx64, float* xx32, void* stream) {
    AMS_REQUIRE(scratch_doubles >= xx_stats_scratch_doubles(M, Cin), "xx_gram: scratch too small");
    return launch_xx_gram(x, M, Cin, scratch, xx64, xx32, (hipStream_t)stream);
}
int ams_k_expand_stats(const double* xx64, int32_t Cin, const float* w_exp, int32_t Cexp, double n, const float* center, const float* gamma,
                       const float* beta, float eps, float one_minus_decay, float* moving_mean, float* moving_var, float* scale, float* shift,
                       float* save_mean, float* save_rstd, double* sums, void* stream) {
    return launch_expand_stats(xx64, Cin, w_exp, Cexp, n, center, gamma, beta, eps, one_minus_decay, moving_mean, moving_var, scale, shift, save_mean,
                               save_rstd, sums, (hipStream_t)stream);
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

