// 1x1 convolutions, streaming variant (huge M, small K x N): see k_pointwise.hip for the overview.
#include "pw_common.hpp"

namespace ams {

// ---- variant S: streaming layers (huge M, small K x N).  The whole weight panel of this column tile stays in LDS for
// the block's lifetime; every wave walks its own 16*RM-row groups (grid-stride), no barrier after the prologue, and
// the A fragment of the NEXT (row group, k chunk) is in flight while the current one feeds the matrix pipe.
// XF = PwArgs::x_mode: 1 — BN + activation of the layer that wrote x, 2 — dz = A x + B + C x2 (second half of BN backward) — applied to every
// operand fragment before it feeds the matrix pipe (per-k vectors in LDS behind everything else); same unfused multiply / add as
// bn_act_kernel / bn_bwd_apply_kernel: bit-identical products.
template <int RM, int NT, int EPI, int XF = 0>
__global__ __launch_bounds__(256, (NT <= 3 ? 3 : NT <= 4 ? 2 : 1)) void pw_gemm_f32_s(PwArgs a, int n_tiles_n, int64_t n_groups) {
    constexpr int PITCH = 16 * NT + 4;
    extern __shared__ __attribute__((aligned(16))) float sW[];          // [Kpad][PITCH] then scale[16NT], shift[16NT]
    const int tile_n = blockIdx.y;
    const int n0 = tile_n * 16 * NT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int K = a.K, n_chunks = (K + 15) / 16;
    float* sSc = sW + n_chunks * 16 * PITCH;
    float* sSh = sSc + 16 * NT;
    float* sOut = sSh + 16 * NT + wave * (16 * (16 * NT + 4));
    pw_stage_w<NT, PITCH>(a, sW, 0, n_chunks * 16, n0, tid, 256);
    pw_stage_affine<NT>(a, sSc, sSh, n0, tid, 256);
    __syncthreads();

    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    const int64_t g_first = (int64_t)blockIdx.x * 4 + wave;
    const bool red = EPI == EPI_PLAIN && a.red_mode != 0;      // block-uniform; the launcher clears red_mode where it does not apply
    float* sRedVec = sSh + 16 * NT + 4 * (16 * (16 * NT + 4));  // behind the waves' output slabs: 4 x 16 NT vectors, then the waves' sums
    float* sXv = sRedVec + (EPI == EPI_PLAIN ? 12 * 16 * NT : 0);  // XF: v0 [Kpad] | v1 [Kpad] (| v2 [Kpad])
    const int Kpad = n_chunks * 16;
    if constexpr (XF != 0) {
        for (int e = tid; e < Kpad; e += 256) {
            sXv[e] = e < K ? a.x_v0[e] : 0.f;
            sXv[Kpad + e] = e < K ? a.x_v1[e] : 0.f;
            if constexpr (XF == 2) sXv[2 * Kpad + e] = e < K ? a.x_v2[e] : 0.f;
        }
        __syncthreads();
    }
    const int64_t x2_off = XF == 2 ? (int64_t)(a.x2 - a.x) : 0;
    float4 rs1[EPI == EPI_PLAIN ? NT : 1], rs2[EPI == EPI_PLAIN ? NT : 1];
    if constexpr (EPI == EPI_PLAIN) {
        if (red) {
            pw_red_stage<NT>(a, sRedVec, n0, tid, 256);
#pragma unroll
            for (int t = 0; t < NT; ++t) { rs1[t] = make_float4(0.f, 0.f, 0.f, 0.f); rs2[t] = rs1[t]; }
            __syncthreads();
        }
    }
    if (g_first >= n_groups && !red) return;                   // (with a fused reduction every wave reaches the block barrier at the end)
    const int64_t my_groups = g_first < n_groups ? (n_groups - 1 - g_first) / wave_stride + 1 : 0;
    const int64_t n_items = my_groups * n_chunks;              // (row group, 16-k chunk) pairs walked by this wave

    // branch-free operand fetch: addresses are clamped into the tensor (rows to M-1, the k offset to K-4) and lanes
    // whose k range lies beyond K are zeroed by a select, so the loop body has no exec-masked VMEM and hipcc can keep
    // the prefetch in flight across the MFMAs (an exec-masked load makes it fall back to s_waitcnt vmcnt(0)).
    auto fetch = [&](int64_t grp, int c, float4 (&dst)[RM], float4 (&dst2)[XF == 2 ? RM : 1]) {
        int koff = c * 16 + 4 * q;
        const bool ok = koff < K;
        if (koff > K - 4) koff = K - 4;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            int64_t m = grp * (16 * RM) + r * 16 + l15;
            if (m > a.M - 1) m = a.M - 1;
            const float4 v = ld4(a.x + m * (int64_t)a.ldx + koff);
            if constexpr (XF != 0) dst[r] = v;           // transformed, then zeroed beyond K, when it is consumed (xform below)
            else dst[r] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            if constexpr (XF == 2) dst2[r] = ld4(a.x + x2_off + m * (int64_t)a.ldx + koff);
        }
    };
    auto xform = [&](int c, float4 (&v)[RM], const float4 (&v2)[XF == 2 ? RM : 1]) {
        int koff = c * 16 + 4 * q;
        const bool ok = koff < K;
        if (koff > K - 4) koff = K - 4;
        const float4 sc = ld4(sXv + koff), sh = ld4(sXv + Kpad + koff);
        if constexpr (XF == 1) {
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const float4 y = muladd4_pk(v[r], sc, sh);
                v[r] = make_float4(ok ? apply_act(y.x, a.x_act) : 0.f, ok ? apply_act(y.y, a.x_act) : 0.f, ok ? apply_act(y.z, a.x_act) : 0.f,
                                   ok ? apply_act(y.w, a.x_act) : 0.f);
            }
        } else {
            const float4 cc = ld4(sXv + 2 * Kpad + koff);
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const float4 y = add4_pk(add4_pk(mul4_pk(sc, v[r]), sh), mul4_pk(cc, v2[r]));        // (A g + B) + C z
                v[r] = make_float4(ok ? y.x : 0.f, ok ? y.y : 0.f, ok ? y.z : 0.f, ok ? y.w : 0.f);
            }
        }
    };
    float4 a_cur[RM], a_nxt[RM];
    float4 z_cur[XF == 2 ? RM : 1], z_nxt[XF == 2 ? RM : 1];
    fetch(g_first < n_groups ? g_first : n_groups - 1, 0, a_cur, z_cur);
    // (Requesting the epilogue's z elements one row group ahead — before the stores of the current one, parked in LDS across the MFMA loop —
    // was built and measured: 61 -> 57 us on the 96-column layer, 90 -> 97 on the 144-column one, nothing on the step.  The kernel tops out at
    // 4.0 - 4.8 TB/s whatever the tile covers; at the step's sizes the fixed costs of a launch are the rest.)
    f32x4 acc[RM][NT];
    int64_t g = g_first;
    int c = 0;
    for (int64_t it = 0; it < n_items; ++it) {
        // next item: next chunk of this group, else chunk 0 of the wave's next group (clamped on the very last item)
        int cn = c + 1;
        int64_t gn = g;
        if (cn == n_chunks) { cn = 0; gn = g + wave_stride; if (gn >= n_groups) gn = g; }
        fetch(gn, cn, a_nxt, z_nxt);
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (XF != 0) xform(c, a_cur, z_cur);
        pw_chunk<RM, NT, PITCH>(acc, a_cur, sW + (c * 16 + 4 * q) * PITCH + l15);
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            a_cur[r] = a_nxt[r];
            // Hand the prefetched fragment over HERE, ahead of the epilogue's stores: vmcnt retires in order and counts
            // stores, so a wait placed after them would sit behind a full store round trip with the matrix pipe idle.
            asm volatile("" : "+v"(a_cur[r].x), "+v"(a_cur[r].y), "+v"(a_cur[r].z), "+v"(a_cur[r].w));
            if constexpr (XF == 2) {
                z_cur[r] = z_nxt[r];
                asm volatile("" : "+v"(z_cur[r].x), "+v"(z_cur[r].y), "+v"(z_cur[r].z), "+v"(z_cur[r].w));
            }
        }
        if (c == n_chunks - 1) {
            if constexpr (EPI == EPI_PLAIN) {
                if (red) pw_red_rowgroups<RM, NT>(a, acc, g * (16 * RM), n0, l15, q, sRedVec, rs1, rs2);
            }
            pw_epilogue_t<RM, NT, EPI>(a, acc, g * (16 * RM), n0, lane, sSc, sSh, sOut);
        }
        c = cn;
        g = gn;
    }
    if constexpr (EPI == EPI_PLAIN) {
        if (red) pw_red_finish<NT>(a, rs1, rs2, lane, wave, 4, sRedVec + 4 * 16 * NT, blockIdx.x, n0, tid, 256);
    }
}

template <int RM, int NT, int EPI, int XF = 0>
static int launch_pw_s_e(const PwArgs& a, hipStream_t st) {
    constexpr int PITCH = 16 * NT + 4;
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    const int64_t n_groups = cdiv64(a.M, 16 * RM);
    // weight panel | scale, shift | four output slabs | (fused reduction: 4 vectors + 4 waves x 2 sums of 16 NT floats) | (XF: 2 x Kpad)
    const size_t lds = ((size_t)((a.K + 15) / 16 * 16) * PITCH + 32 * NT + 4 * 16 * (16 * NT + 4) + (EPI == EPI_PLAIN ? 12 * 16 * NT : 0) +
                        (XF ? (XF + 1) * ((a.K + 15) / 16 * 16) : 0)) * sizeof(float);
    int64_t blocks = cdiv64(n_groups, 4);
    // persistent grid: exactly the blocks that are co-resident (work is pre-partitioned by grid-stride, so any block that
    // has to wait for a slot would run its whole share on a half-empty chip)
    int per_cu = 1, cus = 256;
    RUN_RC(func_allow_lds((const void*)pw_gemm_f32_s<RM, NT, EPI, XF>, lds));
    RUN_RC(func_blocks_per_cu((const void*)pw_gemm_f32_s<RM, NT, EPI, XF>, 256, lds, &per_cu));
    RUN_RC(device_cus(&cus));
    if (knobs().pw_percu > 0) per_cu = knobs().pw_percu;                  // tuning knob AMS_PW_PERCU (tools/bench_kernel.py)
    // (the column tiles are the grid's y dimension: the co-resident blocks are shared between them)
    const int64_t resident = (int64_t)cus * per_cu / n_tiles_n > 0 ? (int64_t)cus * per_cu / n_tiles_n : 1;
    if (blocks > resident) blocks = resident;
    static const std::string nm = "pw_gemm_f32_s<" + std::to_string(RM) + ", " + std::to_string(NT) + ", " + std::to_string(EPI) + (XF ? ", " + std::to_string(XF) + ">" : ">");
    note_kernel(nm.c_str());
    PwArgs b = a;
    if (b.red_mode) {
        if (EPI == EPI_PLAIN && pw_red_ok(b, blocks)) { if (b.red_rows_out) *b.red_rows_out = (int)blocks; }
        else { b.red_mode = 0; if (b.red_rows_out) *b.red_rows_out = 0; }
    }
    hipLaunchKernelGGL((pw_gemm_f32_s<RM, NT, EPI, XF>), dim3((unsigned)blocks, n_tiles_n), dim3(256), lds, st, b, n_tiles_n, n_groups);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int RM, int NT>
static int launch_pw_s(const PwArgs& a, hipStream_t st) {
    if (a.x_mode == 1 && !a.res) return launch_pw_s_e<RM, NT, EPI_PLAIN, 1>(a, st);
    if (a.x_mode == 2 && !a.res) return launch_pw_s_e<RM, NT, EPI_PLAIN, 2>(a, st);
    return a.res ? launch_pw_s_e<RM, NT, EPI_RES>(a, st) : launch_pw_s_e<RM, NT, EPI_PLAIN>(a, st);
}

// column-tile width (in 16s) of the streaming variant, or 0 when it does not apply (weight panel of the tile > 56 KB, or
// an epilogue it does not implement)
static int pw_stream_nt(const PwArgs& a, int force_nt) {
    const int epi = pw_pick_epi(a);
    if (epi != EPI_PLAIN && epi != EPI_RES) return 0;
    const int n16 = cdiv(a.N, 16);
    const int kpad = (a.K + 15) / 16 * 16;
    int nt = n16;
    if (n16 > 6) { const int parts = cdiv(n16, 6); nt = cdiv(n16, parts); }     // column tiles of at most 6 x 16
    // Short contraction, wide result with the plain epilogue (the early blocks' input-gradient GEMMs of the fine-tune step, K = 16 .. 64 ->
    // N = 96 .. 192, with the BN-backward reduction in the epilogue): the kernel is a stream over z and y, and the 96-wide tile costs
    // 236 + 48 registers = ONE wave per SIMD.  48-wide tiles run three waves per SIMD; x is read once per column tile, but x is the small
    // tensor here.  tools/sweep_red.sh on MI355X: 74 -> 59 us (265224 x 24 -> 96), 55 -> 43 (67080 x 32 -> 192), 103 -> 99, 43 -> 39.
    if (epi == EPI_PLAIN && a.K <= 64 && n16 > 3) { const int parts = cdiv(n16, 3); nt = cdiv(n16, parts); }
    if (force_nt > 0) nt = force_nt;
    if ((size_t)kpad * (16 * nt + 4) * 4 > 56 * 1024) return 0;
    return nt;
}

bool pointwise_stream_applies(const PwArgs& a) { return pw_stream_nt(a, 0) > 0; }
// launch_pointwise on this problem runs the streaming kernel AND that kernel applies PwArgs::x_mode on its operand loads
bool pointwise_transforms_on_load(const PwArgs& a) {
    if (a.x_mode == 0) return true;
    if (a.M < 32768 || pw_stream_nt(a, 0) <= 0) return false;
    return !a.res && a.x_v0 && a.x_v1 && (a.x_mode == 1 || (a.x_mode == 2 && a.x_v2 && a.x2 && a.x_act == AMS_ACT_NONE));
}

// returns AMS_OK and sets *handled when the streaming variant applies (weight panel <= 56 KB, vector-friendly layout)
int launch_pointwise_stream(const PwArgs& a, int force_rm, int force_nt, bool* handled, hipStream_t st) {
    *handled = false;
    const int nt = pw_stream_nt(a, force_nt);
    if (nt == 0) return AMS_OK;
    if (a.x_mode != 0 && !(!a.res && a.x_v0 && a.x_v1 && (a.x_mode == 1 || (a.x_mode == 2 && a.x_v2 && a.x2 && a.x_act == AMS_ACT_NONE))))
        return AMS_OK;                                                                              // transforms only on a plain epilogue: the caller materialises
    *handled = true;
    switch (nt) {
        case 1: return launch_pw_s<2, 1>(a, st);
        case 2: return launch_pw_s<2, 2>(a, st);
        case 3: return launch_pw_s<2, 3>(a, st);
        case 4: return launch_pw_s<2, 4>(a, st);
        case 5: return launch_pw_s<2, 5>(a, st);
        case 6: return launch_pw_s<2, 6>(a, st);
        default: break;
    }
    *handled = false;
    return AMS_OK;
}

}  // namespace ams
