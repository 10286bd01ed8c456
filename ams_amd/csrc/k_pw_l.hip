// 1x1 convolutions, tiled exact-f32 variant (late layers, training): see k_pointwise.hip for the overview.
#include "pw_common.hpp"

namespace ams {

// ---- variant L: late layers (M = B*33*65 rows, K and/or N in the hundreds).  One (64*RM) x (16*NT) tile per block,
// K walked in 32-wide stages whose weight panel is double-buffered in LDS: the next stage's panel travels
// global -> registers while the current one is consumed, and is written to the other buffer before the single barrier.
template <int RM, int NT, int EPI>
__global__ __launch_bounds__(256) void pw_gemm_f32_l(PwArgs a, int n_tiles_n, unsigned nblocks) {
    constexpr int BK = 32;
    constexpr int PITCH = 16 * NT + 4;
    constexpr int NREG = (BK * 16 * NT + 255) / 256;              // staged elements per thread
    __shared__ __attribute__((aligned(16))) float sW[2][BK * PITCH];
    __shared__ __attribute__((aligned(16))) float sSc[16 * NT], sSh[16 * NT];
    __shared__ __attribute__((aligned(16))) float sOutAll[EPI == EPI_GENERIC ? 4 : 4 * 16 * (16 * NT + 4)];
    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int tile_n = lb % n_tiles_n;
    const int64_t tile_m = lb / n_tiles_n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int n0 = tile_n * 16 * NT;
    const int64_t m_base = tile_m * (64 * RM) + wave * (16 * RM);
    const int K = a.K, n_chunks = (K + 15) / 16, n_stages = (n_chunks + 1) / 2;
    constexpr int cols = 16 * NT;
    const bool n_contig = a.w_sn == 1;

    float wreg[NREG];
    auto load_stage = [&](int s) {
        const int k0 = s * BK;
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int e = tid + u * 256;
            int kk, nn;
            if (n_contig) { kk = e / cols; nn = e - kk * cols; } else { nn = e / BK; kk = e - nn * BK; }
            float v = 0.f;
            if (e < BK * cols && k0 + kk < a.Kw && n0 + nn < a.N)
                v = a.w[(int64_t)(k0 + kk) * a.w_sk + (int64_t)(n0 + nn) * a.w_sn];
            wreg[u] = v;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int e = tid + u * 256;
            int kk, nn;
            if (n_contig) { kk = e / cols; nn = e - kk * cols; } else { nn = e / BK; kk = e - nn * BK; }
            if (e < BK * cols) sW[buf][kk * PITCH + nn] = wreg[u];
        }
    };

    const float* arow[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        int64_t m = m_base + r * 16 + l15;
        if (m > a.M - 1) m = a.M - 1;
        arow[r] = a.x + m * (int64_t)a.ldx + 4 * q;
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_cur[RM], a_nxt[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) a_cur[r] = (4 * q < K) ? ld4(arow[r]) : zero4;
    f32x4 acc[RM][NT];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_stage(0);
    pw_stage_affine<NT>(a, sSc, sSh, n0, tid, 256);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < n_stages; ++s) {
        if (s + 1 < n_stages) load_stage(s + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * s + h;
            if (c < n_chunks) {
                const bool ok = c + 1 < n_chunks && (c + 1) * 16 + 4 * q < K;
#pragma unroll
                for (int r = 0; r < RM; ++r) a_nxt[r] = ok ? ld4(arow[r] + (c + 1) * 16) : zero4;
                pw_chunk<RM, NT, PITCH>(acc, a_cur, &sW[s & 1][(h * 16 + 4 * q) * PITCH + l15]);
#pragma unroll
                for (int r = 0; r < RM; ++r) a_cur[r] = a_nxt[r];
            }
        }
        if (s + 1 < n_stages) store_stage((s + 1) & 1);
        __syncthreads();
    }
    if (EPI == EPI_GENERIC) pw_epilogue<RM, NT>(a, acc, m_base, n0, l15, q, sSc, sSh);
    else pw_epilogue_t<RM, NT, EPI>(a, acc, m_base, n0, lane, sSc, sSh, sOutAll + wave * (16 * (16 * NT + 4)));
}

template <int RM, int NT, int EPI>
static int launch_pw_l_e(const PwArgs& a, hipStream_t st) {
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    const int64_t nblocks = cdiv64(a.M, 64 * RM) * n_tiles_n;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL) { set_error("pointwise: bad grid %lld", (long long)nblocks); return AMS_E_INVALID; }
    static const std::string nm = "pw_gemm_f32_l<" + std::to_string(RM) + ", " + std::to_string(NT) + ", " + std::to_string(EPI) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_gemm_f32_l<RM, NT, EPI>), dim3((unsigned)nblocks), dim3(256), 0, st, a, n_tiles_n, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int RM, int NT>
static int launch_pw_l(const PwArgs& a, hipStream_t st) {
    switch (pw_pick_epi(a)) {
        case EPI_PLAIN: return launch_pw_l_e<RM, NT, EPI_PLAIN>(a, st);
        case EPI_RES: return launch_pw_l_e<RM, NT, EPI_RES>(a, st);
        case EPI_BIAS: return launch_pw_l_e<RM, NT, EPI_BIAS>(a, st);
        default: return launch_pw_l_e<RM, NT, EPI_GENERIC>(a, st);
    }
}

int launch_pointwise_tiled(const PwArgs& a, int force_rm, int force_nt, hipStream_t st) {
    int rm, nt;
    pw_pick_tile(a.M, a.N, &rm, &nt);
    if (force_rm > 0) { rm = force_rm; nt = force_nt; }
#define PW_L(RM_, NT_) if (rm == RM_ && nt == NT_) return launch_pw_l<RM_, NT_>(a, st);
    PW_L(2, 6) PW_L(2, 5) PW_L(2, 4) PW_L(2, 3) PW_L(2, 2) PW_L(2, 1)
    PW_L(1, 6) PW_L(1, 5) PW_L(1, 4) PW_L(1, 3) PW_L(1, 2) PW_L(1, 1)
#undef PW_L
    set_error("pointwise: no tile configuration (%d, %d)", rm, nt);
    return AMS_E_INVALID;
}

}  // namespace ams
