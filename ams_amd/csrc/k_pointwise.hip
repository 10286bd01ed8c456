// 1x1 convolutions of the AMS student as exact-f32 MFMA GEMMs (v_mfma_f32_16x16x4_f32, gfx950).
//
//   forward : y[M,N]  = epilogue(x[M,K] @ w[K,N])            (expand / project / head 1x1 convs, K4/K8 of SURVEY §2.2)
//   dgrad   : dx[M,K] = dy[M,N] @ w^T                         (same kernel, w addressed with swapped strides)
//   wgrad   : dw[K,N] = x[M,K]^T @ dy[M,N]                    (split over M, deterministic two-stage reduce)
//
// M = B*H*W pixels is huge (up to 8 x 131 841), K and N are channel counts (16..960): every layer is a skinny
// GEMM whose activations stream from HBM exactly once, so the design is bandwidth-first:
//   * A (activations) never touches LDS: each lane loads float4 = 4 consecutive k of one row straight to VGPRs;
//     16 lanes x 16 B = one 64-byte row segment, 4 such rows per MFMA.  The contraction index may be permuted
//     freely as long as A and B agree, so MFMA step j consumes k = 4*(lane>>4) + j of a 16-wide chunk.
//   * B (weights, <= 1.2 MB, L2 resident) lives in LDS with row pitch 16*NT+4 floats (pitch % 8 == 4 makes the two
//     16-lane halves of a ds_read_b32 group hit disjoint banks): the whole panel for streaming layers (variant S,
//     persistent waves, no barrier in the loop), 32-k stages double-buffered for late layers (variant L).
//   * MFMA operand roles are swapped (weights = A, activations = B) so each lane ends up with 4 consecutive output
//     channels of one pixel: the epilogue (per-image bias, BN scale/shift, ReLU/ReLU6, residual, store) is float4.
#include "pw_common.hpp"

namespace ams {

int launch_pointwise_stream(const PwArgs& a, int force_rm, int force_nt, bool* handled, hipStream_t st);
int launch_pointwise_tiled(const PwArgs& a, int force_rm, int force_nt, hipStream_t st);

// M <= 16 rows (the pooled vector of the image-pooling branch: one row per image).  A tiled MFMA kernel would run a
// single block through K/32 barrier-separated stages (~17 us of pure latency); here every output element is one thread
// walking K with coalesced weight reads, 4 partial sums in flight.
__global__ __launch_bounds__(256) void pw_small_m_kernel(PwArgs a) {
    // block = one row m x 16 output columns; thread (n = t & 15, part = t >> 4) sums every 16th k, then a fixed-order
    // LDS reduction over the 16 parts (deterministic)
    __shared__ float sacc[16][17];
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    const int m = blockIdx.y;
    float s0 = 0.f, s1 = 0.f;
    if (n < a.N) {
        const float* xr = a.x + (int64_t)m * a.ldx;
        const float* wp = a.w + (int64_t)n * a.w_sn;
        int k = part;
        for (; k + 16 < a.Kw; k += 32) {
            s0 = fmaf(xr[k], wp[(int64_t)k * a.w_sk], s0);
            s1 = fmaf(xr[k + 16], wp[(int64_t)(k + 16) * a.w_sk], s1);
        }
        if (k < a.Kw) s0 = fmaf(xr[k], wp[(int64_t)k * a.w_sk], s0);
    }
    sacc[part][threadIdx.x & 15] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 16 && n < a.N) {
        float v = 0.f;
        for (int p = 0; p < 16; ++p) v += sacc[p][threadIdx.x];
        if (a.img_bias) v += a.img_bias[(m / a.rows_per_img) * a.N + n];
        v = v * (a.scale ? a.scale[n] : 1.f) + (a.shift ? a.shift[n] : 0.f);
        v = apply_act(v, a.act);
        if (a.res) v += a.res[(int64_t)m * a.ldr + n];
        a.y[(int64_t)m * a.ldy + n] = v;
    }
}

int pointwise_materialize_x(const PwArgs& a, PwArgs* b, hipStream_t st) {
    AMS_REQUIRE(a.x_mode == 1 || a.x_mode == 2, "pointwise: unknown operand transform %d", a.x_mode);
    AMS_REQUIRE(a.x_tmp && a.ldx == a.K && a.K % 4 == 0 && a.x_v0 && a.x_v1, "pointwise: this kernel cannot transform its operand on load and no dense x_tmp was given");
    if (a.x_mode == 1) RUN_RC(launch_bn_act(a.x, a.M, a.K, a.x_v0, a.x_v1, a.x_act, nullptr, a.x_tmp, st));
    else {
        AMS_REQUIRE(a.x_v2 && a.x2 && a.x_act == AMS_ACT_NONE, "pointwise: operand transform 2 needs (A, B, C), z and no activation");
        RUN_RC(launch_bn_bwd_apply(a.x, a.x2, a.M, a.K, a.x_v0, a.x_v1, AMS_ACT_NONE, a.x_v0, a.x_v1, a.x_v2, a.x_tmp, st));
    }
    *b = a;
    b->x = a.x_tmp; b->x_mode = 0; b->x_tmp = nullptr;
    return AMS_OK;
}

int launch_pointwise(const PwArgs& a0, hipStream_t st) {
    PwArgs a = a0;
    if (a.red_rows_out) *a.red_rows_out = 0;               // set by the kernels that can fuse the column reduction (PwArgs::red_mode)
    AMS_REQUIRE(a.M > 0 && a.K > 0 && a.N > 0, "pointwise: empty problem M=%lld K=%d N=%d", (long long)a.M, a.K, a.N);
    AMS_REQUIRE(a.Kw > 0 && a.Kw <= a.K, "pointwise: Kw=%d must be in 1..K=%d", a.Kw, a.K);
    AMS_REQUIRE(a.K % 4 == 0 && a.ldx % 4 == 0, "pointwise: K (%d) and ldx (%d) must be multiples of 4", a.K, a.ldx);
    AMS_REQUIRE((reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "pointwise: x must be 16-byte aligned");
    if (a.M <= 16) {
        if (a.x_mode != 0) { PwArgs b; RUN_RC(pointwise_materialize_x(a, &b, st)); a = b; }
        note_kernel("pw_small_m_kernel");
        hipLaunchKernelGGL(pw_small_m_kernel, dim3(cdiv(a.N, 16), (unsigned)a.M), dim3(256), 0, st, a);
        AMS_CHECK_LAUNCH();
        return AMS_OK;
    }
    const char force = knobs().pw_force;                  // tuning knob AMS_PW_FORCE = "<s|l>,<RM>,<NT>"
    const int frm = knobs().pw_rm, fnt = knobs().pw_nt;
    if ((a.M >= 32768 && force != 'l') || force == 's') {
        bool handled = false;
        const int rc = launch_pointwise_stream(a, frm, force == 's' ? fnt : 0, &handled, st);     // applies PwArgs::x_mode 1 itself
        if (rc || handled) return rc;
    }
    if (a.x_mode != 0) { PwArgs b; RUN_RC(pointwise_materialize_x(a, &b, st)); a = b; }
    return launch_pointwise_tiled(a, force == 'l' ? frm : 0, fnt, st);
}

// =========================================================================================================
// wgrad: dw[K,N] = sum_m x[m,K]^T dy[m,N].  Contraction over M (pixels).  MFMA 16x16x4: lane (i = lane&15,
// kk = lane>>4) supplies A[i][kk] and B[kk][j]; here kk indexes 4 consecutive pixels and i / j index channels.
// A float4 load along channels gives lane i the channels {4i .. 4i+3} of pixel kk, so MFMA (s,u) accumulates the
// output sub-matrix rows {k0 + 4i + s}, cols {n0 + 4j + u}: a VA*16 x VB*16 tile from one vector load per side.
// Each wave walks its own pixels; the block's 4 waves and the grid's M-splits are reduced in a fixed order
// (LDS, then a second kernel over the split partials) so the result is run-to-run deterministic.
// =========================================================================================================
template <int V>
struct VecLd;
template <>
struct VecLd<4> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[4]) {
        const float4 v = ld4(p);
        o[0] = ok ? v.x : 0.f; o[1] = ok ? v.y : 0.f; o[2] = ok ? v.z : 0.f; o[3] = ok ? v.w : 0.f;
    }
};
template <>
struct VecLd<2> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[2]) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        o[0] = ok ? v.x : 0.f; o[1] = ok ? v.y : 0.f;
    }
};
template <>
struct VecLd<1> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[1]) { const float v = *p; o[0] = ok ? v : 0.f; }
};

// XD: the operand transforms of WgArgs (x_mode 1: BN + activation on x; dy_mode 2: second half of BN backward on dy) — a lane's channels
// are the same for every pixel it loads, so the per-channel vectors live in registers; same unfused arithmetic as the elementwise passes
template <int VA, int VB, bool XD = false>
__global__ __launch_bounds__(256) void pw_wgrad_f32(WgArgs a, int tiles_k, int tiles_n, int64_t rows_per_split) {
    __shared__ float sRed[3 * 64 * VA * VB * 4];   // partial tiles of waves 1..3
    const int tile = blockIdx.y;
    const int tk = tile / tiles_n, tn = tile % tiles_n;
    const int split = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, q = lane >> 4;
    const int k0 = tk * 16 * VA, n0 = tn * 16 * VB;
    const int64_t m_begin = split * rows_per_split;
    int64_t m_end = m_begin + rows_per_split;
    if (m_end > a.M) m_end = a.M;

    f32x4 acc[VA][VB];
#pragma unroll
    for (int s = 0; s < VA; ++s)
#pragma unroll
        for (int u = 0; u < VB; ++u) acc[s][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ka = k0 + VA * l15, nb = n0 + VB * l15;
    const bool ka_ok = ka < a.K, nb_ok = nb < a.N;        // K, N are multiples of VA / VB for the chosen instantiation
    const int kac = ka_ok ? ka : 0, nbc = nb_ok ? nb : 0;
    float xs[XD ? VA : 1], xh[XD ? VA : 1], dA[XD ? VB : 1], dB[XD ? VB : 1], dC[XD ? VB : 1];
    if constexpr (XD) {
#pragma unroll
        for (int s = 0; s < VA; ++s) { xs[s] = a.x_mode == 1 ? a.x_v0[kac + s] : 1.f; xh[s] = a.x_mode == 1 ? a.x_v1[kac + s] : 0.f; }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            dA[u] = a.dy_mode == 2 ? a.dy_v0[nbc + u] : 1.f; dB[u] = a.dy_mode == 2 ? a.dy_v1[nbc + u] : 0.f; dC[u] = a.dy_mode == 2 ? a.dy_v2[nbc + u] : 0.f;
        }
    }
    // each wave takes every 4th group of 4 pixels; 2 groups in flight for latency hiding
    // (loop bounds are wave-uniform: an MFMA must be issued by the whole wave)
    for (int64_t mg = m_begin + wave * 4; mg < m_end; mg += 32) {
        float xa[2][VA], yb[2][VB];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t mm = mg + q + h * 16;
            const bool ok = mm < m_end;
            const int64_t mc = ok ? mm : m_end - 1;                    // clamped row, zeroed by select: no exec-masked loads
            VecLd<VA>::ld(a.x + mc * (int64_t)a.ldx + kac, ok && ka_ok, xa[h]);
            VecLd<VB>::ld(a.dy + mc * (int64_t)a.ldy + nbc, ok && nb_ok, yb[h]);
            if constexpr (XD) {
                if (a.x_mode == 1) {
#pragma unroll
                    for (int s = 0; s < VA; ++s) { const float y = apply_act(xa[h][s] * xs[s] + xh[s], a.x_act); xa[h][s] = (ok && ka_ok) ? y : 0.f; }
                }
                if (a.dy_mode == 2) {
                    float zz[VB];
                    VecLd<VB>::ld(a.dy2 + mc * (int64_t)a.ldy + nbc, ok && nb_ok, zz);
#pragma unroll
                    for (int u = 0; u < VB; ++u) { const float y = (dA[u] * yb[h][u] + dB[u]) + dC[u] * zz[u]; yb[h][u] = (ok && nb_ok) ? y : 0.f; }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < VA; ++s)
#pragma unroll
                for (int u = 0; u < VB; ++u)
                    acc[s][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[h][s], yb[h][u], acc[s][u], 0, 0, 0);
    }
    // reduce the 4 waves in fixed order through LDS
    if (wave > 0) {
        float* dst = sRed + (wave - 1) * 64 * VA * VB * 4;
#pragma unroll
        for (int s = 0; s < VA; ++s)
#pragma unroll
            for (int u = 0; u < VB; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[((s * VB + u) * 4 + i) * 64 + lane] = acc[s][u][i];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.scratch + ((int64_t)split * a.K) * a.N;
#pragma unroll
        for (int s = 0; s < VA; ++s)
#pragma unroll
            for (int u = 0; u < VB; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[s][u][i];
                    for (int w = 0; w < 3; ++w) v += sRed[w * 64 * VA * VB * 4 + ((s * VB + u) * 4 + i) * 64 + lane];
                    // C/D layout: col j = lane&15, row i' = 4*(lane>>4) + i  ->  channel k = k0 + VA*i' + s, n = n0 + VB*j + u
                    const int kk = k0 + VA * (4 * q + i) + s;
                    const int nn = n0 + VB * l15 + u;
                    if (kk < a.K && nn < a.N) out[(int64_t)kk * a.N + nn] = v;
                }
    }
}

static int wgrad_splits(int64_t M, int K, int N) {
    // enough M-splits to fill the chip (~4 blocks per CU overall), at least 256 rows each
    const int tiles = cdiv(K, 64) * cdiv(N, 64);
    int64_t want = (1024 + tiles - 1) / tiles;
    int64_t max_by_rows = (M + 255) / 256;
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    return (int)want;
}

size_t pointwise_wgrad_scratch(int64_t M, int K, int N) {
    const size_t f32 = (size_t)wgrad_splits(M, K, N) * K * N, x6 = (size_t)wgrad_x6_splits(M, K, N) * K * N;
    return f32 > x6 ? f32 : x6;
}

template <int VA, int VB>
static int launch_wg_t(const WgArgs& a, int splits, hipStream_t st) {
    const int tiles_k = cdiv(a.K, 16 * VA), tiles_n = cdiv(a.N, 16 * VB);
    int64_t rows = cdiv64(a.M, splits);
    rows = (rows + 3) / 4 * 4;
    static const std::string nm = "pw_wgrad_f32<" + std::to_string(VA) + ", " + std::to_string(VB) + ">";
    note_kernel(nm.c_str());
    if (a.x_mode != 0 || a.dy_mode != 0)
        hipLaunchKernelGGL((pw_wgrad_f32<VA, VB, true>), dim3(splits, tiles_k * tiles_n), dim3(256), 0, st, a, tiles_k, tiles_n, rows);
    else
    hipLaunchKernelGGL((pw_wgrad_f32<VA, VB>), dim3(splits, tiles_k * tiles_n), dim3(256), 0, st, a, tiles_k, tiles_n, rows);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_pointwise_wgrad(const WgArgs& a, hipStream_t st) {
    AMS_REQUIRE(a.M > 0 && a.K > 0 && a.N > 0, "wgrad: empty problem");
    AMS_REQUIRE(a.x_mode == 0 || (a.x_mode == 1 && a.x_v0 && a.x_v1), "wgrad: operand transform of x: mode %d or missing vectors", a.x_mode);
    AMS_REQUIRE(a.dy_mode == 0 || (a.dy_mode == 2 && a.dy_v0 && a.dy_v1 && a.dy_v2 && a.dy2), "wgrad: operand transform of dy: mode %d or missing operands", a.dy_mode);
    if (a.allow_split && pointwise_wgrad_x6_applies(a.M, a.K, a.N, a.ldx, a.ldy)) {
        const int sp = wgrad_x6_splits(a.M, a.K, a.N);
        AMS_REQUIRE(a.scratch_floats >= (size_t)sp * a.K * a.N, "wgrad: scratch too small (%zu < %zu)", a.scratch_floats,
                    (size_t)sp * a.K * a.N);
        int rc = launch_pointwise_wgrad_x6(a, sp, st);
        if (rc) return rc;
        return launch_reduce_splits(a.scratch, sp, (int64_t)a.K * a.N, a.dw, st);
    }
    const int splits = wgrad_splits(a.M, a.K, a.N);
    AMS_REQUIRE(a.scratch_floats >= (size_t)splits * a.K * a.N, "wgrad: scratch too small (%zu < %zu)", a.scratch_floats,
                (size_t)splits * a.K * a.N);
    // vector width per side: widest of 4/2/1 that divides the channel count and keeps the row stride aligned
    auto vec = [](int c, int ld) { return (c % 4 == 0 && ld % 4 == 0) ? 4 : (c % 2 == 0 && ld % 2 == 0) ? 2 : 1; };
    int va = vec(a.K, a.ldx), vb = vec(a.N, a.ldy);
    if (a.K <= 16) va = 1; else if (a.K <= 32 && va > 2) va = 2;      // do not waste MFMA rows on absent channels
    if (a.N <= 16) vb = 1; else if (a.N <= 32 && vb > 2) vb = 2;
    int rc;
#define WG_CASE(A_, B_) if (va == A_ && vb == B_) { rc = launch_wg_t<A_, B_>(a, splits, st); if (rc) return rc; } else
    WG_CASE(4, 4) WG_CASE(4, 2) WG_CASE(4, 1) WG_CASE(2, 4) WG_CASE(2, 2) WG_CASE(2, 1) WG_CASE(1, 4) WG_CASE(1, 2) WG_CASE(1, 1)
    { set_error("wgrad: no instantiation"); return AMS_E_INVALID; }
#undef WG_CASE
    return launch_reduce_splits(a.scratch, splits, (int64_t)a.K * a.N, a.dw, st);
}

}  // namespace ams
