// Fine-tune step, early blocks: the BN statistics of the expand layer from the Gram matrix of the block input.
//
// z_e = x . W_e is linear in the block input x (K = 16 .. 32 channels, no bias), so over any set of pixels
//     sum z_e[c]   = g0 . W_e[:, c]                    g0 = sum_p x[p]           (K)
//     sum z_e[c]^2 = W_e[:, c]^T XX W_e[:, c]          XX = sum_p x[p] x[p]^T    (K x K)
// Round 3 recomputed z_e for all 96 .. 192 expanded channels only to form those two sums (xdw_train_kernel<XT_FWD_STATS>: 20 .. 75 us
// per block, 0.25 ms per 8-frame step) — and XX, g0 themselves were already needed for the expand weight gradient
// (dW_e = A G1 + g0^T B + C (XX W_e), k_xdw_train.hip).  Here ONE pass over x forms XX and g0 and the statistics follow per channel:
//   * xx_f64_kernel: XX on the f64 matrix pipe (v_mfma_f64_16x16x4_f64: the products of f32 values are exact in f64 and the accumulation
//     carries 53 bits), g0 in f64 per lane; a wave walks its own 4-pixel groups, the block's waves are added in a fixed order through LDS:
//     one partial row [KP KP + KP] of doubles per block.  With f32 accumulation the quadratic form below would lose what the cancellation
//     inside w^T XX w costs (the terms x_k w_k of one pixel largely cancel); in f64 the sums are exact to ~1e-13 and the statistics are
//     BETTER than the shifted f32 sums they replace.
//   * xx_reduce_kernel: the partial rows in ascending order -> XX | g0 as doubles (the data-parallel step all-reduces these KP KP + KP
//     doubles instead of 2 C) and as floats (the layout xdw_dwe_kernel reads).
//   * expand_stats_kernel: per expanded channel, in f64: S1 = g0 . w, S2 = w^T XX w, shifted about the centre the other statistics
//     kernels use — sum(z - c) = S1 - n c, sum((z - c)^2) = S2 - 2 c S1 + n c^2 — and handed to the same BnFwdFin functor: scale, shift,
//     saved mean / rstd, moving averages.
#include "kernels.hpp"
#include "bn_fin.hpp"

namespace ams {

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int KC>
__global__ __launch_bounds__(256) void xx_f64_kernel(const float* __restrict__ x, int64_t M, int Cin, double* __restrict__ part) {
    constexpr int KP = 16 * KC;
    constexpr int ROW = KP * KP + KP;
    __shared__ double sred[3][KC * KC * 256 + KC * 64];      // waves 1 .. 3: their accumulators, lane-major
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = tid >> 6;
    // this wave's contiguous range of 4-pixel groups
    const int64_t n_groups = (M + 3) / 4;
    const int64_t waves_total = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t per = (n_groups + waves_total - 1) / waves_total;
    const int64_t g_lo = wid * per, g_hi = g_lo + per < n_groups ? g_lo + per : n_groups;
    f64x4 acc[KC][KC];
    double g0[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        g0[i] = 0.0;
#pragma unroll
        for (int j = 0; j < KC; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    }
    // lane (q, l15): pixel 4 g + q, channels 16 c + l15 — operand A[i = l15][k = q] and B[k = q][j = l15] of the 16x16x4 f64 MFMA at once
    bool cok[KC];
    int coff[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) { cok[c] = 16 * c + l15 < Cin; coff[c] = cok[c] ? 16 * c + l15 : 0; }
    constexpr int NG = 8;                                    // groups (32 pixels) in flight: the loads are 4-byte gathers, latency is what they cost
    for (int64_t g = g_lo; g < g_hi; g += NG) {
        float v[NG][KC];
#pragma unroll
        for (int h = 0; h < NG; ++h) {
            const int64_t p = 4 * (g + h) + q;
            const bool ok = g + h < g_hi && p < M;
            const float* px = x + (ok ? p : 0) * (int64_t)Cin;
#pragma unroll
            for (int c = 0; c < KC; ++c) { const float t = px[coff[c]]; v[h][c] = (ok && cok[c]) ? t : 0.f; }
        }
#pragma unroll
        for (int h = 0; h < NG; ++h) {
            double d[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) { d[c] = (double)v[h][c]; g0[c] += d[c]; }
#pragma unroll
            for (int i = 0; i < KC; ++i)
#pragma unroll
                for (int j = 0; j < KC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[i], d[j], acc[i][j], 0, 0, 0);
        }
    }
    // g0: the four pixel slots (q) of a channel
#pragma unroll
    for (int c = 0; c < KC; ++c) { g0[c] += __shfl_xor(g0[c], 16, 64); g0[c] += __shfl_xor(g0[c], 32, 64); }
    if (wave > 0) {
        double* dst = sred[wave - 1];
#pragma unroll
        for (int i = 0; i < KC; ++i)
#pragma unroll
            for (int j = 0; j < KC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((i * KC + j) * 4 + r) * 64 + lane] = acc[i][j][r];
#pragma unroll
        for (int c = 0; c < KC; ++c) dst[KC * KC * 256 + c * 64 + lane] = g0[c];
    }
    __syncthreads();
    if (wave == 0) {
        double* out = part + (int64_t)blockIdx.x * ROW;
#pragma unroll
        for (int i = 0; i < KC; ++i)
#pragma unroll
            for (int j = 0; j < KC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double s = acc[i][j][r];
                    for (int w = 0; w < 3; ++w) s += sred[w][((i * KC + j) * 4 + r) * 64 + lane];
                    // D layout of the f64 16x16x4 MFMA: register r of lane (q, l15) holds row 4 r + q (A index), column l15 (B index) — not the
                    // 4 q + r of the f32 form (XX is symmetric, so only the row interleave matters; pinned by the kernel-level test)
                    out[(16 * i + 4 * r + q) * KP + 16 * j + l15] = s;
                }
        if (q == 0) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                double s = g0[c];
                for (int w = 0; w < 3; ++w) s += sred[w][KC * KC * 256 + c * 64 + lane];
                out[KP * KP + 16 * c + l15] = s;
            }
        }
    }
}

// out64[i] = sum over the partial rows; out32 = the same as floats.  A block owns 16 outputs: thread (kpart, col) sums every 16th row with all
// its loads in flight, the 16 partials of an output are added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void xx_reduce_kernel(const double* __restrict__ part, int rows, int n, double* __restrict__ out64,
                                                        float* __restrict__ out32) {
    __shared__ double sacc[16][17];
    const int col = threadIdx.x & 15, kpart = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + col;
    double s0 = 0.0, s1 = 0.0;
    if (i < n) {
        int r = kpart;
        for (; r + 16 < rows; r += 32) { s0 += part[(int64_t)r * n + i]; s1 += part[(int64_t)(r + 16) * n + i]; }
        if (r < rows) s0 += part[(int64_t)r * n + i];
    }
    sacc[kpart][col] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 16 && i < n) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sacc[k][col];
        out64[i] = t;
        if (out32) out32[i] = (float)t;
    }
}

// block = 8 channels x 32 k: thread (c, k) forms t_k = sum_k2 XX[k][k2] w[k2] and its share w_k t_k of S2 = w^T XX w and g0_k w_k of S1;
// the 32 shares of a channel are added in ascending k (deterministic)
__global__ __launch_bounds__(256) void expand_stats_kernel(const double* __restrict__ xx, int KP, int Cin, const float* __restrict__ w_exp, int Cexp,
                                                           double* __restrict__ sums, BnFwdFin fin) {
    __shared__ double sxx[32 * 32 + 32];                      // XX [KP][KP] | g0 [KP]
    __shared__ double sw[8][33], sp1[8][33], sp2[8][33];
    for (int e = threadIdx.x; e < KP * KP + KP; e += 256) sxx[e] = xx[e];
    const int cl = threadIdx.x >> 5, k = threadIdx.x & 31;
    const int c = blockIdx.x * 8 + cl;
    const bool live = c < Cexp && k < Cin;
    sw[cl][k] = live ? (double)w_exp[(int64_t)k * Cexp + c] : 0.0;
    __syncthreads();
    double t = 0.0;
    if (live)
        for (int k2 = 0; k2 < Cin; ++k2) t += sxx[k * KP + k2] * sw[cl][k2];
    sp1[cl][k] = live ? sxx[KP * KP + k] * sw[cl][k] : 0.0;
    sp2[cl][k] = sw[cl][k] * t;
    __syncthreads();
    if (k == 0 && c < Cexp) {
        double S1 = 0.0, S2 = 0.0;
        for (int kk = 0; kk < 32; ++kk) { S1 += sp1[cl][kk]; S2 += sp2[cl][kk]; }
        const double ctr = fin.center ? (double)fin.center[c] : 0.0;
        const double s0 = S1 - fin.n * ctr, s1 = S2 - 2.0 * ctr * S1 + fin.n * ctr * ctr;
        if (sums) { sums[c] = s0; sums[Cexp + c] = s1; }
        fin(c, Cexp, s0, s1 < 0.0 ? 0.0 : s1);
    }
}

int xx_stats_blocks(int64_t M) {
    int64_t b = (M + 255) / 256;                             // >= 64 pixels per wave
    if (b > 1024) b = 1024;                                  // four waves per SIMD hide the gathers' latency (256 blocks: 88 us on 1 M pixels; every
                                                             // partial row is KP KP + KP doubles for the second stage)
    if (b < 1) b = 1;
    return (int)b;
}
size_t xx_stats_scratch_doubles(int64_t M, int Cin) {
    const int KP = (Cin + 15) / 16 * 16;
    return (size_t)(xx_stats_blocks(M) + 1) * (size_t)(KP * KP + KP);
}

// XX = x^T x [KP][KP] and g0 = sum x [KP] over the M rows of x [M, Cin] (KP = Cin rounded up to 16, Cin <= 32): f64 in xx64 (KP KP + KP
// doubles), floats in xx32 (may be null); scratch: xx_stats_scratch_doubles doubles
int launch_xx_gram(const float* x, int64_t M, int Cin, double* scratch, double* xx64, float* xx32, hipStream_t st) {
    AMS_REQUIRE(x && scratch && xx64 && M > 0 && Cin >= 4 && Cin <= 32, "xx_gram: bad arguments (Cin=%d)", Cin);
    const int KC = (Cin + 15) / 16, KP = 16 * KC, n = KP * KP + KP;
    const int blocks = xx_stats_blocks(M);
    note_kernel(KC == 1 ? "xx_f64_kernel<1>" : "xx_f64_kernel<2>");
    if (KC == 1) hipLaunchKernelGGL((xx_f64_kernel<1>), dim3(blocks), dim3(256), 0, st, x, M, Cin, scratch);
    else hipLaunchKernelGGL((xx_f64_kernel<2>), dim3(blocks), dim3(256), 0, st, x, M, Cin, scratch);
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL(xx_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, (const double*)scratch, blocks, n, xx64, xx32);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// BN forward coefficients of z = x . w_exp from (XX | g0) over n pixels (all ranks' sums in a data-parallel step); sums [2][Cexp] optional
int launch_expand_stats(const double* xx64, int Cin, const float* w_exp, int Cexp, double n, const float* center, const float* gamma,
                        const float* beta, float eps, float one_minus_decay, float* moving_mean, float* moving_var, float* scale, float* shift,
                        float* save_mean, float* save_rstd, double* sums, hipStream_t st) {
    AMS_REQUIRE(xx64 && w_exp && gamma && beta && scale && shift && Cin >= 4 && Cin <= 32 && Cexp > 0, "expand_stats: bad arguments");
    const int KP = (Cin + 15) / 16 * 16;
    const BnFwdFin fin{n, center, gamma, beta, eps, one_minus_decay, moving_mean, moving_var, scale, shift, save_mean, save_rstd};
    hipLaunchKernelGGL(expand_stats_kernel, dim3(cdiv(Cexp, 8)), dim3(256), 0, st, xx64, KP, Cin, w_exp, Cexp, sums, fin);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams
