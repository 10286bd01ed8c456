// BatchNorm pieces (training-mode statistics, coefficient folding, backward), pooling, column reductions and
// the fused Adam update.  All HBM-bound streaming kernels: float4 per lane, lanes along channels first so a
// wave touches 1 KiB of contiguous NHWC memory per instruction.  Every reduction is two-stage with a fixed
// summation order (block partials -> f64 finalize), so results are run-to-run deterministic.
#include <hip/hip_fp16.h>

#include "kernels.hpp"
#include "bn_fin.hpp"

namespace ams {

// ---------------------------------------------------------------------------------------------------------
// generic column reduction over rows of a [groups*rows, C] tensor: NQ quantities per element, float4 lanes.
// grid = (chunks, groups); block = CG x slots threads.  part[(group*chunks + chunk)][NQ][C]
// ---------------------------------------------------------------------------------------------------------
struct ColGeom {
    int C, CG, slots, chunks, groups;
    int64_t rows_per_group, rows_per_chunk;
    int ldx;
};

static ColGeom col_geom(int64_t rows_per_group, int groups, int C, int ldx, int max_chunks) {
    ColGeom g;
    g.C = C; g.CG = C / 4; g.ldx = ldx; g.groups = groups; g.rows_per_group = rows_per_group;
    g.slots = 256 / g.CG < 1 ? 1 : 256 / g.CG;
    int64_t chunks = cdiv64(rows_per_group, (int64_t)g.slots * 16);     // >= 16 rows per thread
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    g.chunks = (int)chunks;
    g.rows_per_chunk = cdiv64(rows_per_group, chunks);
    return g;
}

// Each functor: prepare(c0) caches this thread's per-channel constants once; operator() maps one row to (q0, q1).
struct OpStats {          // q0 = z - center, q1 = (z - center)^2
    const float* z; const float* center;
    float4 c;
    __device__ __forceinline__ void prepare(int c0) { c = center ? ld4(center + c0) : make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void operator()(int64_t row, int c0, int ld, float4& q0, float4& q1) const {
        const float4 v = ld4(z + row * ld + c0);
        q0 = sub4_pk(v, c);                  // packed f32: the same IEEE operations, half the instructions
        q1 = mul4_pk(q0, q0);
    }
};

__device__ __forceinline__ float act_grad_mask(float y, int act) {
    if (act == AMS_ACT_RELU6) return (y > 0.f && y < 6.f) ? 1.f : 0.f;     // Relu6Grad: 0 < features < 6
    if (act == AMS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}

struct OpBnBwd {          // q0 = dy, q1 = dy * xhat ; dy = da * act'(z*scale+shift), xhat = (z-mean)*rstd
    const float* da; const float* z; const float* scale; const float* shift; const float* mean; const float* rstd; int act;
    float4 sc, sh, mu, rs;
    __device__ __forceinline__ void prepare(int c0) { sc = ld4(scale + c0); sh = ld4(shift + c0); mu = ld4(mean + c0); rs = ld4(rstd + c0); }
    __device__ __forceinline__ void operator()(int64_t row, int c0, int ld, float4& q0, float4& q1) const {
        const float4 g = ld4(da + row * ld + c0), v = ld4(z + row * ld + c0);
        const float4 y = muladd4_pk(v, sc, sh);
        const float4 mk = make_float4(act_grad_mask(y.x, act), act_grad_mask(y.y, act), act_grad_mask(y.z, act), act_grad_mask(y.w, act));
        q0 = mul4_pk(g, mk);
        q1 = mul4_pk(mul4_pk(q0, sub4_pk(v, mu)), rs);
    }
};

struct OpSum {            // q0 = x
    const float* x;
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void operator()(int64_t row, int c0, int ld, float4& q0, float4& q1) const {
        q0 = ld4(x + row * ld + c0);
        q1 = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

template <int NQ, class Op>
__global__ __launch_bounds__(256) void col_reduce_kernel(Op op, ColGeom g, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sred[];       // [slots][NQ][C]
    const int cg = threadIdx.x % g.CG, slot = threadIdx.x / g.CG;
    const int c0 = cg * 4;
    const int chunk = blockIdx.x, group = blockIdx.y;
    const int64_t r_begin = (int64_t)chunk * g.rows_per_chunk;
    int64_t r_end = r_begin + g.rows_per_chunk;
    if (r_end > g.rows_per_group) r_end = g.rows_per_group;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (slot < g.slots) {
        op.prepare(c0);
        const int64_t base = (int64_t)group * g.rows_per_group;
        int64_t r = r_begin + slot;
        // 4 rows in flight per thread: the loads are independent, the adds keep the original row order
        for (; r + 3 * g.slots < r_end; r += 4 * g.slots) {
            float4 q0[4], q1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) op(base + r + u * g.slots, c0, g.ldx, q0[u], q1[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = add4_pk(a0, q0[u]);
                if (NQ > 1) a1 = add4_pk(a1, q1[u]);
            }
        }
        for (; r < r_end; r += g.slots) {
            float4 q0, q1;
            op(base + r, c0, g.ldx, q0, q1);
            a0 = add4_pk(a0, q0);
            if (NQ > 1) a1 = add4_pk(a1, q1);
        }
        st4(sred + ((int64_t)slot * NQ + 0) * g.C + c0, a0);
        if (NQ > 1) st4(sred + ((int64_t)slot * NQ + 1) * g.C + c0, a1);
    }
    __syncthreads();
    float* out = part + ((int64_t)group * g.chunks + chunk) * NQ * g.C;
    for (int e = threadIdx.x; e < NQ * g.C; e += blockDim.x) {
        float s = 0.f;
        for (int sl = 0; sl < g.slots; ++sl) s += sred[(int64_t)sl * NQ * g.C + e];
        out[e] = s;
    }
}

// second stage: out[group][e] = alpha * sum_chunk part[group][chunk][e], e = (q, c) flattened.  One block sums 16
// columns: thread t owns column (t & 15) and every 16th chunk starting at (t >> 4); the 16 per-thread f64 partials of a
// column are then added in a fixed order, so the result does not depend on scheduling.
template <typename TOut>
__global__ __launch_bounds__(256) void col_finalize_kernel(const float* __restrict__ part, int chunks, int per_group /* NQ*C */,
                                                           int groups, double alpha, TOut* __restrict__ out) {
    __shared__ double sacc[16][17];
    const int col16 = blockIdx.x, group = blockIdx.y;
    const int e = col16 * 16 + (threadIdx.x & 15), kpart = threadIdx.x >> 4;
    double s = 0.0;
    if (e < per_group) {
        const float* p = part + (int64_t)group * chunks * per_group + e;
        // 8 loads in flight per thread (the loop is latency-bound)
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = kpart;
        for (; k + 112 < chunks; k += 128) {
            const float v0 = p[(int64_t)k * per_group], v1 = p[(int64_t)(k + 16) * per_group];
            const float v2 = p[(int64_t)(k + 32) * per_group], v3 = p[(int64_t)(k + 48) * per_group];
            const float v4 = p[(int64_t)(k + 64) * per_group], v5 = p[(int64_t)(k + 80) * per_group];
            const float v6 = p[(int64_t)(k + 96) * per_group], v7 = p[(int64_t)(k + 112) * per_group];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; k + 16 < chunks; k += 32) { s0 += p[(int64_t)k * per_group]; s1 += p[(int64_t)(k + 16) * per_group]; }
        if (k < chunks) s0 += p[(int64_t)k * per_group];
        s = (s0 + s1) + (s2 + s3);
    }
    sacc[kpart][threadIdx.x & 15] = s;
    __syncthreads();
    if (threadIdx.x < 16 && col16 * 16 + threadIdx.x < per_group) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sacc[k][threadIdx.x];
        out[(int64_t)group * per_group + col16 * 16 + threadIdx.x] = (TOut)(t * alpha);
    }
}

// Second stage fused with the per-channel BN arithmetic (single-GPU path: no all-reduce sits between the sums and their
// use).  part is [chunks][2][C]; a block of 512 threads owns 16 channels: thread t = (q = t >> 8, kpart = (t >> 4) & 15, col = t & 15).
template <class Fin>
__global__ __launch_bounds__(512) void col_finalize_bn_kernel(const float* __restrict__ part, int chunks, int C,
                                                              double* __restrict__ sums, Fin fin, int64_t row_stride = 0) {
    __shared__ double sacc[2][16][17];
    const int col = threadIdx.x & 15, kpart = (threadIdx.x >> 4) & 15, q = threadIdx.x >> 8;
    const int c = blockIdx.x * 16 + col;
    double s = 0.0;
    if (c < C) {
        const int64_t per = row_stride > 0 ? row_stride : 2 * (int64_t)C;       // floats between the partial rows
        const float* p = part + (int64_t)q * C + c;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = kpart;
        for (; k + 112 < chunks; k += 128) {          // 8 loads in flight
            const float v0 = p[(int64_t)k * per], v1 = p[(int64_t)(k + 16) * per];
            const float v2 = p[(int64_t)(k + 32) * per], v3 = p[(int64_t)(k + 48) * per];
            const float v4 = p[(int64_t)(k + 64) * per], v5 = p[(int64_t)(k + 80) * per];
            const float v6 = p[(int64_t)(k + 96) * per], v7 = p[(int64_t)(k + 112) * per];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; k < chunks; k += 16) s0 += p[(int64_t)k * per];
        s = (s0 + s1) + (s2 + s3);
    }
    sacc[q][kpart][col] = s;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < 16; ++k) { t0 += sacc[0][k][col]; t1 += sacc[1][k][col]; }
        sums[c] = t0;
        sums[C + c] = t1;
        fin(c, C, t0, t1);
    }
}

constexpr int kMaxChunks = 1024;

template <int NQ, class Op, typename TOut>
static int run_col_reduce(Op op, int64_t rows_per_group, int groups, int C, int ldx, double alpha, float* scratch, TOut* out,
                          hipStream_t st) {
    AMS_REQUIRE(C % 4 == 0 && C / 4 <= 256 && ldx % 4 == 0, "column reduce: C=%d ld=%d must be multiples of 4 (C <= 1024)", C, ldx);
    const ColGeom g = col_geom(rows_per_group, groups, C, ldx, kMaxChunks);
    const size_t lds = (size_t)g.slots * NQ * C * sizeof(float);
    hipLaunchKernelGGL((col_reduce_kernel<NQ, Op>), dim3(g.chunks, groups), dim3(g.CG * g.slots), lds, st, op, g, scratch);
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL((col_finalize_kernel<TOut>), dim3(cdiv(NQ * C, 16), groups), dim3(256), 0, st, scratch, g.chunks, NQ * C, groups,
                       alpha, out);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

size_t colstats_scratch(int64_t M, int C) { (void)M; return (size_t)kMaxChunks * 2 * C; }

int launch_colstats(const float* z, int64_t M, int C, const float* center, double* sums, float* scratch, hipStream_t st) {
    OpStats op{z, center, {}};
    note_kernel("col_reduce_kernel<2, OpStats>");
    return run_col_reduce<2, OpStats, double>(op, M, 1, C, C, 1.0, scratch, sums, st);
}

template <class Op, class Fin>
static int run_col_reduce_bn(Op op, int64_t M, int C, float* scratch, double* sums, Fin fin, hipStream_t st) {
    AMS_REQUIRE(C % 4 == 0 && C / 4 <= 256, "column reduce: C=%d must be a multiple of 4 (<= 1024)", C);
    const ColGeom g = col_geom(M, 1, C, C, kMaxChunks);
    const size_t lds = (size_t)g.slots * 2 * C * sizeof(float);
    hipLaunchKernelGGL((col_reduce_kernel<2, Op>), dim3(g.chunks, 1), dim3(g.CG * g.slots), lds, st, op, g, scratch);
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL((col_finalize_bn_kernel<Fin>), dim3(cdiv(C, 16)), dim3(512), 0, st, scratch, g.chunks, C, sums, fin, (int64_t)0);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// launch_colstats + launch_bn_finalize in two launches instead of three (no cross-rank sum in between)
int launch_colstats_bn(const float* z, int64_t M, int C, const float* center, double* sums, float* scratch, double n,
                       const float* gamma, const float* beta, float eps, float one_minus_decay, float* moving_mean,
                       float* moving_var, float* scale, float* shift, float* save_mean, float* save_rstd, hipStream_t st) {
    OpStats op{z, center, {}};
    note_kernel("col_reduce_kernel<2, OpStats>");
    const BnFwdFin fin{n, center, gamma, beta, eps, one_minus_decay, moving_mean, moving_var, scale, shift, save_mean, save_rstd};
    return run_col_reduce_bn(op, M, C, scratch, sums, fin, st);
}

// second stages alone, over partial rows [rows][2][C] that another kernel produced (k_xdw_train.hip), `row_stride` floats apart
int launch_bn_fwd_finalize_partials(const float* part, int rows, int64_t row_stride, int C, double* sums, double n, const float* center,
                                    const float* gamma, const float* beta, float eps, float one_minus_decay, float* moving_mean,
                                    float* moving_var, float* scale, float* shift, float* save_mean, float* save_rstd, hipStream_t st) {
    const BnFwdFin fin{n, center, gamma, beta, eps, one_minus_decay, moving_mean, moving_var, scale, shift, save_mean, save_rstd};
    hipLaunchKernelGGL((col_finalize_bn_kernel<BnFwdFin>), dim3(cdiv(C, 16)), dim3(512), 0, st, part, rows, C, sums, fin, row_stride);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}
int launch_bn_bwd_finalize_partials(const float* part, int rows, int64_t row_stride, int C, double* sums, double n, const float* gamma,
                                    const float* mean, const float* rstd, float* coefA, float* coefB, float* coefC, float* dgamma,
                                    float* dbeta, hipStream_t st) {
    const BnBwdFin fin{n, gamma, mean, rstd, coefA, coefB, coefC, dgamma, dbeta};
    hipLaunchKernelGGL((col_finalize_bn_kernel<BnBwdFin>), dim3(cdiv(C, 16)), dim3(512), 0, st, part, rows, C, sums, fin, row_stride);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}
struct SumsOnlyFin { __device__ void operator()(int, int, double, double) const {} };
// sums[2][C] (f64) only: the data-parallel step all-reduces them before the per-channel arithmetic
int launch_partials_to_sums(const float* part, int rows, int64_t row_stride, int C, double* sums, hipStream_t st) {
    hipLaunchKernelGGL((col_finalize_bn_kernel<SumsOnlyFin>), dim3(cdiv(C, 16)), dim3(512), 0, st, part, rows, C, sums, SumsOnlyFin{}, row_stride);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// launch_bn_bwd_reduce + launch_bn_param_grads + launch_bn_bwd_coef in two launches instead of four
int launch_bn_bwd_reduce_coef(const float* da, const float* z, int64_t M, int C, const float* scale, const float* shift, int act,
                              const float* mean, const float* rstd, double* sums, float* scratch, double n, const float* gamma,
                              float* coefA, float* coefB, float* coefC, float* dgamma, float* dbeta, hipStream_t st) {
    OpBnBwd op{da, z, scale, shift, mean, rstd, act, {}, {}, {}, {}};
    note_kernel("col_reduce_kernel<2, OpBnBwd>");
    const BnBwdFin fin{n, gamma, mean, rstd, coefA, coefB, coefC, dgamma, dbeta};
    return run_col_reduce_bn(op, M, C, scratch, sums, fin, st);
}

int launch_bn_bwd_reduce(const float* da, const float* z, int64_t M, int C, const float* scale, const float* shift,
                         int act, const float* mean, const float* rstd, double* sums, float* scratch, hipStream_t st) {
    OpBnBwd op{da, z, scale, shift, mean, rstd, act, {}, {}, {}, {}};
    note_kernel("col_reduce_kernel<2, OpBnBwd>");
    return run_col_reduce<2, OpBnBwd, double>(op, M, 1, C, C, 1.0, scratch, sums, st);
}

int launch_colsum(const float* x, int64_t M, int C, int ldx, float* out, float* scratch, hipStream_t st) {
    OpSum op{x};
    note_kernel("col_reduce_kernel<1, OpSum>");
    return run_col_reduce<1, OpSum, float>(op, M, 1, C, ldx, 1.0, scratch, out, st);
}

// per-image column sums (scratch: B * kMaxChunks * C floats); used for the global mean (alpha = 1/HW) and its gradient
static int image_colsum_impl(const float* x, int B, int64_t HW, int C, int ldx, double alpha, float* out, float* scratch,
                             hipStream_t st) {
    OpSum op{x};
    note_kernel("col_reduce_kernel<1, OpSum>");
    return run_col_reduce<1, OpSum, float>(op, HW, B, C, ldx, alpha, scratch, out, st);
}

size_t image_colsum_scratch(int B, int C) { return (size_t)B * kMaxChunks * C; }

int launch_global_mean(const float* x, int B, int64_t HW, int C, float* y, float* scratch, hipStream_t st) {
    return image_colsum_impl(x, B, HW, C, C, 1.0 / (double)HW, y, scratch, st);
}

int launch_image_colsum(const float* x, int B, int64_t HW, int C, int ldx, float* out, float* scratch, hipStream_t st) {
    return image_colsum_impl(x, B, HW, C, ldx, 1.0, out, scratch, st);
}

// ---------------------------------------------------------------------------------------------------------
// BN coefficient kernels (per channel, tiny)
// ---------------------------------------------------------------------------------------------------------
// FusedBatchNormV3(is_training=True): normalise with the biased variance, feed the
// UNBIASED one (var*n/max(n-1,1)) to AssignMovingAvg: moving -= (moving - stat) * (1 - decay), f32 arithmetic.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double n, int C, const float* __restrict__ center,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float one_minus_decay, float* moving_mean, float* moving_var, float* scale, float* shift,
                                   float* save_mean, float* save_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double ctr = center ? (double)center[c] : 0.0;
    const double d1 = sums[c] / n, d2 = sums[C + c] / n;
    double var = d2 - d1 * d1;
    if (var < 0) var = 0;
    const float mean = (float)(ctr + d1);
    const float varf = (float)var;
    const float rstd = 1.0f / sqrtf(varf + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) { save_mean[c] = mean; save_rstd[c] = rstd; }
    if (moving_mean) {
        const float unbiased = (float)(var * (n / (n > 1.5 ? n - 1.0 : 1.0)));
        moving_mean[c] = moving_mean[c] - (moving_mean[c] - mean) * one_minus_decay;
        moving_var[c] = moving_var[c] - (moving_var[c] - unbiased) * one_minus_decay;
    }
}

int launch_bn_finalize(const double* sums, double n, int C, const float* center, const float* gamma, const float* beta,
                       float eps, float one_minus_decay, float* moving_mean, float* moving_var, float* scale,
                       float* shift, float* save_mean, float* save_rstd, hipStream_t st) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, sums, n, C, center, gamma, beta, eps,
                       one_minus_decay, moving_mean, moving_var, scale, shift, save_mean, save_rstd);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                               float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.0f / sqrtf(var[c] + eps));
    scale[c] = sc;
    shift[c] = beta[c] - mean[c] * sc;
}

int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                   float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, gamma, beta, mean, var, eps, C, scale, shift);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// dz = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)) rewritten as dz = A*dy + B + C*z per channel
__global__ void bn_bwd_coef_kernel(const double* __restrict__ sums, double n, int C, const float* gamma, const float* mean,
                                   const float* rstd, float* coefA, float* coefB, float* coefC, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double sdy = sums[c], sdyx = sums[C + c];
    const double A = (double)gamma[c] * rstd[c];
    const double k = A * (sdyx / n) * rstd[c];
    coefA[c] = (float)A;
    coefC[c] = (float)(-k);
    coefB[c] = (float)(-A * (sdy / n) + k * mean[c]);
    if (dgamma) { dgamma[c] = (float)sdyx; dbeta[c] = (float)sdy; }
}

// dgamma = sum(dy * xhat), dbeta = sum(dy) from the LOCAL sums (data-parallel ranks add theirs up in the gradient all-reduce)
__global__ void bn_param_grads_kernel(const double* __restrict__ sums, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)sums[c];
    dgamma[c] = (float)sums[C + c];
}

int launch_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta, hipStream_t st) {
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, sums, C, dgamma, dbeta);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_bn_bwd_coef(const double* sums, double n, int C, const float* gamma, const float* mean, const float* rstd,
                       float* coefA, float* coefB, float* coefC, float* dgamma, float* dbeta, hipStream_t st) {
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, sums, n, C, gamma, mean, rstd, coefA, coefB,
                       coefC, dgamma, dbeta);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// elementwise passes over [M, C]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ z, int64_t n4, int C4,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int act, const float* __restrict__ res, float* __restrict__ a) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % C4) * 4;
        const float4 v = ld4(z + i * 4), sc = ld4(scale + c0), sh = ld4(shift + c0);
        const float4 y = muladd4_pk(v, sc, sh);            // packed f32: the same IEEE operations, half the instructions
        float4 o = make_float4(apply_act(y.x, act), apply_act(y.y, act), apply_act(y.z, act), apply_act(y.w, act));
        if (res) o = add4_pk(o, ld4(res + i * 4));
        st4(a + i * 4, o);
    }
}

static int stream_grid(int64_t n_items) {
    int64_t g = cdiv64(n_items, 256);
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

int launch_bn_act(const float* z, int64_t M, int C, const float* scale, const float* shift, int act, const float* res,
                  float* a, hipStream_t st) {
    AMS_REQUIRE(C % 4 == 0, "bn_act: C=%d must be a multiple of 4", C);
    const int64_t n4 = M * C / 4;
    note_kernel("bn_act_kernel");
    hipLaunchKernelGGL(bn_act_kernel, dim3(stream_grid(n4)), dim3(256), 0, st, z, n4, C / 4, scale, shift, act, res, a);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// (da and dz may be the same tensor: every element is read before it is written, by the same thread)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* da, const float* __restrict__ z,
                                                           int64_t n4, int C4, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act,
                                                           const float* __restrict__ cA, const float* __restrict__ cB,
                                                           const float* __restrict__ cC, float* dz) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % C4) * 4;
        const float4 g = ld4(da + i * 4), v = ld4(z + i * 4);
        const float4 sc = ld4(scale + c0), sh = ld4(shift + c0), A = ld4(cA + c0), Bc = ld4(cB + c0), Cc = ld4(cC + c0);
        // A * (g * mask) + B + C * v, evaluated left to right as written (packed f32: the same IEEE operations)
        const float4 y = muladd4_pk(v, sc, sh);
        const float4 mk = make_float4(act_grad_mask(y.x, act), act_grad_mask(y.y, act), act_grad_mask(y.z, act), act_grad_mask(y.w, act));
        const float4 o = add4_pk(add4_pk(mul4_pk(A, mul4_pk(g, mk)), Bc), mul4_pk(Cc, v));
        st4(dz + i * 4, o);
    }
}

int launch_bn_bwd_apply(const float* da, const float* z, int64_t M, int C, const float* scale, const float* shift, int act,
                        const float* coefA, const float* coefB, const float* coefC, float* dz, hipStream_t st) {
    AMS_REQUIRE(C % 4 == 0, "bn_bwd_apply: C=%d must be a multiple of 4", C);
    const int64_t n4 = M * C / 4;
    note_kernel("bn_bwd_apply_kernel");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, st, da, z, n4, C / 4, scale, shift, act, coefA,
                       coefB, coefC, dz);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// K14-K16: Adam (TF1 form: w -= lr_t * m / (sqrt(v) + eps), lr_t carries the bias correction) + mask revert.
// The moments advance for every entry; entries with mask == 0 keep their old value (tf.where(mask, new, backup)).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const uint8_t* __restrict__ mask, int64_t n,
                                                   float lr_t, float b1, float b2, float eps) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // the form of TensorFlow's ApplyAdam functor (core/kernels/training_ops.cc), in f32 like its T = float instantiation:
        //   m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2);  var -= (m * alpha) / (sqrt(v) + epsilon)
        const float gi = g[i];
        const float m0 = m[i], v0 = v[i];
        const float mi = m0 + (gi - m0) * (1.f - b1);
        const float vi = v0 + (gi * gi - v0) * (1.f - b2);
        m[i] = mi;
        v[i] = vi;
        if (!mask || mask[i]) p[i] = p[i] - (mi * lr_t) / (sqrtf(vi) + eps);
    }
}

int launch_adam(float* p, const float* g, float* m, float* v, const uint8_t* mask, int64_t n, float lr_t, float b1,
                float b2, float eps, hipStream_t st) {
    note_kernel("adam_kernel");
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(256), 0, st, p, g, m, v, mask, n, lr_t, b1, b2, eps);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// one block per job: does any |w| exceed `limit` (or fail to be finite)?
__global__ __launch_bounds__(256) void weights_beyond_kernel(const WeightRange* __restrict__ jobs, float limit, int* __restrict__ flags) {
    const WeightRange j = jobs[blockIdx.x];
    int bad = 0;
    for (int64_t i = threadIdx.x; i < j.n; i += 256) bad |= !(fabsf(j.w[i]) <= limit);
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) flags[blockIdx.x] = bad ? 1 : 0;
}

int weights_beyond(const std::vector<WeightRange>& jobs, float limit, int* scratch_dev, int* over_host, hipStream_t st) {
    const size_t n = jobs.size();
    AMS_REQUIRE(n * (sizeof(WeightRange) + sizeof(int)) <= 4096 * sizeof(float), "weights_beyond: %zu jobs do not fit the scratch", n);
    WeightRange* jobs_dev = reinterpret_cast<WeightRange*>(scratch_dev);
    int* flags_dev = reinterpret_cast<int*>(jobs_dev + n);
    AMS_CHECK_HIP(hipMemcpyAsync(jobs_dev, jobs.data(), n * sizeof(WeightRange), hipMemcpyHostToDevice, st));
    AMS_CHECK_HIP(hipStreamSynchronize(st));             // (the table is a pageable host vector: it must not be read after this function returns)
    hipLaunchKernelGGL(weights_beyond_kernel, dim3((unsigned)n), dim3(256), 0, st, jobs_dev, limit, flags_dev);
    AMS_CHECK_LAUNCH();
    AMS_CHECK_HIP(hipMemcpyAsync(over_host, flags_dev, n * sizeof(int), hipMemcpyDeviceToHost, st));
    AMS_CHECK_HIP(hipStreamSynchronize(st));
    return AMS_OK;
}

// regularize=True of create_student_v3 (utils/graph_utils.py:451-456): loss += 0.01 * reduce_mean([l2_loss(v) for v in tvars]), l2_loss(v) =
// sum(v^2) / 2, tvars = every trainable variable or (train_biases_only) those without 'weight' in their name.  Over the flat arena with a byte
// mask of the regularised entries: g += (coef / n_vars) * p, and the blocks' sums of p^2 (f64, fixed order) for the loss term.
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ p, float* __restrict__ g, const uint8_t* __restrict__ mask, int64_t n,
                                                     float gscale, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const float v = p[i];
        g[i] = g[i] + gscale * v;
        acc += (double)v * (double)v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// loss[0] (the CE sum over the valid pixels) += lscale * sum(p^2) * loss[1]: the caller's mean loss[0] / loss[1] then carries the regulariser
__global__ __launch_bounds__(64) void l2_reg_finish_kernel(const double* __restrict__ part, int nparts, double lscale, double* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) acc += part[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) loss[0] += lscale * acc * loss[1];
}

int launch_l2_regularizer(const float* p, float* g, const uint8_t* mask, int64_t n, int n_vars, float coef, double* part_scratch, double* loss, hipStream_t st) {
    AMS_REQUIRE(p && g && mask && part_scratch && loss && n_vars > 0, "l2_regularizer: bad arguments");
    const int grid = 256;
    note_kernel("l2_reg_kernel");
    hipLaunchKernelGGL(l2_reg_kernel, dim3(grid), dim3(256), 0, st, p, g, mask, n, coef / (float)n_vars, part_scratch);
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL(l2_reg_finish_kernel, dim3(1), dim3(64), 0, st, part_scratch, grid, 0.5 * (double)coef / (double)n_vars, loss);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// out[i] = sum_k part[k*n + i].  One block per 16 outputs: thread t owns output (t & 15) and every LP-th split starting at (t >> 4),
// eight loads in flight; the LP f64 partials of an output are added in a fixed order (deterministic, and LP x shorter dependent load
// chains than one thread per output: with ~1000 splits the serial form cost > 100 us per call, 16 lanes with two loads in flight
// still 21 us for the 2048 rows of the first-block kernel).  LP = 64 from 256 splits on.
template <int LP>
__global__ __launch_bounds__(16 * LP) void reduce_splits_kernel(const float* __restrict__ part, int splits, int64_t n,
                                                                float* __restrict__ out, int64_t stride) {
    __shared__ double sacc[LP][17];
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int kpart = threadIdx.x >> 4;
    double s = 0.0;
    if (i < n) {
        const float* p = part + i;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = kpart;
        for (; k + 7 * LP < splits; k += 8 * LP) {
            const float v0 = p[(int64_t)k * stride], v1 = p[(int64_t)(k + LP) * stride], v2 = p[(int64_t)(k + 2 * LP) * stride],
                        v3 = p[(int64_t)(k + 3 * LP) * stride], v4 = p[(int64_t)(k + 4 * LP) * stride], v5 = p[(int64_t)(k + 5 * LP) * stride],
                        v6 = p[(int64_t)(k + 6 * LP) * stride], v7 = p[(int64_t)(k + 7 * LP) * stride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; k < splits; k += LP) s0 += p[(int64_t)k * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    sacc[kpart][threadIdx.x & 15] = s;
    __syncthreads();
    if (threadIdx.x < 16 && (int64_t)blockIdx.x * 16 + threadIdx.x < n) {
        double t = 0.0;
        for (int k = 0; k < LP; ++k) t += sacc[k][threadIdx.x];
        out[(int64_t)blockIdx.x * 16 + threadIdx.x] = (float)t;
    }
}

// Several such reductions in ONE launch (the fine-tune step defers the reductions that only feed the optimizer to the end of its backward
// pass: no launch, event or wait per layer).  Job j owns blocks [first_j, first_{j+1}); 64 lanes per output.
template <int LP>
__global__ __launch_bounds__(16 * LP) void reduce_batch_kernel(ReduceJobs jobs) {
    __shared__ double sacc[LP][17];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;          // block-uniform
    const float* part = jobs.part[j];
    const int splits = jobs.splits[j];
    const int64_t n = jobs.count[j], stride = jobs.stride[j];
    const int64_t i = (int64_t)((int)blockIdx.x - jobs.first_block[j]) * 16 + (threadIdx.x & 15);
    const int kpart = threadIdx.x >> 4;
    double s = 0.0;
    if (i < n) {
        const float* p = part + i;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int k = kpart;
        for (; k + 7 * LP < splits; k += 8 * LP) {
            const float v0 = p[(int64_t)k * stride], v1 = p[(int64_t)(k + LP) * stride], v2 = p[(int64_t)(k + 2 * LP) * stride],
                        v3 = p[(int64_t)(k + 3 * LP) * stride], v4 = p[(int64_t)(k + 4 * LP) * stride], v5 = p[(int64_t)(k + 5 * LP) * stride],
                        v6 = p[(int64_t)(k + 6 * LP) * stride], v7 = p[(int64_t)(k + 7 * LP) * stride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
        }
        for (; k < splits; k += LP) s0 += p[(int64_t)k * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    sacc[kpart][threadIdx.x & 15] = s;
    __syncthreads();
    if (threadIdx.x < 16 && i < n) {
        double t = 0.0;
        for (int k = 0; k < LP; ++k) t += sacc[k][threadIdx.x];
        jobs.out[j][i] = (float)t;
    }
}

int launch_reduce_batch(ReduceJobs& jobs, hipStream_t st) {
    if (jobs.n <= 0) return AMS_OK;
    int total = 0, most = 0;
    for (int j = 0; j < jobs.n; ++j) {
        jobs.first_block[j] = total;
        total += (int)cdiv(jobs.count[j], 16);
        if (jobs.splits[j] > most) most = jobs.splits[j];
    }
    note_kernel("reduce_batch_kernel");
    // 64 lanes per output only for long sums: with a few hundred rows the launch is bound by its thread count, not by its load chains
    if (most >= 1024) hipLaunchKernelGGL((reduce_batch_kernel<64>), dim3(total), dim3(1024), 0, st, jobs);
    else hipLaunchKernelGGL((reduce_batch_kernel<16>), dim3(total), dim3(256), 0, st, jobs);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_reduce_splits(const float* part, int splits, int64_t n, float* out, hipStream_t st, int64_t stride) {
    if (splits >= 256) hipLaunchKernelGGL((reduce_splits_kernel<64>), dim3(cdiv(n, 16)), dim3(1024), 0, st, part, splits, n, out, stride > 0 ? stride : n);
    else hipLaunchKernelGGL((reduce_splits_kernel<16>), dim3(cdiv(n, 16)), dim3(256), 0, st, part, splits, n, out, stride > 0 ? stride : n);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// x <- float(bf16(x)), round to nearest even: what bf16 STORAGE of a tensor would leave behind (the study in tools/bf16_storage_study.py)
__global__ __launch_bounds__(256) void round_bf16_kernel(float* __restrict__ p, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = ld4(p + 4 * i);
        v.x = (float)(__bf16)v.x; v.y = (float)(__bf16)v.y; v.z = (float)(__bf16)v.z; v.w = (float)(__bf16)v.w;
        st4(p + 4 * i, v);
    }
}
int launch_round_bf16(float* p, int64_t n, hipStream_t st) {
    AMS_REQUIRE(n % 4 == 0, "round_bf16: n must be a multiple of 4");
    hipLaunchKernelGGL(round_bf16_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, st, p, n / 4);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill(float* p, int64_t n, float v, hipStream_t st) {
    if (n <= 0) return AMS_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n)), dim3(256), 0, st, p, n, v);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}
int launch_copy(float* dst, const float* src, int64_t n, hipStream_t st) {
    if (n <= 0) return AMS_OK;
    AMS_CHECK_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return AMS_OK;
}

// masked parameters -> fp16, compacted in order (downlink payload of run.py:316-336).  Single block per 4096-entry
// segment computes its count, an exclusive scan over segments gives the offsets (two tiny kernels).
__global__ void pack_count_kernel(const uint8_t* mask, int64_t n, int64_t seg, int64_t* counts) {
    const int64_t s = blockIdx.x;
    const int64_t b = s * seg, e = b + seg < n ? b + seg : n;
    int local = 0;
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) local += (!mask || mask[i]) ? 1 : 0;
    __shared__ int sh[256];
    sh[threadIdx.x] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[s] = sh[0];
}
__global__ void pack_scan_kernel(int64_t* counts, int nseg, int64_t* total) {
    if (threadIdx.x || blockIdx.x) return;
    int64_t run = 0;
    for (int i = 0; i < nseg; ++i) { const int64_t c = counts[i]; counts[i] = run; run += c; }
    *total = run;
}
__global__ void pack_write_kernel(const float* p, const uint8_t* mask, int64_t n, int64_t seg, const int64_t* offs, __half* out) {
    // one thread per segment keeps the output order identical to a sequential host loop
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t b = s * seg;
    if (b >= n) return;
    const int64_t e = b + seg < n ? b + seg : n;
    int64_t o = offs[s];
    for (int64_t i = b; i < e; ++i)
        if (!mask || mask[i]) out[o++] = __float2half_rn(p[i]);
}

size_t pack_fp16_scratch(int64_t n) { return (size_t)cdiv64(n, 256); }

// counts: caller-owned device scratch of pack_fp16_scratch(n) int64 (no allocation here: the call is capture-safe and two
// students on different devices or streams never share state)
int launch_pack_fp16(const float* p, const uint8_t* mask, int64_t n, uint16_t* out, int64_t* n_out, int64_t* counts, hipStream_t st) {
    const int64_t seg = 256;
    const int nseg = (int)cdiv64(n, seg);
    hipLaunchKernelGGL(pack_count_kernel, dim3(nseg), dim3(256), 0, st, mask, n, seg, counts);
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1), 0, st, counts, nseg, n_out);
    AMS_CHECK_LAUNCH();
    hipLaunchKernelGGL(pack_write_kernel, dim3(cdiv(nseg, 64)), dim3(64), 0, st, p, mask, n, seg, counts, (__half*)out);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

}  // namespace ams
