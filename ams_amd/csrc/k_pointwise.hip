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
#include <string>

#include "kernels.hpp"

namespace ams {

// ---------------------------------------------------------------------------------------------------------
// Operand roles are SWAPPED in the MFMA (a = weights, b = activations): the 16x16 result then has the output
// channel along the accumulator registers (row = 4*(lane>>4) + i) and the pixel along lanes (col = lane & 15),
// so every lane owns 4 consecutive output channels of one pixel and the epilogue is float4 loads/stores
// (scale, shift, per-image bias, residual, result) instead of four scalar stores per tile.
// ---------------------------------------------------------------------------------------------------------
template <int RM, int NT>
__device__ __forceinline__ void pw_epilogue(const PwArgs& a, f32x4 (&acc)[RM][NT], int64_t m_base, int n0, int l15, int q) {
    const bool y_vec = (a.ldy & 3) == 0, r_vec = (a.ldr & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n4 = n0 + 16 * t + 4 * q;
        if (n4 >= a.N) continue;
        const bool full = n4 + 3 < a.N;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) {
            if (a.scale) { const float4 v = ld4(a.scale + n4); sc[0] = v.x; sc[1] = v.y; sc[2] = v.z; sc[3] = v.w; }
            if (a.shift) { const float4 v = ld4(a.shift + n4); sh[0] = v.x; sh[1] = v.y; sh[2] = v.z; sh[3] = v.w; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (n4 + i < a.N) { if (a.scale) sc[i] = a.scale[n4 + i]; if (a.shift) sh[i] = a.shift[n4 + i]; }
        }
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int64_t m = m_base + r * 16 + l15;
            if (m >= a.M) continue;
            float v[4] = {acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]};
            if (a.img_bias) {
                const float* bp = a.img_bias + (m / a.rows_per_img) * a.N + n4;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n4 + i < a.N) v[i] += bp[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i] * sc[i] + sh[i], a.act);
            if (a.res) {
                const float* rp = a.res + m * a.ldr + n4;
                if (full && r_vec) { const float4 rv = ld4(rp); v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (n4 + i < a.N) v[i] += rp[i];
                }
            }
            float* yp = a.y + m * a.ldy + n4;
            if (full && y_vec) st4(yp, make_float4(v[0], v[1], v[2], v[3]));
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n4 + i < a.N) yp[i] = v[i];
            }
        }
    }
}

// one 16-k chunk: 4 MFMA k-steps x NT column tiles x RM row groups; sB points at this lane's (k = 4q, n = l15) element
template <int RM, int NT, int PITCH>
__device__ __forceinline__ void pw_chunk(f32x4 (&acc)[RM][NT], const float4 (&av)[RM], const float* sB) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float xv[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) xv[r] = j == 0 ? av[r].x : j == 1 ? av[r].y : j == 2 ? av[r].z : av[r].w;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float wv = sB[j * PITCH + 16 * t];
#pragma unroll
            for (int r = 0; r < RM; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv[r], acc[r][t], 0, 0, 0);
        }
    }
}

// stage w[k0 .. k0+rows) x [n0 .. n0+16*NT) into LDS (row pitch PITCH), zero-filled outside Kw x N.
// Loads are issued in batches of U independent requests before any LDS store, so a panel costs a few L2 round trips
// instead of one per element (the panel is re-staged by every block: it must not serialise).
template <int NT, int PITCH>
__device__ __forceinline__ void pw_stage_w(const PwArgs& a, float* dst, int k0, int rows, int n0, int tid, int nthreads) {
    constexpr int cols = 16 * NT;
    constexpr int U = 8;
    if (a.w_sn == 1 && (a.N & 3) == 0 && (a.w_sk & 3) == 0) {
        constexpr int c4 = cols / 4;
        const int pieces = rows * c4;
        for (int base = tid; base < pieces; base += nthreads * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * nthreads;
                const int kk = e / c4, nn = (e - kk * c4) * 4;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < pieces && k0 + kk < a.Kw && n0 + nn < a.N) v[u] = ld4(a.w + (int64_t)(k0 + kk) * a.w_sk + (n0 + nn));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * nthreads;
                const int kk = e / c4, nn = (e - kk * c4) * 4;
                if (e < pieces) st4(dst + kk * PITCH + nn, v[u]);
            }
        }
        return;
    }
    const int total = rows * cols;
    const bool n_contig = a.w_sn == 1;
    for (int base = tid; base < total; base += nthreads * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * nthreads;
            int kk, nn;
            if (n_contig) { kk = e / cols; nn = e - kk * cols; } else { nn = e / rows; kk = e - nn * rows; }
            v[u] = 0.f;
            if (e < total && k0 + kk < a.Kw && n0 + nn < a.N) v[u] = a.w[(int64_t)(k0 + kk) * a.w_sk + (int64_t)(n0 + nn) * a.w_sn];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * nthreads;
            int kk, nn;
            if (n_contig) { kk = e / cols; nn = e - kk * cols; } else { nn = e / rows; kk = e - nn * rows; }
            if (e < total) dst[kk * PITCH + nn] = v[u];
        }
    }
}

// ---- variant S: streaming layers (huge M, small K x N).  The whole weight panel of this column tile stays in LDS for
// the block's lifetime; every wave walks its own 16*RM-row groups (grid-stride), no barrier after the prologue, and
// the A fragment of the NEXT (row group, k chunk) is in flight while the current one feeds the matrix pipe.
template <int RM, int NT>
__global__ __launch_bounds__(256) void pw_gemm_f32_s(PwArgs a, int n_tiles_n, int64_t n_groups) {
    constexpr int PITCH = 16 * NT + 4;
    extern __shared__ __attribute__((aligned(16))) float sW[];          // [Kpad][PITCH]
    const int tile_n = blockIdx.y;
    const int n0 = tile_n * 16 * NT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int K = a.K, n_chunks = (K + 15) / 16;
    pw_stage_w<NT, PITCH>(a, sW, 0, n_chunks * 16, n0, tid, 256);
    __syncthreads();

    const int64_t wave_stride = (int64_t)gridDim.x * 4;
    int64_t g = (int64_t)blockIdx.x * 4 + wave;
    if (g >= n_groups) return;
    auto row_ptr = [&](int64_t grp, int r) {
        int64_t m = grp * (16 * RM) + r * 16 + l15;
        if (m > a.M - 1) m = a.M - 1;
        return a.x + m * (int64_t)a.ldx + 4 * q;
    };
    float4 a_cur[RM], a_nxt[RM];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < RM; ++r) a_cur[r] = (4 * q < K) ? ld4(row_ptr(g, r)) : zero4;
    f32x4 acc[RM][NT];
    while (g < n_groups) {
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int64_t g_next = g + wave_stride;
        for (int c = 0; c < n_chunks; ++c) {
            // prefetch: next chunk of this group, or chunk 0 of the next group
            if (c + 1 < n_chunks) {
                const bool ok = (c + 1) * 16 + 4 * q < K;
#pragma unroll
                for (int r = 0; r < RM; ++r) a_nxt[r] = ok ? ld4(row_ptr(g, r) + (c + 1) * 16) : zero4;
            } else if (g_next < n_groups) {
#pragma unroll
                for (int r = 0; r < RM; ++r) a_nxt[r] = (4 * q < K) ? ld4(row_ptr(g_next, r)) : zero4;
            }
            pw_chunk<RM, NT, PITCH>(acc, a_cur, sW + (c * 16 + 4 * q) * PITCH + l15);
#pragma unroll
            for (int r = 0; r < RM; ++r) a_cur[r] = a_nxt[r];
        }
        pw_epilogue<RM, NT>(a, acc, g * (16 * RM), n0, l15, q);
        g = g_next;
    }
}

// ---- variant L: late layers (M = B*33*65 rows, K and/or N in the hundreds).  One (64*RM) x (16*NT) tile per block,
// K walked in 32-wide stages whose weight panel is double-buffered in LDS: the next stage's panel travels
// global -> registers while the current one is consumed, and is written to the other buffer before the single barrier.
template <int RM, int NT>
__global__ __launch_bounds__(256) void pw_gemm_f32_l(PwArgs a, int n_tiles_n, unsigned nblocks) {
    constexpr int BK = 32;
    constexpr int PITCH = 16 * NT + 4;
    constexpr int NREG = (BK * 16 * NT + 255) / 256;              // staged elements per thread
    __shared__ float sW[2][BK * PITCH];
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
    pw_epilogue<RM, NT>(a, acc, m_base, n0, l15, q);
}

template <int RM, int NT>
static int launch_pw_l(const PwArgs& a, hipStream_t st) {
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    const int64_t nblocks = cdiv64(a.M, 64 * RM) * n_tiles_n;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL) { set_error("pointwise: bad grid %lld", (long long)nblocks); return AMS_E_INVALID; }
    static const std::string nm = "pw_gemm_f32_l<" + std::to_string(RM) + ", " + std::to_string(NT) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_gemm_f32_l<RM, NT>), dim3((unsigned)nblocks), dim3(256), 0, st, a, n_tiles_n, (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int RM, int NT>
static int launch_pw_s(const PwArgs& a, hipStream_t st) {
    constexpr int PITCH = 16 * NT + 4;
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    const int64_t n_groups = cdiv64(a.M, 16 * RM);
    const size_t lds = (size_t)((a.K + 15) / 16 * 16) * PITCH * sizeof(float);
    int64_t blocks = cdiv64(n_groups, 4);
    // persistent: a few blocks per CU, as many as the LDS panel allows
    const int per_cu = lds > 48 * 1024 ? 2 : lds > 24 * 1024 ? 4 : 6;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    static const std::string nm = "pw_gemm_f32_s<" + std::to_string(RM) + ", " + std::to_string(NT) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_gemm_f32_s<RM, NT>), dim3((unsigned)blocks, n_tiles_n), dim3(256), lds, st, a, n_tiles_n, n_groups);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_pointwise(const PwArgs& a, hipStream_t st) {
    AMS_REQUIRE(a.M > 0 && a.K > 0 && a.N > 0, "pointwise: empty problem M=%lld K=%d N=%d", (long long)a.M, a.K, a.N);
    AMS_REQUIRE(a.Kw > 0 && a.Kw <= a.K, "pointwise: Kw=%d must be in 1..K=%d", a.Kw, a.K);
    AMS_REQUIRE(a.K % 4 == 0 && a.ldx % 4 == 0, "pointwise: K (%d) and ldx (%d) must be multiples of 4", a.K, a.ldx);
    AMS_REQUIRE((reinterpret_cast<uintptr_t>(a.x) & 15) == 0, "pointwise: x must be 16-byte aligned");
    const int n16 = cdiv(a.N, 16);
    const int kpad = (a.K + 15) / 16 * 16;
    // ---- streaming variant: weight panel resident in LDS (<= 56 KB), plenty of rows --------------------------
    if (a.M >= 32768) {
        int nt = n16 <= 12 ? n16 : (n16 % 12 == 0 ? 12 : n16 % 10 == 0 ? 10 : n16 % 8 == 0 ? 8 : 6);
        if (nt == 7) nt = 8; if (nt == 11) nt = 12;
        if ((size_t)kpad * (16 * nt + 4) * 4 <= 56 * 1024) {
            switch (nt) {
                case 1: return launch_pw_s<2, 1>(a, st);
                case 2: return launch_pw_s<2, 2>(a, st);
                case 3: return launch_pw_s<2, 3>(a, st);
                case 4: return launch_pw_s<2, 4>(a, st);
                case 5: return launch_pw_s<2, 5>(a, st);
                case 6: return launch_pw_s<2, 6>(a, st);
                case 8: return launch_pw_s<2, 8>(a, st);
                case 9: return launch_pw_s<2, 9>(a, st);
                case 10: return launch_pw_s<1, 10>(a, st);
                case 12: return launch_pw_s<1, 12>(a, st);
                default: break;
            }
        }
    }
    // ---- tiled variant: pick the column-tile width (in 16s) that wastes least while giving the chip >= ~3 blocks per CU
    int best_nt = 1, best_rm = 1;
    double best = -1;
    for (int nt = 6; nt >= 1; --nt)
        for (int rm = 2; rm >= 1; --rm) {
            const int tn = cdiv(n16, nt);
            const double blocks = (double)cdiv64(a.M, 64 * rm) * tn;
            const double useful = (double)n16 / (tn * nt);                      // fraction of computed columns that exist
            const double fill = blocks >= 768 ? 1.0 : blocks / 768.0;           // parallelism
            const double reuse = 0.85 + 0.15 * (nt * rm) / 12.0;                // bigger tiles re-read less
            const double score = useful * fill * reuse;
            if (score > best) { best = score; best_nt = nt; best_rm = rm; }
        }
#define PW_L(RM_, NT_) if (best_rm == RM_ && best_nt == NT_) return launch_pw_l<RM_, NT_>(a, st);
    PW_L(2, 6) PW_L(2, 5) PW_L(2, 4) PW_L(2, 3) PW_L(2, 2) PW_L(2, 1)
    PW_L(1, 6) PW_L(1, 5) PW_L(1, 4) PW_L(1, 3) PW_L(1, 2) PW_L(1, 1)
#undef PW_L
    set_error("pointwise: no tile configuration");
    return AMS_E_INVALID;
}

// =========================================================================================================
// Split-bf16 ("bf16x3") late-layer GEMM.  f32 activations stay f32 in HBM; inside the kernel every operand is split
// into bf16 hi + bf16 lo (16 significand bits together) and the product is formed as hi*hi + lo*hi + hi*lo on the
// bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, f32 accumulate): 3 instructions per 32 k instead of 8 f32-MFMA
// instructions of twice the latency (the f32-input MFMA runs at 1/16 of the bf16 rate on gfx950).  Dropped terms are
// <= 2^-16 relative per product, i.e. ~1e-5 on a layer output — two orders inside the 1e-3 logit tolerance.  The
// weights are split once per ams_student_freeze into [N][Kp] hi / lo panels (k contiguous, Kp = K rounded up to 32).
// =========================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void split_w_kernel(const float* __restrict__ w, int64_t sk, int64_t sn, int K, int N, int Kp,
                               unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Kp) return;
    const int n = (int)(i / Kp), k = (int)(i % Kp);
    const float v = k < K ? w[k * sk + n * sn] : 0.f;
    const unsigned short h = bf16_rne_bits(v);
    const float hf = __uint_as_float((unsigned)h << 16);
    hi[i] = h;
    lo[i] = bf16_rne_bits(v - hf);
}

int launch_split_weights(const float* w, int64_t sk, int64_t sn, int K, int N, int Kp, uint16_t* hi, uint16_t* lo, hipStream_t st) {
    const int64_t n = (int64_t)N * Kp;
    hipLaunchKernelGGL(split_w_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, w, sk, sn, K, N, Kp, hi, lo);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// 8 consecutive f32 -> bf16x8 hi and lo
__device__ __forceinline__ void split8(const float4& u, const float4& v, bf16x8& hi, bf16x8& lo) {
    const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)f[j];
        hi[j] = h;
        lo[j] = (__bf16)(f[j] - (float)h);
    }
}

template <int RM, int NT>
__global__ __launch_bounds__(256) void pw_gemm_bf16x3_l(PwArgs a, const unsigned short* __restrict__ whi,
                                                        const unsigned short* __restrict__ wlo, int Kp, int n_tiles_n,
                                                        unsigned nblocks) {
    constexpr int PITCH = 40;                        // bf16 elements per LDS row: 80 B, conflict-free for ds_read_b128
    constexpr int ROWS = 16 * NT;
    constexpr int NPIECE = 2 * ROWS * 4;             // 16-byte pieces per stage (hi + lo panels, 32 k = 4 pieces per row)
    constexpr int NREG = (NPIECE + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short sW[2][2][ROWS * PITCH];     // [buffer][hi/lo]
    const unsigned lb = xcd_remap(blockIdx.x, nblocks);
    const int tile_n = lb % n_tiles_n;
    const int64_t tile_m = lb / n_tiles_n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int n0 = tile_n * ROWS;
    const int64_t m_base = tile_m * (64 * RM) + wave * (16 * RM);
    const int K = a.K, n_stages = Kp / 32;

    uint4 wreg[NREG];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int e = tid + u * 256;
            const int which = e / (ROWS * 4), r = e - which * (ROWS * 4), n = r >> 2, part = r & 3;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (e < NPIECE && n0 + n < a.N)
                v = *reinterpret_cast<const uint4*>((which ? wlo : whi) + (int64_t)(n0 + n) * Kp + s * 32 + part * 8);
            wreg[u] = v;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int e = tid + u * 256;
            const int which = e / (ROWS * 4), r = e - which * (ROWS * 4), n = r >> 2, part = r & 3;
            if (e < NPIECE) *reinterpret_cast<uint4*>(&sW[buf][which][n * PITCH + part * 8]) = wreg[u];
        }
    };

    const float* arow[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        int64_t m = m_base + r * 16 + l15;
        if (m > a.M - 1) m = a.M - 1;
        arow[r] = a.x + m * (int64_t)a.ldx + 8 * q;
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_cur[RM][2], a_nxt[RM][2];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const bool ok = 8 * q < K;
        a_cur[r][0] = ok ? ld4(arow[r]) : zero4;
        a_cur[r][1] = ok ? ld4(arow[r] + 4) : zero4;
    }
    f32x4 acc[RM][NT];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < n_stages; ++s) {
        if (s + 1 < n_stages) {
            load_stage(s + 1);
            const bool ok = (s + 1) * 32 + 8 * q < K;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                a_nxt[r][0] = ok ? ld4(arow[r] + (s + 1) * 32) : zero4;
                a_nxt[r][1] = ok ? ld4(arow[r] + (s + 1) * 32 + 4) : zero4;
            }
        }
        bf16x8 xh[RM], xl[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) split8(a_cur[r][0], a_cur[r][1], xh[r], xl[r]);
        const unsigned short* bh = &sW[s & 1][0][l15 * PITCH + 8 * q];
        const unsigned short* bl = &sW[s & 1][1][l15 * PITCH + 8 * q];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(bh + t * 16 * PITCH);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(bl + t * 16 * PITCH);
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[r], acc[r][t], 0, 0, 0);
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[r], acc[r][t], 0, 0, 0);
                acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[r], acc[r][t], 0, 0, 0);
            }
        }
        if (s + 1 < n_stages) {
            store_stage((s + 1) & 1);
#pragma unroll
            for (int r = 0; r < RM; ++r) { a_cur[r][0] = a_nxt[r][0]; a_cur[r][1] = a_nxt[r][1]; }
        }
        __syncthreads();
    }
    pw_epilogue<RM, NT>(a, acc, m_base, n0, l15, q);
}

template <int RM, int NT>
static int launch_pw_x3(const PwArgs& a, const uint16_t* whi, const uint16_t* wlo, int Kp, hipStream_t st) {
    const int n_tiles_n = cdiv(a.N, 16 * NT);
    const int64_t nblocks = cdiv64(a.M, 64 * RM) * n_tiles_n;
    static const std::string nm = "pw_gemm_bf16x3_l<" + std::to_string(RM) + ", " + std::to_string(NT) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_gemm_bf16x3_l<RM, NT>), dim3((unsigned)nblocks), dim3(256), 0, st, a, whi, wlo, Kp, n_tiles_n,
                       (unsigned)nblocks);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

// y = epilogue(x @ w) with w given as pre-split bf16 hi/lo panels [N][Kp]; requires K % 8 == 0
int launch_pointwise_split(const PwArgs& a, const uint16_t* whi, const uint16_t* wlo, int Kp, hipStream_t st) {
    AMS_REQUIRE(a.M > 0 && a.K > 0 && a.N > 0 && Kp % 32 == 0 && Kp >= a.K, "pointwise_split: bad problem");
    AMS_REQUIRE(a.K % 8 == 0 && a.ldx % 4 == 0, "pointwise_split: K (%d) must be a multiple of 8", a.K);
    const int n16 = cdiv(a.N, 16);
    int best_nt = 1, best_rm = 1;
    double best = -1;
    for (int nt = 6; nt >= 1; --nt)
        for (int rm = 2; rm >= 1; --rm) {
            const int tn = cdiv(n16, nt);
            const double blocks = (double)cdiv64(a.M, 64 * rm) * tn;
            const double useful = (double)n16 / (tn * nt);
            const double fill = blocks >= 768 ? 1.0 : blocks / 768.0;
            const double reuse = 0.85 + 0.15 * (nt * rm) / 12.0;
            const double score = useful * fill * reuse;
            if (score > best) { best = score; best_nt = nt; best_rm = rm; }
        }
#define PW_X(RM_, NT_) if (best_rm == RM_ && best_nt == NT_) return launch_pw_x3<RM_, NT_>(a, whi, wlo, Kp, st);
    PW_X(2, 6) PW_X(2, 5) PW_X(2, 4) PW_X(2, 3) PW_X(2, 2) PW_X(2, 1)
    PW_X(1, 6) PW_X(1, 5) PW_X(1, 4) PW_X(1, 3) PW_X(1, 2) PW_X(1, 1)
#undef PW_X
    set_error("pointwise_split: no tile configuration");
    return AMS_E_INVALID;
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
        float4 v = ok ? ld4(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
template <>
struct VecLd<2> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[2]) {
        float2 v = ok ? *reinterpret_cast<const float2*>(p) : make_float2(0.f, 0.f);
        o[0] = v.x; o[1] = v.y;
    }
};
template <>
struct VecLd<1> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[1]) { o[0] = ok ? *p : 0.f; }
};

template <int VA, int VB>
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
    // each wave takes every 4th group of 4 pixels; 2 groups in flight for latency hiding
    // (loop bounds are wave-uniform: an MFMA must be issued by the whole wave)
    for (int64_t mg = m_begin + wave * 4; mg < m_end; mg += 32) {
        float xa[2][VA], yb[2][VB];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t mm = mg + q + h * 16;
            const bool ok = mm < m_end;
            VecLd<VA>::ld(a.x + mm * (int64_t)a.ldx + ka, ok && ka_ok, xa[h]);
            VecLd<VB>::ld(a.dy + mm * (int64_t)a.ldy + nb, ok && nb_ok, yb[h]);
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

size_t pointwise_wgrad_scratch(int64_t M, int K, int N) { return (size_t)wgrad_splits(M, K, N) * K * N; }

template <int VA, int VB>
static int launch_wg_t(const WgArgs& a, int splits, hipStream_t st) {
    const int tiles_k = cdiv(a.K, 16 * VA), tiles_n = cdiv(a.N, 16 * VB);
    int64_t rows = cdiv64(a.M, splits);
    rows = (rows + 3) / 4 * 4;
    static const std::string nm = "pw_wgrad_f32<" + std::to_string(VA) + ", " + std::to_string(VB) + ">";
    note_kernel(nm.c_str());
    hipLaunchKernelGGL((pw_wgrad_f32<VA, VB>), dim3(splits, tiles_k * tiles_n), dim3(256), 0, st, a, tiles_k, tiles_n, rows);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

int launch_pointwise_wgrad(const WgArgs& a, hipStream_t st) {
    AMS_REQUIRE(a.M > 0 && a.K > 0 && a.N > 0, "wgrad: empty problem");
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
