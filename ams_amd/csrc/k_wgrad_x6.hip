// Weight gradient of a 1x1 convolution on the bf16 matrix pipe: dw[K,N] = sum_m x[m,K]^T dy[m,N], late layers.
//
// The f32-input MFMA kernel (k_pointwise.hip) is matrix-pipe bound once K*N is large (960x160 at M = 16384: 115-175 us,
// ~0.6 TB/s of operands).  Here both operands are split into three bf16 parts (hi, mid, lo: all 24 significand bits) and
// the six products hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi run on v_mfma_f32_16x16x32_bf16 (f32 accumulate); the
// dropped terms are <= 2^-24 relative, i.e. f32 rounding level.
//
// The contraction index is the PIXEL, but both operands are pixel-major in memory ([m][channel]): an MFMA operand
// fragment (8 consecutive contraction values of one channel per lane) is a COLUMN of the tile.  The tile is therefore
// staged row-major in LDS as bf16 (coalesced float4 global loads, 8-byte LDS writes) and read back with the gfx950
// transpose read ds_read_b64_tr_b16 (lane p of a 16-lane group supplies the address of 4 contiguous bf16 of a 4x16 block
// and receives column p; mapping verified by tools/probes/tr_read.hip).  Contraction slot (q, e) of a lane group holds
// pixel 4q + e (e < 4) or 16 + 4q + (e - 4): any bijection works as long as both operands use it, and this one makes
// every transpose read touch 16 consecutive tile rows; with a row pitch of an odd number of 32-byte units the 8 rows of
// a 32-lane pass fall into distinct bank groups.
//
// Block = 4 waves, output tile (16*KT) x (16*NW); wave w owns the k-tiles w, w+4, ...; the pixel range is split over
// blockIdx.x and the split partials are added in a fixed order by launch_reduce_splits (deterministic).
#include "pw_common.hpp"

namespace ams {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_read8(const unsigned short* lo_rows, const unsigned short* hi_rows) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_rows);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_rows);
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}

// f32 x4 -> three bf16 x4 parts, written to the three planes of the LDS image
__device__ __forceinline__ void split_store(const float4& v, unsigned short* p0, int plane) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 a = (__bf16)f[j];
        const float r1 = f[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[j] = a; m[j] = b; l[j] = (__bf16)(r1 - (float)b);
    }
    *reinterpret_cast<bf16x4*>(p0) = h;
    *reinterpret_cast<bf16x4*>(p0 + plane) = m;
    *reinterpret_cast<bf16x4*>(p0 + 2 * plane) = l;
}

constexpr int wg_pitch(int channels) {       // bf16 elements per LDS row: multiple of 16 elements (32 B), odd count of them
    return ((channels + 15) / 16 * 16 / 16) % 2 == 1 ? (channels + 15) / 16 * 16 : (channels + 15) / 16 * 16 + 16;
}

// XD: the operand transforms of WgArgs applied between the global load and the split into the LDS image (a thread's float4 covers the
// same four channels in every step: the per-channel vectors are loaded once)
// NWV = 8 (wide tiles): eight waves share one staged tile, split 4 (k) x 2 (n) — a wave owns KT / 4 k-tiles and half of the n-tiles, so the
// dy fragments are re-read by four waves instead of all of them and two waves per SIMD overlap one's LDS reads and split with the other's
// MFMAs (with four waves of 120 MFMAs each, one per SIMD, nothing overlapped: 104-185 us on the 960-wide layers inside the step).  The
// order of the products of every output element is unchanged: same bits as NWV = 4.
template <int KT, int NW, int XD = 0, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void pw_wgrad_bf16x6(WgArgs a, int tiles_n, int64_t rows_per_split) {
    constexpr int NTH = 64 * NWV;                              // threads of a block
    constexpr int NSPL = NWV / 4;                              // n-split of the waves (1 | 2)
    constexpr int NWH = (NW + NSPL - 1) / NSPL;                // n-tiles per wave
    constexpr bool XT = (XD & 1) != 0, DT = (XD & 2) != 0;     // operand transform on x (WgArgs::x_mode 1) / on dy (dy_mode 2)
    constexpr int CX = 16 * KT, CY = 16 * NW;                 // channels per tile side
    constexpr int PX = wg_pitch(CX), PY = wg_pitch(CY);       // LDS row pitch in bf16 elements
    constexpr int PLX = 32 * PX, PLY = 32 * PY;               // one part plane (32 pixels)
    constexpr int VX = CX / 4, VY = CY / 4;                   // float4 per tile row
    constexpr int NLX = (32 * VX + NTH - 1) / NTH, NLY = (32 * VY + NTH - 1) / NTH;
    static_assert(KT % 4 == 0, "k-tiles are dealt to the four waves evenly");
    constexpr int MYK = KT / 4;                                // k-tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* sX = smem;                                 // [3][32][PX]
    unsigned short* sY = smem + 3 * PLX;                       // [3][32][PY]

    const int tile = blockIdx.y, tk = tile / tiles_n, tn = tile - tk * tiles_n;
    const int k0 = tk * CX, n0 = tn * CY;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the tile offsets below stay wave-uniform
    const int wave = wave_id & 3, wn = wave_id >> 2;                    // k-tiles wave, wave + 4, ...; n-tiles wn * NWH ...
    const int64_t m_begin = (int64_t)blockIdx.x * rows_per_split;
    int64_t m_end = m_begin + rows_per_split;
    if (m_end > a.M) m_end = a.M;
    const int n_steps = m_begin < m_end ? (int)((m_end - m_begin + 31) / 32) : 0;

    // global -> register staging of one 32-pixel step (branch-free: clamped addresses, zero select)
    float4 rx[NLX], ry[NLY];
    float4 rz[DT ? NLY : 1];                                      // dy_mode 2: the raw output beside the gradient
    unsigned rx_ok = 0, ry_ok = 0;                                // which of this thread's pieces lie inside the problem (raw values are kept until they are staged)
    float4 vxs[XT ? NLX : 1], vxh[XT ? NLX : 1], vA[DT ? NLY : 1], vB[DT ? NLY : 1], vC[DT ? NLY : 1];
    if constexpr (XT) {
#pragma unroll
        for (int u = 0; u < NLX; ++u) {
            const int e = tid + NTH * u, row = e / VX, c4 = (e - row * VX) * 4;
            int k = k0 + c4;
            if (k > a.K - 4) k = a.K - 4;
            vxs[u] = ld4(a.x_v0 + k);
            vxh[u] = ld4(a.x_v1 + k);
        }
    }
    if constexpr (DT) {
#pragma unroll
        for (int u = 0; u < NLY; ++u) {
            const int e = tid + NTH * u, row = e / VY, c4 = (e - row * VY) * 4;
            int n = n0 + c4;
            if (n > a.N - 4) n = a.N - 4;
            vA[u] = ld4(a.dy_v0 + n);
            vB[u] = ld4(a.dy_v1 + n);
            vC[u] = ld4(a.dy_v2 + n);
        }
    }
    auto fetch = [&](int s) {
        const int64_t mb = m_begin + (int64_t)s * 32;
#pragma unroll
        for (int u = 0; u < NLX; ++u) {
            const int e = tid + NTH * u, row = e / VX, c4 = (e - row * VX) * 4;
            int64_t m = mb + row;
            int k = k0 + c4;
            const bool ok = e < 32 * VX && m < m_end && k < a.K;
            if (m > a.M - 1) m = a.M - 1;
            if (k > a.K - 4) k = a.K - 4;
            const float4 v = ld4(a.x + m * (int64_t)a.ldx + k);
            if constexpr (XT) {
                // raw value now, transform when it is staged: arithmetic on the loaded value HERE would make the wave wait for the load
                // it has just issued, i.e. take the prefetch out from under the MFMAs of the current step
                rx[u] = v;
                rx_ok = ok ? (rx_ok | (1u << u)) : (rx_ok & ~(1u << u));
            } else {
                rx[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < NLY; ++u) {
            const int e = tid + NTH * u, row = e / VY, c4 = (e - row * VY) * 4;
            int64_t m = mb + row;
            int n = n0 + c4;
            const bool ok = e < 32 * VY && m < m_end && n < a.N;
            if (m > a.M - 1) m = a.M - 1;
            if (n > a.N - 4) n = a.N - 4;
            const float4 v = ld4(a.dy + m * (int64_t)a.ldy + n);
            if constexpr (DT) {
                ry[u] = v;
                ry_ok = ok ? (ry_ok | (1u << u)) : (ry_ok & ~(1u << u));
                rz[u] = ld4(a.dy2 + m * (int64_t)a.ldy + n);
            } else {
                ry[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < NLX; ++u) {
            const int e = tid + NTH * u, row = e / VX, c4 = (e - row * VX) * 4;
            if constexpr (XT) {
                const float4 y = muladd4_pk(rx[u], vxs[u], vxh[u]);
                const float4 v = make_float4(apply_act(y.x, a.x_act), apply_act(y.y, a.x_act), apply_act(y.z, a.x_act), apply_act(y.w, a.x_act));
                const bool inside = (rx_ok >> u) & 1u;                 // rows / columns outside the problem contribute zeros
                rx[u] = make_float4(inside ? v.x : 0.f, inside ? v.y : 0.f, inside ? v.z : 0.f, inside ? v.w : 0.f);
            }
            if (e < 32 * VX) split_store(rx[u], sX + row * PX + c4, PLX);
        }
#pragma unroll
        for (int u = 0; u < NLY; ++u) {
            const int e = tid + NTH * u, row = e / VY, c4 = (e - row * VY) * 4;
            if constexpr (DT) {
                // (A g + B) + C z as bn_bwd_apply_kernel evaluates it; rows / columns outside the problem stay zero (B is not)
                const float4 y = add4_pk(add4_pk(mul4_pk(vA[u], ry[u]), vB[u]), mul4_pk(vC[u], rz[u]));
                const bool inside = (ry_ok >> u) & 1u;
                ry[u] = inside ? y : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (e < 32 * VY) split_store(ry[u], sY + row * PY + c4, PLY);
        }
    };

    f32x4 acc[MYK][NWH];
#pragma unroll
    for (int i = 0; i < MYK; ++i)
#pragma unroll
        for (int t = 0; t < NWH; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane transpose-read offsets inside a plane: rows 4q + (l15 >> 2) (+16 for the upper half), 4 columns at 4*(l15 & 3)
    const int tr_row = 4 * q + (l15 >> 2), tr_col = 4 * (l15 & 3);
    if (n_steps > 0) fetch(0);
    for (int s = 0; s < n_steps; ++s) {
        stage();
        __syncthreads();
        if (s + 1 < n_steps) fetch(s + 1);
        bf16x8 xa[MYK][3];
#pragma unroll
        for (int i = 0; i < MYK; ++i) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned short* b = sX + p * PLX + tr_row * PX + 16 * (wave + 4 * i) + tr_col;
                xa[i][p] = tr_read8(b, b + 16 * PX);
            }
        }
#pragma unroll
        for (int t = 0; t < NWH; ++t) {
            if (NSPL > 1 && wn * NWH + t >= NW) break;                  // an odd tile count leaves the second half one short (wave-uniform)
            bf16x8 yb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const unsigned short* b = sY + p * PLY + tr_row * PY + 16 * (wn * NWH + t) + tr_col;
                yb[p] = tr_read8(b, b + 16 * PY);
            }
            // A = dy fragment (rows of D = n), B = x fragment (cols of D = k): a lane ends up with 4 consecutive n of one k
            // -> float4 stores.  Smallest terms first; the k-tiles alternate so consecutive MFMAs hit different accumulators.
#pragma unroll
            for (int i = 0; i < MYK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb[2], xa[i][0], acc[i][t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MYK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb[0], xa[i][2], acc[i][t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MYK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb[1], xa[i][1], acc[i][t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MYK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb[1], xa[i][0], acc[i][t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MYK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb[0], xa[i][1], acc[i][t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MYK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb[0], xa[i][0], acc[i][t], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: column (lane & 15) = B index = k, rows 4q..4q+3 = A index = n
    float* out = a.scratch + (int64_t)blockIdx.x * a.K * a.N;
#pragma unroll
    for (int i = 0; i < MYK; ++i) {
        const int kt = wave + 4 * i;
        const int k = k0 + 16 * kt + l15;
        if (k >= a.K) continue;
#pragma unroll
        for (int t = 0; t < NWH; ++t) {
            if (NSPL > 1 && wn * NWH + t >= NW) break;
            const int n = n0 + 16 * (wn * NWH + t) + 4 * q;
            if (n + 3 < a.N) st4(out + (int64_t)k * a.N + n, make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]));
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < a.N) out[(int64_t)k * a.N + n + r] = acc[i][t][r];
            }
        }
    }
}

// applies when the f32 kernel would be matrix-pipe bound: few pixels, many channel pairs
bool pointwise_wgrad_x6_applies(int64_t M, int K, int N, int ldx, int ldy) {
    return M <= 32768 && M >= 1024 && K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (int64_t)K * N >= 64 * 64;
}

static void wg6_config(int K, int N, int* kt, int* nw) {
    *kt = K > 64 ? 8 : 4;
    const int n16 = cdiv(N, 16);
    if (n16 <= 10) *nw = n16 <= 4 ? 4 : (n16 <= 5 ? 5 : (n16 <= 6 ? 6 : (n16 <= 8 ? 8 : 10)));
    else if (n16 % 10 == 0) *nw = 10;
    else if (n16 % 8 == 0) *nw = 8;
    else if (n16 % 6 == 0) *nw = 6;
    else *nw = 8;
}

int wgrad_x6_splits(int64_t M, int K, int N) {
    int kt, nw;
    wg6_config(K, N, &kt, &nw);
    const int tiles = cdiv(K, 16 * kt) * cdiv(N, 16 * nw);
    int64_t want = (512 + tiles - 1) / tiles;              // ~2 blocks per CU
    const int64_t max_by_rows = (M + 255) / 256;          // at least 8 steps per block
    if (want > max_by_rows) want = max_by_rows;
    // 32 pixel splits at most (the sweep at 8 x 512x1024: 64 -> 8.10 ms a step, 48 -> 8.07, 32 -> 8.04, 24 -> 8.03, 16 -> 8.3: the partial
    // products and their reduction are traffic the main stream's kernels compete with)
    const int cap = knobs().wg6_split_cap > 0 ? knobs().wg6_split_cap : 32;        // tuning knob AMS_WG6_SPLITS
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

template <int KT, int NW, int XD, int NWV>
static int launch_wg6_k(const WgArgs& a, int splits, int tiles_k, int tiles_n, int64_t rows, size_t lds, hipStream_t st) {
    RUN_RC(func_allow_lds((const void*)pw_wgrad_bf16x6<KT, NW, XD, NWV>, lds));
    hipLaunchKernelGGL((pw_wgrad_bf16x6<KT, NW, XD, NWV>), dim3(splits, tiles_k * tiles_n), dim3(64 * NWV), lds, st, a, tiles_n, rows);
    AMS_CHECK_LAUNCH();
    return AMS_OK;
}

template <int KT, int NW>
static int launch_wg6_t(const WgArgs& a, int splits, hipStream_t st) {
    const int tiles_k = cdiv(a.K, 16 * KT), tiles_n = cdiv(a.N, 16 * NW);
    int64_t rows = cdiv64(a.M, splits);
    rows = (rows + 31) / 32 * 32;
    const size_t lds = (size_t)3 * 32 * (wg_pitch(16 * KT) + wg_pitch(16 * NW)) * sizeof(unsigned short);
    // eight waves per tile (from 96 output columns on): built and bit-identical, measured NEUTRAL in the step (7.99-8.02 ms either way: the
    // weight gradients run beside the main chain, which is what the step waits for) — opt-in, tuning knob AMS_WG6_EIGHT_WAVES
    constexpr bool kWide = NW >= 6;
    const bool eight = kWide && knobs().wg6_eight_waves;
    static const std::string nm = "pw_wgrad_bf16x6<" + std::to_string(KT) + ", " + std::to_string(NW) + ">";
    note_kernel(nm.c_str());
    const int xd = (a.x_mode != 0 ? 1 : 0) | (a.dy_mode != 0 ? 2 : 0);
    if constexpr (kWide) {
        if (eight && xd == 0) return launch_wg6_k<KT, NW, 0, 8>(a, splits, tiles_k, tiles_n, rows, lds, st);
        if (eight && xd == 1) return launch_wg6_k<KT, NW, 1, 8>(a, splits, tiles_k, tiles_n, rows, lds, st);
    }
    switch (xd) {
        case 0: return launch_wg6_k<KT, NW, 0, 4>(a, splits, tiles_k, tiles_n, rows, lds, st);
        case 1: return launch_wg6_k<KT, NW, 1, 4>(a, splits, tiles_k, tiles_n, rows, lds, st);
        case 2: return launch_wg6_k<KT, NW, 2, 4>(a, splits, tiles_k, tiles_n, rows, lds, st);
        default: return launch_wg6_k<KT, NW, 3, 4>(a, splits, tiles_k, tiles_n, rows, lds, st);
    }
}

// writes `splits` partial [K,N] matrices into a.scratch; the caller reduces them
int launch_pointwise_wgrad_x6(const WgArgs& a, int splits, hipStream_t st) {
    int kt, nw;
    wg6_config(a.K, a.N, &kt, &nw);
#define WG6(KT_, NW_) if (kt == KT_ && nw == NW_) return launch_wg6_t<KT_, NW_>(a, splits, st);
    WG6(8, 4) WG6(8, 5) WG6(8, 6) WG6(8, 8) WG6(8, 10)
    WG6(4, 4) WG6(4, 5) WG6(4, 6) WG6(4, 8) WG6(4, 10)
#undef WG6
    set_error("wgrad_x6: no instantiation for K=%d N=%d", a.K, a.N);
    return AMS_E_INVALID;
}

}  // namespace ams
