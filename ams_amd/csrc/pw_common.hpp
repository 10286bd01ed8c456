// Device pieces shared by the pointwise (1x1 conv) GEMM kernels: epilogues, MFMA chunk, weight-panel staging.
#pragma once
#include <stdlib.h>

#include <string>

#include "kernels.hpp"
#include "split_bf16.hpp"

namespace ams {

// ---------------------------------------------------------------------------------------------------------
// Operand roles are SWAPPED in the MFMA (a = weights, b = activations): the 16x16 result then has the output
// channel along the accumulator registers (row = 4*(lane>>4) + i) and the pixel along lanes (col = lane & 15),
// so every lane owns 4 consecutive output channels of one pixel and the epilogue is float4 loads/stores
// (scale, shift, per-image bias, residual, result) instead of four scalar stores per tile.
// ---------------------------------------------------------------------------------------------------------
// BN scale / shift of this column tile are staged in LDS once per block (sSc, sSh: 16*NT floats each, identity where
// absent) and the residual / per-image-bias operands of a whole row group are requested before any of them is used:
// the epilogue then costs one memory round trip per row group instead of one per 16-column tile.
template <int NT>
__device__ __forceinline__ void pw_stage_affine(const PwArgs& a, float* sSc, float* sSh, int n0, int tid, int nthreads) {
    for (int e = tid; e < 16 * NT; e += nthreads) {
        const int n = n0 + e;
        sSc[e] = (a.scale && n < a.N) ? a.scale[n] : 1.f;
        sSh[e] = (a.shift && n < a.N) ? a.shift[n] : 0.f;
    }
}

template <int RM, int NT>
__device__ __forceinline__ void pw_epilogue(const PwArgs& a, f32x4 (&acc)[RM][NT], int64_t m_base, int n0, int l15, int q,
                                            const float* sSc, const float* sSh, int nrg = RM) {
    const bool y_vec = (a.ldy & 3) == 0, r_vec = (a.ldr & 3) == 0, n_vec = (a.N & 3) == 0;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int64_t m = m_base + r * 16 + l15;
        if (m >= a.M || r >= nrg) continue;
        float4 add[NT];                                   // residual + per-image bias contributions, gathered first
        bool has_add = false;
        if (a.res || a.img_bias) {
            has_add = true;
            const float* rp = a.res ? a.res + m * a.ldr : nullptr;
            const float* bp = a.img_bias ? a.img_bias + (m / a.rows_per_img) * a.N : nullptr;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n4 = n0 + 16 * t + 4 * q;
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), bv = rv;
                if (n4 + 3 < a.N) {
                    if (rp) rv = r_vec ? ld4(rp + n4) : make_float4(rp[n4], rp[n4 + 1], rp[n4 + 2], rp[n4 + 3]);
                    if (bp) bv = n_vec ? ld4(bp + n4) : make_float4(bp[n4], bp[n4 + 1], bp[n4 + 2], bp[n4 + 3]);
                } else if (n4 < a.N) {
                    float tr[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (n4 + i < a.N) { if (rp) tr[i] = rp[n4 + i]; if (bp) tb[i] = bp[n4 + i]; }
                    rv = make_float4(tr[0], tr[1], tr[2], tr[3]);
                    bv = make_float4(tb[0], tb[1], tb[2], tb[3]);
                }
                add[t] = rv;
                // the bias enters BEFORE scale/shift: fold it into the accumulator now
                acc[r][t][0] += bv.x; acc[r][t][1] += bv.y; acc[r][t][2] += bv.z; acc[r][t][3] += bv.w;
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c4 = 16 * t + 4 * q, n4 = n0 + c4;
            if (n4 >= a.N) continue;
            const float4 sc = ld4(sSc + c4), sh = ld4(sSh + c4);
            float4 v;
            const float4 bn = muladd4_pk(make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]), sc, sh);   // packed, two roundings
            v.x = apply_act(bn.x, a.act); v.y = apply_act(bn.y, a.act);
            v.z = apply_act(bn.z, a.act); v.w = apply_act(bn.w, a.act);
            if (has_add && a.res) { v.x += add[t].x; v.y += add[t].y; v.z += add[t].z; v.w += add[t].w; }
            float* yp = a.y + m * a.ldy + n4;
            if (n4 + 3 < a.N && y_vec) st4(yp, v);
            else {
                const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n4 + i < a.N) yp[i] = o[i];
            }
        }
    }
}

// Coalesced form of the epilogue: the MFMA result leaves a lane with 4 channels of ONE pixel, i.e. a wave-level store
// touches 16 different rows in 64-byte pieces (measured: 32-byte L1->L2 write requests, 4x the request count of a
// linear store).  Here each 16-row x 16*NT-column slab goes through a per-wave LDS staging buffer and is written back
// row-major: consecutive lanes hold consecutive float4 of a row, so stores (and the residual loads) are full lines.
// All global loads of a slab (per-image bias, residual) are issued together with clamped addresses and no branches, so
// the slab costs one memory round trip.  Requires N, ldy (and ldr) multiples of 4 and N >= 4.
// sOut: this wave's buffer, 16 x (16*NT + 4) floats.
enum { EPI_PLAIN = 0, EPI_RES = 1, EPI_BIAS = 2, EPI_GENERIC = 3 };

// SPLIT_OUT (split-bf16 kernel): when a.ysplit is set the finished rows are also written as bf16 parts (hi, mid[, lo] — the
// successive roundings of split8), so that the next block's expand GEMM loads its operand ready-made.
template <int RM, int NT, int EPI, bool SPLIT_OUT = false>
__device__ __forceinline__ void pw_epilogue_t(const PwArgs& a, f32x4 (&acc)[RM][NT], int64_t m_base, int n0, int lane,
                                              const float* sSc, const float* sSh, float* sOut, int nrg = RM) {
    constexpr int OP = 16 * NT + 4;
    constexpr int V4 = 4 * NT;                 // float4 per slab row
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        if (r >= nrg) break;                   // wave-uniform: a half-height block (k_pw_x3.hip) owns fewer row groups
        const int64_t m0 = m_base + r * 16;
        if (EPI == EPI_BIAS) {                 // added before scale/shift; row = this lane's pixel
            int64_t m = m0 + l15;
            if (m > a.M - 1) m = a.M - 1;
            const float* bp = a.img_bias + (m / a.rows_per_img) * a.N;
            float4 bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int n4 = n0 + 16 * t + 4 * q;
                if (n4 > a.N - 4) n4 = a.N - 4;
                bv[t] = ld4(bp + n4);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[r][t][0] += bv[t].x; acc[r][t][1] += bv[t].y; acc[r][t][2] += bv[t].z; acc[r][t][3] += bv[t].w; }
        }
        // no affine, no activation (every 1x1 product of the fine-tune step: its BN needs the complete statistics first): the tile goes to the
        // slab as it is — the multiply by 1, add of 0 and the activation switch were ~5 VALU instructions per element of a kernel whose
        // SIMDs spend a third of their time on VALU work (profiles: tools/pmc_f32s.sh)
        const bool ident = !a.scale && !a.shift && a.act == AMS_ACT_NONE;          // kernel-uniform
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c4 = 16 * t + 4 * q;
            float4 v = make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
            if (!ident) {
                const float4 sc = ld4(sSc + c4), sh = ld4(sSh + c4);
                const float4 bn = muladd4_pk(v, sc, sh);   // packed, two roundings
                v.x = apply_act(bn.x, a.act); v.y = apply_act(bn.y, a.act);
                v.z = apply_act(bn.z, a.act); v.w = apply_act(bn.w, a.act);
            }
            st4(sOut + l15 * OP + c4, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // wide tiles (NT > 6) take the slab in two halves: the residual operands of a whole 160-column row group are 40 registers
        constexpr int UG = NT > 6 ? (NT + 1) / 2 : NT;
#pragma unroll
        for (int u0 = 0; u0 < NT; u0 += UG) {
        float4 rv[UG];
        if (EPI == EPI_RES) {
#pragma unroll
            for (int uu = 0; uu < UG; ++uu) {
                const int u = u0 + uu < NT ? u0 + uu : NT - 1;
                const int f = lane + 64 * u;
                const int row = f / V4;
                int col = n0 + (f - row * V4) * 4;
                int64_t m = m0 + row;
                if (m > a.M - 1) m = a.M - 1;
                if (col > a.N - 4) col = a.N - 4;
                rv[uu] = ld4(a.res + m * a.ldr + col);
            }
        }
#pragma unroll
        for (int uu = 0; uu < UG; ++uu) {
            const int u = u0 + uu;
            if (u >= NT) break;
            const int f = lane + 64 * u;
            const int row = f / V4, c4 = (f - row * V4) * 4;
            const int64_t m = m0 + row;
            float4 v = ld4(sOut + row * OP + c4);
            if (EPI == EPI_RES) { v.x += rv[uu].x; v.y += rv[uu].y; v.z += rv[uu].z; v.w += rv[uu].w; }
            if (m < a.M && n0 + c4 < a.N) {
                // a result that also leaves as part planes is read next through THOSE (8 - 18 times, by the streaming kernel's channel blocks); its f32
                // form waits for the residual add a whole block later: stored non-temporal, it does not push the planes out of L2 (+0.4 % on the step)
                if (SPLIT_OUT && a.ysplit) __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(a.y + m * a.ldy + n0 + c4));
                else st4(a.y + m * a.ldy + n0 + c4, v);
            }
            if (SPLIT_OUT && a.ysplit && a.ysplit_fmt == 1 && m < a.M && n0 + c4 < a.N) {
                // two fp16 parts (hi | lo 2^11, split_bf16.hpp): planes [part][M][N]
                unsigned h[2], l[2];
                split4_f16(v, h, l);
                uint16_t* sp = a.ysplit + m * (int64_t)a.N + n0 + c4;
                *reinterpret_cast<uint2*>(sp) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(sp + a.ysplit_plane) = make_uint2(l[0], l[1]);
            } else if (SPLIT_OUT && a.ysplit && m < a.M && n0 + c4 < a.N) {
                const float f[4] = {v.x, v.y, v.z, v.w};
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 p0, p1, p2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const __bf16 h = (__bf16)f[j];
                    const float r1 = f[j] - (float)h;
                    const __bf16 md = (__bf16)r1;
                    p0[j] = h; p1[j] = md; p2[j] = (__bf16)(r1 - (float)md);
                }
                uint16_t* sp = a.ysplit + m * (int64_t)a.N + n0 + c4;
                *reinterpret_cast<bf16x4*>(sp) = p0;
                if (a.ysplit_np >= 2) *reinterpret_cast<bf16x4*>(sp + a.ysplit_plane) = p1;
                if (a.ysplit_np == 3) *reinterpret_cast<bf16x4*>(sp + 2 * a.ysplit_plane) = p2;
            }
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------------------
// Column reductions fused into the GEMM epilogue (PwArgs::red_mode).  Done in the MFMA layout, where a lane owns 4 consecutive columns
// (16t + 4q ..) of row l15 for every tile t: the lane's columns are the same for every row group it ever sees, so the sums stay in 8 NT
// registers until the wave is done; then 16 rows are folded by shuffles and the block's waves in a fixed order through LDS.
// Mode 2 also rewrites the accumulators (gradient x activation derivative): what the epilogue stores afterwards is the masked value.
// sRedVec (mode 2): scale | shift | mean | rstd of this column tile, 4 x 16 NT floats, staged by pw_red_stage.
// ---------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void pw_red_stage(const PwArgs& a, float* sRedVec, int n0, int tid, int nthreads) {
    for (int e = tid; e < 4 * 16 * NT; e += nthreads) {
        const int which = e / (16 * NT), n = n0 + e - which * (16 * NT);
        float v = which == 0 ? 1.f : 0.f;
        if (n < a.N) {
            if (a.red_mode == 2) v = (which == 0 ? a.red_scale : which == 1 ? a.red_shift : which == 2 ? a.red_mean : a.red_rstd)[n];
            else v = (which == 0 && a.red_center) ? a.red_center[n] : 0.f;
        }
        sRedVec[e] = v;
    }
}

template <int RM, int NT>
__device__ __forceinline__ void pw_red_rowgroups(const PwArgs& a, f32x4 (&acc)[RM][NT], int64_t m_base, int n0, int l15, int q,
                                                 const float* sRedVec, float4 (&s1)[NT], float4 (&s2)[NT], int nrg = RM) {
    const float lo = a.red_act == AMS_ACT_NONE ? -__builtin_huge_valf() : 0.f, hi = a.red_act == AMS_ACT_RELU6 ? 6.f : __builtin_huge_valf();
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        if (r >= nrg) break;
        int64_t m = m_base + r * 16 + l15;
        const float w = m < a.M ? 1.f : 0.f;               // rows beyond M (tail tiles) add nothing
        const bool whole = m_base + r * 16 + 16 <= a.M;    // wave-uniform: every row of this group exists (no weighting needed)
        if (m > a.M - 1) m = a.M - 1;
        if (a.red_mode == 2) {
            float4 zv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int n4 = n0 + 16 * t + 4 * q;
                if (n4 > a.N - 4) n4 = a.N - 4;
                zv[t] = ld4(a.red_z + m * a.ldy + n4);
            }
            if (a.red_res) {                               // the residual branch's gradient joins before the mask and the sums (acc + res, as EPI_RES adds it)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    int n4 = n0 + 16 * t + 4 * q;
                    if (n4 > a.N - 4) n4 = a.N - 4;
                    const float4 rv = ld4(a.red_res + m * a.red_ldr + n4);
                    acc[r][t][0] += rv.x; acc[r][t][1] += rv.y; acc[r][t][2] += rv.z; acc[r][t][3] += rv.w;
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c4 = 16 * t + 4 * q;
                const float4 sc = ld4(sRedVec + c4), sh = ld4(sRedVec + 16 * NT + c4), mu = ld4(sRedVec + 32 * NT + c4), rs = ld4(sRedVec + 48 * NT + c4);
                const float4 y = muladd4_pk(zv[t], sc, sh);
                float4 d;
                d.x = (y.x > lo && y.x < hi) ? acc[r][t][0] : 0.f; d.y = (y.y > lo && y.y < hi) ? acc[r][t][1] : 0.f;
                d.z = (y.z > lo && y.z < hi) ? acc[r][t][2] : 0.f; d.w = (y.w > lo && y.w < hi) ? acc[r][t][3] : 0.f;
                acc[r][t][0] = d.x; acc[r][t][1] = d.y; acc[r][t][2] = d.z; acc[r][t][3] = d.w;
                if (n0 + c4 < a.N) {
                    const float4 dw = whole ? d : make_float4(d.x * w, d.y * w, d.z * w, d.w * w);
                    s1[t] = add4_pk(s1[t], dw);
                    s2[t] = add4_pk(s2[t], mul4_pk(mul4_pk(dw, sub4_pk(zv[t], mu)), rs));
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int c4 = 16 * t + 4 * q;
                if (n0 + c4 >= a.N) continue;
                const float4 ctr = ld4(sRedVec + c4);
                float4 d = sub4_pk(make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]), ctr);
                if (!whole) d = make_float4(d.x * w, d.y * w, d.z * w, d.w * w);
                s1[t] = add4_pk(s1[t], d);
                s2[t] = add4_pk(s2[t], mul4_pk(d, d));
            }
        }
    }
}

// the wave's sums -> the block's partial row `row` of red_part ([2][N]); sRed: >= nwaves * 2 * 16 NT floats of LDS, free to use.
// Every wave of the block must call this (block barrier inside).
template <int NT>
__device__ __forceinline__ void pw_red_finish(const PwArgs& a, float4 (&s1)[NT], float4 (&s2)[NT], int lane, int wave, int nwaves, float* sRed,
                                              int64_t row, int n0, int tid, int nthreads) {
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float4 u = s1[t], v = s2[t];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            u.x += __shfl_xor(u.x, off, 64); u.y += __shfl_xor(u.y, off, 64); u.z += __shfl_xor(u.z, off, 64); u.w += __shfl_xor(u.w, off, 64);
            v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
        }
        if (l15 == 0) {
            st4(sRed + (wave * 2 + 0) * 16 * NT + 16 * t + 4 * q, u);
            st4(sRed + (wave * 2 + 1) * 16 * NT + 16 * t + 4 * q, v);
        }
    }
    __syncthreads();
    float* out = a.red_part + row * 2 * (int64_t)a.N;
    for (int e = tid; e < 2 * 16 * NT; e += nthreads) {
        const int which = e / (16 * NT), c = e - which * (16 * NT);
        if (n0 + c >= a.N) continue;
        float s = 0.f;
        for (int wv = 0; wv < nwaves; ++wv) s += sRed[(wv * 2 + which) * 16 * NT + c];
        out[(int64_t)which * a.N + n0 + c] = s;
    }
}

// the fused reduction needs a plain vector epilogue: the value reduced is the raw product
static inline bool pw_red_ok(const PwArgs& a, int64_t rows) {
    return a.red_mode != 0 && a.red_part && (size_t)rows * 2 * (size_t)a.N <= a.red_part_floats && !a.scale && !a.shift && !a.img_bias && !a.res && a.act == AMS_ACT_NONE && a.N % 4 == 0 && a.ldy % 4 == 0 &&
           a.N >= 4;
}

// which epilogue a problem can use
static inline int pw_pick_epi(const PwArgs& a) {
    const bool vec_ok = a.N >= 4 && (a.N & 3) == 0 && (a.ldy & 3) == 0 && (!a.res || (a.ldr & 3) == 0);
    if (!vec_ok || (a.res && a.img_bias)) return EPI_GENERIC;
    return a.res ? EPI_RES : a.img_bias ? EPI_BIAS : EPI_PLAIN;
}

// Column-tile width (in 16s) and row groups per wave for the tiled kernels.  Rule distilled from a sweep over the
// late-layer shapes on MI355X (tools/bench_kernel.py, AMS_PWX_FORCE): 64-wide tiles when they divide N, else 80, 48, 32;
// 96-wide tiles lose to LDS pressure; two row groups per wave only when that still leaves >= 2 blocks per CU.
static inline void pw_pick_tile(int64_t M, int N, int* rm_out, int* nt_out) {
    int nt;
    if (N % 64 == 0) nt = 4;
    else if (N % 80 == 0) nt = 5;
    else if (N % 48 == 0) nt = 3;
    else if (N % 32 == 0) nt = 2;
    else nt = N <= 16 ? 1 : N <= 32 ? 2 : N <= 48 ? 3 : 4;
    const int tiles_n = cdiv(N, 16 * nt);
    *rm_out = (cdiv64(M, 128) * tiles_n >= 512) ? 2 : 1;
    *nt_out = nt;
}

// Tail plan.  A launch of equal tiles whose block count is 1.05x or 2.1x the number of resident blocks ends with a round that
// leaves most of the chip idle (68640 rows x 320 columns at two blocks per CU: 1074 blocks on 512 slots).  The launcher may give
// the first k * slots blocks the full height and the remaining rows to half-height blocks (32 * RM rows, half the accumulators in
// use).  Cost model from measurements on MI355X (68640 x 960 -> 160 and -> 320, tools/probes/README.md): a round that fills the
// fraction f of the slots costs 0.35 + 0.65 f of a full one (a lone block per CU is bound by its own load -> split -> MFMA ->
// barrier chain, not by throughput), and a half-height round 0.79 of the full-height round with the same f.  Returns the number
// of full-height strips; the half-height strips that follow through *half_strips_out.
static inline int64_t pw_plan_tail(int64_t M, int rm, int n_tiles_n, int slots, int64_t* half_strips_out, int64_t rows_full = 0) {
    if (rows_full <= 0) rows_full = 64 * rm;             // four waves of 16 rm rows; blocks of more waves pass their own height
    const int64_t rows_half = rows_full / 2;
    const int64_t full_all = cdiv64(M, rows_full);
    *half_strips_out = 0;
    if (rm < 2 || slots <= 0 || knobs().pwx_no_tail) return full_all;
    const int64_t spr = slots / n_tiles_n > 0 ? slots / n_tiles_n : 1;          // strips per round
    auto rounds = [&](int64_t strips) {
        const int64_t whole = strips / spr, rest = strips % spr;
        return (double)whole + (rest ? 0.35 + 0.65 * (double)rest / (double)spr : 0.0);
    };
    double best = rounds(full_all);
    int64_t best_full = full_all;
    for (int64_t k = 0; k * spr < full_all; ++k) {
        const int64_t full = k * spr;
        const int64_t halves = cdiv64(M - full * rows_full, rows_half);
        const double cost = (double)k + 0.79 * rounds(halves);
        if (cost < best - 1e-9) { best = cost; best_full = full; *half_strips_out = halves; }
    }
    return best_full;
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

}  // namespace ams
